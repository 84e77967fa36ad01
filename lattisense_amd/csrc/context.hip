// context.hip — parameter-set context: device tables, cached conversion plans, workspace.
#include "build_flags.h"
#include "lsa_internal.h"
#include <cstdlib>

namespace lsa {

const char* context_build_flags() { return LSA_BUILD_FLAGS_TEXT; }

static thread_local std::string g_last_error;
void set_last_error(const std::string& m) { g_last_error = m; }
const std::string& last_error() { return g_last_error; }

Context::Context(int algo_, int n_, const u64* q, int nq_, const u64* p, int np_, u64 t_, int device_)
    : algo(algo_), n(n_), nq(nq_), np(np_), device(device_), t(t_) {
    LSA_REQUIRE(algo == LSA_ALGO_BFV || algo == LSA_ALGO_CKKS, "unknown algorithm");
    LSA_REQUIRE(n >= 512 && n <= (1 << 17) && (n & (n - 1)) == 0, "ring degree must be a power of two in [2^9, 2^17]");
    LSA_REQUIRE(nq >= 1 && np >= 0 && np <= LSA_BC_MAX_SRC, "bad modulus chain lengths");
    std::vector<u64> mods(q, q + nq);
    mods.insert(mods.end(), p, p + np);
    nmul = 0;
    int lg = 0;
    while ((1 << lg) < n) lg++;
    if (algo == LSA_ALGO_BFV) {
        LSA_REQUIRE(t > 1, "BFV needs a plaintext modulus");
        nmul = bfv_aux_count(q, nq, lg);
        std::vector<u64> aux = gen_aux_primes(n, nmul, mods);
        mods.insert(mods.end(), aux.begin(), aux.end());
    }
    nmod = (int)mods.size();
    LSA_REQUIRE(nmod < LSA_ROW_SKIP, "too many moduli");
    T.build(n, mods);
    logn = T.logn;
#ifndef LSA_NTT_TAU
#define LSA_NTT_TAU 12   // log2 of the LDS tile (points per workgroup pass)
#endif
    int mu_a = 0;
    if (const char* e = std::getenv("LSA_NTT_MU_A")) mu_a = std::atoi(e);   // A/B: stages of the first pass
    plan = make_ntt_plan(logn, LSA_NTT_TAU, mu_a);
    // N = 2^13 / 2^14: the whole limb also fits one workgroup's LDS (69 / 136 KiB) and can be transformed in a single pass
    // of 512 / 1024 threads: half the HBM traffic, but one or two workgroups per CU and R limbs fill only R CUs --
    // launch_ntt picks per launch; LSA_NTT_WIDE=0 / 1 forces never / always.
    plan_wide = make_ntt_plan(logn, (logn == 13 || logn == 14) ? logn : LSA_NTT_TAU);
    if (const char* e = std::getenv("LSA_NTT_FP_RAW")) fp_raw = e[0] != '0';
    if (const char* wide = std::getenv("LSA_NTT_WIDE")) wide_mode = wide[0] == '0' ? 0 : wide[0] == '1' ? 1 : 2;
    LSA_REQUIRE(plan.npass == 1 || plan.pass[1].mu <= plan.pass[1].tau, "ring degree too large for the NTT tile size");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        throw Error(LSA_ERR_NO_DEVICE, "no HIP device available: this library has no CPU fallback");
    LSA_REQUIRE(device >= 0 && device < ndev, "device index out of range");
    use_device();
    LSA_HIP(hipMalloc((void**)&d_mods, nmod * sizeof(ModDev)));
    LSA_HIP(hipMalloc((void**)&d_scale, (size_t)nmod * 4 * sizeof(u64)));
    LSA_HIP(hipMalloc((void**)&d_scale_d, (size_t)nmod * 2 * sizeof(double)));
    LSA_HIP(hipMemcpy(d_mods, T.mods.data(), nmod * sizeof(ModDev), hipMemcpyHostToDevice));
    LSA_HIP(hipMemcpy(d_scale, T.scale.data(), (size_t)nmod * 4 * sizeof(u64), hipMemcpyHostToDevice));
    // twiddle tables in the order the plan's sub-passes read them (ntt_core.h, "twiddle table layout"); the whole-limb plan of
    // N = 2^13 / 2^14 groups its stages differently and gets its own copies
    auto upload = [&](const NttPlan& pl, u64** psi, u64** psiinv, double** psi_d, double** psiinv_d) {
        const size_t tw_bytes = (size_t)nmod * n * sizeof(u64);
        std::vector<u64> pi(T.psi.size()), pii(T.psiinv.size());
        std::vector<double> pd(T.psi_d.size()), pdi(T.psiinv_d.size());
        for (int m = 0; m < nmod; m++) {
            const size_t o = (size_t)m * n;
            ntt_permute_twiddles(pl, logn, T.psi.data() + 2 * o, pi.data() + 2 * o, 2, false);
            ntt_permute_twiddles(pl, logn, T.psiinv.data() + 2 * o, pii.data() + 2 * o, 2, false);
            ntt_permute_twiddles(pl, logn, T.psi_d.data() + o, pd.data() + o, 1, true);
            ntt_permute_twiddles(pl, logn, T.psiinv_d.data() + o, pdi.data() + o, 1, true);
        }
        LSA_HIP(hipMalloc((void**)psi, 2 * tw_bytes));      // {w, Shoup quotient} pairs
        LSA_HIP(hipMalloc((void**)psiinv, 2 * tw_bytes));
        LSA_HIP(hipMalloc((void**)psi_d, tw_bytes));
        LSA_HIP(hipMalloc((void**)psiinv_d, tw_bytes));
        LSA_HIP(hipMemcpy(*psi, pi.data(), 2 * tw_bytes, hipMemcpyHostToDevice));
        LSA_HIP(hipMemcpy(*psiinv, pii.data(), 2 * tw_bytes, hipMemcpyHostToDevice));
        LSA_HIP(hipMemcpy(*psi_d, pd.data(), tw_bytes, hipMemcpyHostToDevice));
        LSA_HIP(hipMemcpy(*psiinv_d, pdi.data(), tw_bytes, hipMemcpyHostToDevice));
    };
    upload(plan, &d_psi, &d_psiinv, &d_psi_d, &d_psiinv_d);
    if (plan_wide.npass < plan.npass) upload(plan_wide, &d_psi_w, &d_psiinv_w, &d_psi_d_w, &d_psiinv_d_w);
    LSA_HIP(hipMemcpy(d_scale_d, T.scale_d.data(), (size_t)nmod * 2 * sizeof(double), hipMemcpyHostToDevice));
}

Context::~Context() {
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    (void)hipFree(d_mods);
    (void)hipFree(d_psi);
    (void)hipFree(d_psiinv);
    (void)hipFree(d_scale);
    (void)hipFree(d_psi_d);
    (void)hipFree(d_psiinv_d);
    (void)hipFree(d_psi_w);
    (void)hipFree(d_psiinv_w);
    (void)hipFree(d_psi_d_w);
    (void)hipFree(d_psiinv_d_w);
    (void)hipFree(d_scale_d);
    (void)hipFree(ws);
    (void)hipFree(ws2);
    if (aux_stream) {
        (void)hipStreamDestroy(aux_stream);
        (void)hipEventDestroy(ev_fork);
        (void)hipEventDestroy(ev_join);
    }
    for (auto& sm : prof_samples) {
        (void)hipEventDestroy(sm.e0);
        (void)hipEventDestroy(sm.e1);
    }
    for (auto& e : prof_pool) (void)hipEventDestroy(e);
    for (auto& kv : bconv) (void)hipFree(kv.second.dev);
    for (auto& kv : perm_ntt) (void)hipFree(kv.second);
    for (auto& kv : perm_coeff) (void)hipFree(kv.second);
    for (auto& kv : consts) (void)hipFree(kv.second);
}

u64* Context::workspace(size_t words, hipStream_t s) {
    if (words > ws_words) {
        // growing: earlier users of the arena were enqueued on `s` (one in-flight operator per context)
        LSA_HIP(hipStreamSynchronize(s));
        if (ws) LSA_HIP(hipFree(ws));
        ws = nullptr;
        ws_words = 0;
        LSA_HIP(hipMalloc((void**)&ws, words * sizeof(u64)));
        ws_words = words;
    }
    return ws;
}

void Context::fork_aux(hipStream_t s) {
    if (!aux_stream) {
        LSA_HIP(hipStreamCreateWithFlags(&aux_stream, hipStreamNonBlocking));
        LSA_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        LSA_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    }
    LSA_HIP(hipEventRecord(ev_fork, s));
    LSA_HIP(hipStreamWaitEvent(aux_stream, ev_fork, 0));
}

void Context::join_aux(hipStream_t s) {
    LSA_HIP(hipEventRecord(ev_join, aux_stream));
    LSA_HIP(hipStreamWaitEvent(s, ev_join, 0));
}

u64* Context::workspace2(size_t words, hipStream_t s) {
    if (words > ws2_words) {
        LSA_HIP(hipStreamSynchronize(s));
        if (ws2) LSA_HIP(hipFree(ws2));
        ws2 = nullptr;
        ws2_words = 0;
        LSA_HIP(hipMalloc((void**)&ws2, words * sizeof(u64)));
        ws2_words = words;
    }
    return ws2;
}

const BaseConvPlan* Context::baseconv(const std::vector<int>& src, const std::vector<int>& dst, bool centered,
                                      bool pinv_scaled, const BaseConvFold* fold) {
    std::string key = centered ? "c" : "u";
    if (pinv_scaled) key += "s";
    if (fold) {
        LSA_REQUIRE(!fold->tag.empty() && (fold->src_pre.empty() || fold->src_pre.size() == src.size()) &&
                        (fold->dst_scale.empty() || fold->dst_scale.size() == dst.size()),
                    "base conversion: folded factors do not match the limb lists");
        key += "f:" + fold->tag;
    }
    for (int x : src) key += "|" + std::to_string(x);
    key += "->";
    for (int x : dst) key += "|" + std::to_string(x);
    std::lock_guard<std::mutex> lk(mu);
    auto it = bconv.find(key);
    if (it != bconv.end()) return &it->second;
    const int ns = (int)src.size(), nd = (int)dst.size();
    LSA_REQUIRE(ns >= 1 && ns <= LSA_BC_MAX_SRC && nd >= 1 && nd <= LSA_BC_MAX_DST, "base conversion too wide");
    auto K = std::make_unique<BaseConvConsts>();
    K->ns = ns;
    K->nd = nd;
    K->centered = centered ? 1 : 0;
    for (int i = 0; i < ns; i++) {
        const u64 qi = T.mod[src[i]];
        K->src_mod[i] = src[i];
        u64 pr = 1 % qi;
        for (int l = 0; l < ns; l++)
            if (l != i) pr = mul_mod_host(pr, T.mod[src[l]] % qi, qi);
        K->shat_inv_m[i] = to_mont_host(inv_mod(pr, qi), qi);
        K->half_src[i] = mul_mod_host(qi - 1, (qi + 1) >> 1, qi);  // floor(S/2) mod q_i with S == 0 mod q_i, S odd
        if (fold && !fold->src_pre.empty()) {
            // y_i = (x * f + half) * shat_inv = (x + half * f^-1) * (f * shat_inv): the kernel's add-then-multiply shape is kept
            const u64 f = fold->src_pre[i] % qi;
            LSA_REQUIRE(f != 0, "base conversion: folded source factor is zero");
            K->shat_inv_m[i] = to_mont_host(mul_mod_host(inv_mod(pr, qi), f, qi), qi);
            K->half_src[i] = mul_mod_host(K->half_src[i], inv_mod(f, qi), qi);
        }
        K->qf[i] = (double)qi;
        K->rf[i] = 1.0 / (double)qi;
    }
    for (int j = 0; j < nd; j++) {
        const u64 pj = T.mod[dst[j]];
        K->dst_mod[j] = dst[j];
        u64 all = 1 % pj;
        for (int l = 0; l < ns; l++) all = mul_mod_host(all, T.mod[src[l]] % pj, pj);
        u64 scale = 1 % pj;   // pinv_scaled: every constant of the (linear) output formula times P^-1 mod p_j
        if (pinv_scaled && j + 1 < nd) {
            u64 pp = 1;
            for (int l = 0; l < np; l++) pp = mul_mod_host(pp, T.mod[p_mod(l)] % pj, pj);
            scale = inv_mod(pp, pj);
        }
        if (fold && !fold->dst_scale.empty()) scale = mul_mod_host(scale, fold->dst_scale[j] % pj, pj);
        for (int i = 0; i < ns; i++) {
            u64 pr = 1 % pj;
            for (int l = 0; l < ns; l++)
                if (l != i) pr = mul_mod_host(pr, T.mod[src[l]] % pj, pj);
            K->shat_m[j][i] = to_mont_host(mul_mod_host(pr, scale, pj), pj);
        }
        for (int v = 0; v <= ns; v++) K->vs[j][v] = mul_mod_host(mul_mod_host((u64)v % pj, all, pj), scale, pj);
        K->half_dst[j] = mul_mod_host(mul_mod_host((all + pj - 1 % pj) % pj, (pj + 1) >> 1, pj), scale, pj);
    }
    bool small = std::getenv("LSA_BC_NO_SPLIT") == nullptr;
    for (int i = 0; i < ns; i++) small = small && (T.mod[src[i]] >> 58) == 0;
    for (int j = 0; j < nd; j++) small = small && (T.mod[dst[j]] >> 58) == 0;
    K->split29 = small ? 1 : 0;
    for (int j = 0; j < nd; j++)
        for (int i = 0; i < ns; i++) {
            K->shat_lo[j][i] = (u32)(K->shat_m[j][i] & ((1u << 29) - 1));
            K->shat_hi[j][i] = (u32)(K->shat_m[j][i] >> 29);
            K->shat_sum[j][i] = K->shat_lo[j][i] + K->shat_hi[j][i];
        }
    for (int j = 0; j < nd; j++) {
        const u64 pj = T.mod[dst[j]];
        K->corr_a[j] = to_mont_host((pj - K->vs[j][1]) % pj, pj);   // vs[j][1] = S (times the P^-1 scale) mod p_j
        K->corr_b[j] = centered ? to_mont_host((pj - K->half_dst[j]) % pj, pj) : 0;
    }
    use_device();
    BaseConvConsts* d = nullptr;
    LSA_HIP(hipMalloc((void**)&d, sizeof(BaseConvConsts)));
    LSA_HIP(hipMemcpy(d, K.get(), sizeof(BaseConvConsts), hipMemcpyHostToDevice));
    BaseConvPlan pl;
    pl.dev = d;
    pl.ns = ns;
    pl.nd = nd;
    pl.split29 = small;
    bconv[key] = pl;
    return &bconv[key];
}

static unsigned brv_bits(unsigned x, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; i++) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
}

// NTT-domain automorphism X -> X^g as a gather table: out[i] = in[perm[i]]
const u32* Context::ntt_perm(u64 g) {
    std::lock_guard<std::mutex> lk(mu);
    auto it = perm_ntt.find(g);
    if (it != perm_ntt.end()) return it->second;
    LSA_REQUIRE((g & 1) == 1 && g < 2 * (u64)n, "Galois element must be odd and < 2N");
    std::vector<u32> h(n);
    const u64 mask = 2 * (u64)n - 1;
    for (int i = 0; i < n; i++) {
        const u64 e = 2 * (u64)brv_bits((unsigned)i, logn) + 1;
        const u64 e2 = (g * e) & mask;
        h[i] = brv_bits((unsigned)((e2 - 1) >> 1), logn);
    }
    use_device();
    u32* d = nullptr;
    LSA_HIP(hipMalloc((void**)&d, n * sizeof(u32)));
    LSA_HIP(hipMemcpy(d, h.data(), n * sizeof(u32), hipMemcpyHostToDevice));
    perm_ntt[g] = d;
    return d;
}

// coefficient-domain automorphism as a gather with sign: out[o] = (+/-) in[src(o)], bit 31 = negate
const u32* Context::coeff_perm(u64 g) {
    std::lock_guard<std::mutex> lk(mu);
    auto it = perm_coeff.find(g);
    if (it != perm_coeff.end()) return it->second;
    LSA_REQUIRE((g & 1) == 1 && g < 2 * (u64)n, "Galois element must be odd and < 2N");
    std::vector<u32> h(n);
    const u64 mask = 2 * (u64)n - 1;
    for (int i = 0; i < n; i++) {
        u64 idx = ((u64)i * g) & mask;
        u32 sign = 0;
        if (idx >= (u64)n) {
            idx -= n;
            sign = 1u << 31;
        }
        h[idx] = (u32)i | sign;
    }
    use_device();
    u32* d = nullptr;
    LSA_HIP(hipMalloc((void**)&d, n * sizeof(u32)));
    LSA_HIP(hipMemcpy(d, h.data(), n * sizeof(u32), hipMemcpyHostToDevice));
    perm_coeff[g] = d;
    return d;
}

// P^-1 mod q_i for i <= level (ModDown tail), Montgomery form
const u64* Context::pinv_vec(int level) {
    const int L = level + 1;
    std::vector<int> mods(L);
    std::vector<u64> pinv(L);
    for (int i = 0; i < L; i++) {
        mods[i] = i;
        u64 q = T.mod[i], pr = 1;
        for (int l = 0; l < np; l++) pr = mul_mod_host(pr, T.mod[p_mod(l)] % q, q);
        pinv[i] = inv_mod(pr, q);
    }
    return const_vec("pinv" + std::to_string(L), mods, pinv);
}

// q_level^-1 mod q_i for i < level (rescale tail), Montgomery form
const u64* Context::pmodq_vec(int level) {
    const int L = level + 1;
    std::vector<int> mods(L);
    std::vector<u64> pm(L);
    for (int j = 0; j < L; j++) {
        mods[j] = j;
        const u64 q = T.mod[j];
        u64 pr = 1;
        for (int l = 0; l < np; l++) pr = mul_mod_host(pr, T.mod[p_mod(l)] % q, q);
        pm[j] = pr;
    }
    return const_vec("pmodq" + std::to_string(L), mods, pm);
}

const u64* Context::qlinv_vec(int level) {
    std::vector<int> mods(level);
    std::vector<u64> v(level);
    const u64 ql = T.mod[level];
    for (int i = 0; i < level; i++) {
        mods[i] = i;
        v[i] = inv_mod(ql % T.mod[i], T.mod[i]);
    }
    return const_vec("qlinv" + std::to_string(level), mods, v);
}

const u64* Context::const_vec(const std::string& name, const std::vector<int>& mods, const std::vector<u64>& vals) {
    std::lock_guard<std::mutex> lk(mu);
    auto it = consts.find(name);
    if (it != consts.end()) return it->second;
    std::vector<u64> m(vals.size());
    for (size_t i = 0; i < vals.size(); i++) m[i] = to_mont_host(vals[i] % T.mod[mods[i]], T.mod[mods[i]]);
    use_device();
    u64* d = nullptr;
    LSA_HIP(hipMalloc((void**)&d, m.size() * sizeof(u64)));
    LSA_HIP(hipMemcpy(d, m.data(), m.size() * sizeof(u64), hipMemcpyHostToDevice));
    consts[name] = d;
    return d;
}

const u64* Context::raw_vec(const std::string& name, const std::vector<u64>& vals) {
    std::lock_guard<std::mutex> lk(mu);
    auto it = consts.find(name);
    if (it != consts.end()) return it->second;
    use_device();
    u64* d = nullptr;
    LSA_HIP(hipMalloc((void**)&d, vals.size() * sizeof(u64)));
    LSA_HIP(hipMemcpy(d, vals.data(), vals.size() * sizeof(u64), hipMemcpyHostToDevice));
    consts[name] = d;
    return d;
}

}  // namespace lsa
