// bootstrap.hip — CKKS bootstrapping as a program of the device operators (the `bootstrap` node of a task graph).
//
// Reference: mega_ag_runners/gpu/mega_ag_executors_gpu.cu:410-426 hands the ciphertext to HEonGPU's
// regular_bootstrapping_v2 (absent submodule); parameters mega_ag_runners/gpu/gpu_wrapper.cu:86-117 and
// frontend/custom_task.py:383-468; the rotations a caller generates Galois keys for: frontend/bootstrap_params.py:104-263.
// The algorithm is the one restated (and explained) in oracle/ckks_bootstrap.py, which is also its oracle:
//   scale-up by an integer -> (switch to the sparse secret) -> ModRaise -> (switch back) -> CoeffsToSlots (merged radix-2
//   layers of the inverse special FFT, baby-step / giant-step with the planner's split) -> conjugate split into real and
//   imaginary coefficients -> EvalMod (Chebyshev interpolant of cos on [-1,1] by binary splitting, double-angle steps) ->
//   recombine -> SlotsToCoeffs.
// Only floating-point CONSTANTS are computed here (the encoded diagonals, the Chebyshev coefficients); every operation on
// ciphertexts is one of the integer operators of ops.hip / kernels.hip, so a replay of the same program with the same
// constants on the CPU oracle gives identical residues (tests/test_gpu_bootstrap.py).
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>

#include "lsa_internal.h"

namespace lsa {

void ckks_rescale(Context&, int, int, const u64*, u64*, int, long long, long long, hipStream_t);
void ckks_rotate(Context&, int, const u64*, u64, const Key&, u64*, int, long long, long long, hipStream_t);
void ckks_switch_key(Context&, int, const u64*, const Key&, u64*, int, long long, long long, hipStream_t);
void ckks_rotate_many(Context&, int, const u64*, int, const u64*, const Key* const*, u64* const*, int, long long, long long,
                      hipStream_t);
void ckks_mult_relin_rescale(Context&, int, const u64*, const u64*, const Key&, u64*, int, long long, long long, long long,
                             hipStream_t);
void ckks_mult_relin_rescale_rpp(Context&, int, const u64*, const u64*, const Key&, u64*, int, long long, long long, long long,
                                 hipStream_t, int, int);
void ckks_lift_ext(Context&, int, const u64*, u64*, int, long long, long long, hipStream_t);
void ckks_rotate_many_ext(Context&, int, const u64*, int, const u64*, const Key* const*, u64* const*, int, long long, long long,
                          hipStream_t);
void ckks_rotate_ext(Context&, int, const u64*, u64, const Key&, u64*, bool, int, long long, long long, hipStream_t);
void ckks_moddown_ext(Context&, int, u64*, u64*, int, long long, long long, hipStream_t);

using cplx = std::complex<double>;
using Diags = std::map<int, std::vector<cplx>>;   // diagonal k: d[t] multiplies x[(t + k) mod n]

namespace {

const double kPi = 3.14159265358979323846;

std::vector<int> rot_group(int n_slots) {
    std::vector<int> g(n_slots);
    long long v = 1;
    const long long m = 4LL * n_slots;
    for (int i = 0; i < n_slots; i++) {
        g[i] = (int)v;
        v = v * 5 % m;
    }
    return g;
}

// one radix-2 layer of the special FFT (forward: slots <- coefficients) or of its inverse, as three diagonals
Diags layer_diagonals(int n, int length, bool inverse, const std::vector<int>& rg) {
    const long long m = 4LL * n;
    const int lenh = length / 2;
    const long long lenq = 4LL * length;
    std::vector<cplx> d0(n), dp(n), dm(n);
    auto root = [&](long long idx) { return std::polar(1.0, 2.0 * kPi * (double)idx / (double)m); };
    for (int t = 0; t < n; t++) {
        const int j = t % length;
        if (j < lenh) {
            d0[t] = 1.0;
            dp[t] = inverse ? cplx(1.0) : root((rg[j] % lenq) * (m / lenq));
        } else {
            const int jj = j - lenh;
            if (inverse) {
                const cplx w = root((lenq - (rg[jj] % lenq)) * (m / lenq));
                d0[t] = -w;
                dm[t] = w;
            } else {
                d0[t] = -root((rg[jj] % lenq) * (m / lenq));
                dm[t] = 1.0;
            }
        }
    }
    Diags out;
    out[0] = d0;
    const int kp = lenh % n, km = ((-lenh) % n + n) % n;
    if (kp == km) {   // the widest layer: +n/2 and -n/2 are the same rotation
        for (int t = 0; t < n; t++) dp[t] += dm[t];
        out[kp] = dp;
    } else {
        out[kp] = dp;
        out[km] = dm;
    }
    return out;
}

// diagonals of (second o first)
Diags compose(const Diags& first, const Diags& second, int n) {
    Diags out;
    for (auto& a : second)
        for (auto& b : first) {
            const int k = (a.first + b.first) % n;
            auto& dst = out[k];
            if (dst.empty()) dst.assign(n, cplx(0.0));
            for (int t = 0; t < n; t++) dst[t] += a.second[t] * b.second[(t + a.first) % n];
        }
    for (auto it = out.begin(); it != out.end();) {
        double mx = 0;
        for (auto& v : it->second) mx = std::max(mx, std::abs(v));
        it = mx > 1e-300 ? std::next(it) : out.erase(it);
    }
    return out;
}

// `depth` merged groups in application order, grouped like frontend/bootstrap_params.py:104-119
std::vector<Diags> merged_matrices(int n, int depth, bool inverse) {
    int log_n = 0;
    while ((1 << log_n) < n) log_n++;
    LSA_REQUIRE(depth >= 1 && depth <= log_n, "bootstrap: linear-transform depth out of range");
    const std::vector<int> rg = rot_group(n);
    std::vector<int> lengths;
    if (inverse)
        for (int l = log_n; l >= 1; l--) lengths.push_back(1 << l);
    else
        for (int l = 1; l <= log_n; l++) lengths.push_back(1 << l);
    std::vector<int> sizes;
    int left = log_n;
    for (int i = 0; i < depth; i++) {
        const int s = (left + (depth - i) - 1) / (depth - i);
        sizes.push_back(s);
        left -= s;
    }
    if (!inverse) std::reverse(sizes.begin(), sizes.end());
    std::vector<Diags> mats;
    size_t pos = 0;
    for (int s : sizes) {
        Diags m;
        for (int i = 0; i < s; i++) {
            Diags lay = layer_diagonals(n, lengths[pos + i], inverse, rg);
            m = i == 0 ? lay : compose(m, lay, n);
        }
        mats.push_back(std::move(m));
        pos += s;
    }
    return mats;
}

// group sizes of `depth` merged matrices over log_n layers (frontend/bootstrap_params.py:104-119)
std::vector<int> group_sizes(int log_n, int depth, bool inverse) {
    std::vector<int> sizes;
    int left = log_n;
    for (int i = 0; i < depth; i++) {
        const int s = (left + (depth - i) - 1) / (depth - i);
        sizes.push_back(s);
        left -= s;
    }
    if (!inverse) std::reverse(sizes.begin(), sizes.end());
    return sizes;
}

std::vector<cplx> tiled(const std::vector<cplx>& v, int n) {
    std::vector<cplx> out(n);
    for (int t = 0; t < n; t++) out[t] = v[t % v.size()];
    return out;
}

void bsgs_sets(const std::vector<int>& ks, int n, int n1, std::vector<int>& giants, std::vector<int>& babies) {
    std::map<int, bool> g, b;
    for (int k : ks) {
        g[((k % n) / n1) * n1 % n] = true;
        b[(k % n) % n1] = true;
    }
    giants.clear();
    babies.clear();
    for (auto& kv : g) giants.push_back(kv.first);
    for (auto& kv : b) babies.push_back(kv.first);
}

// the planner's baby-step count (frontend/bootstrap_params.py:193-207): the caller's Galois keys exist for this choice
int bsgs_split(const std::vector<int>& ks, int n, double ratio) {
    int n1 = 1;
    std::vector<int> g, b;
    while (n1 < n) {
        bsgs_sets(ks, n, n1, g, b);
        const int nb_g = (int)g.size() - 1, nb_b = (int)b.size() - 1;
        if (nb_g == 0 || (double)nb_b / nb_g == ratio) return n1;
        if ((double)nb_b / nb_g > ratio) return n1 / 2;
        n1 <<= 1;
    }
    return 1;
}

// slots -> coefficients: t = U^-1 z (inverse special FFT), m_k = Re t_k, m_{k+n} = Im t_k
std::vector<double> slots_to_coeffs(std::vector<cplx> v, const std::vector<int>& rg) {
    const int n = (int)v.size();
    const long long m = 4LL * n;
    for (int len = n; len >= 2; len >>= 1) {
        const int lenh = len >> 1;
        const long long lenq = 4LL * len;
        for (int i = 0; i < n; i += len)
            for (int j = 0; j < lenh; j++) {
                const long long idx = (lenq - (rg[j] % lenq)) * (m / lenq);
                const cplx w = std::polar(1.0, 2.0 * kPi * (double)idx / (double)m);
                const cplx a = v[i + j], b = v[i + j + lenh];
                v[i + j] = a + b;
                v[i + j + lenh] = (a - b) * w;
            }
    }
    int lg = 0;
    while ((1 << lg) < n) lg++;
    std::vector<double> out(2 * (size_t)n);
    for (int i = 0; i < n; i++) {
        int r = 0;
        for (int b = 0; b < lg; b++) r |= ((i >> b) & 1) << (lg - 1 - b);
        out[i] = v[r].real() / n;
        out[i + n] = v[r].imag() / n;
    }
    return out;
}

// Python's round(): ties to even.  A constant beyond 2^62 means the modulus chain does not fit the level plan (e.g. EvalMod
// running on primes much smaller than its scale): refuse instead of computing garbage.
long long round_even(double v) {
    LSA_REQUIRE(std::fabs(v) < 4.6e18, "bootstrap: encoded constant out of range -- the modulus chain does not match the "
                                       "bootstrap level plan (depths / scales)");
    return (long long)std::nearbyint(v);
}

}  // namespace

// ------------------------------------------------------------------------------------------------ plan
struct BtMatrix {
    int level = 0, n1 = 1;
    int period = 0;                      // period of the diagonals (index arithmetic mod period); N/2 for dense packing
    bool naive = false;
    std::vector<int> ks;                 // diagonal indices, ascending
    std::vector<u64*> plains;            // per diagonal: NTT-domain plaintext [rows][N] of rot_{-giant}(diag)
    int rows = 0;                        // level + 1, or level + 1 + k (the special primes too) for a double-hoisted matrix
};

struct Bootstrap {
    Context& c;
    int cts_depth, stc_depth, K, r, top_level;
    double mr, in_scale, out_scale;
    int log_slots = 0;                   // 0 / logN-1: dense packing; less: sparse (subring of X^gap)
    bool sparse = false;
    std::vector<BtMatrix> cts, stc;
    BtMatrix p1, p2;                     // sparse: last CoeffsToSlots matrix split for x and conj(x), real | imaginary halves
    std::vector<double> cheb;            // 2^cheb_depth Chebyshev coefficients of cos(2 pi (K x - 1/4) / 2^r) on [-1,1]
    int sine_deg = 30, arcsine_deg = 0;  // reference: btp_eval_mod_sine_deg / _arcsine_deg (gpu_wrapper.cu:100-103)
    int cheb_depth = 5, asin_depth = 0;  // levels of the cosine interpolant (ceil log2(sine_deg + 1)) and of the arcsine polynomial
    std::vector<double> asin_coef;       // arcsine Taylor coefficients in the monomial basis, padded to 2^asin_depth (empty: none)
    // baby-step / giant-step matrices keep their sums over Q u P and divide by P once per giant step + once at the end
    // (Lattigo's MultiplyByDiagMatrixBSGS); LSA_BT_DOUBLE_HOIST=0 at plan creation: one division per rotation (A/B)
    bool double_hoist = true;
    int evalmod_depth() const { return cheb_depth + r + asin_depth; }
    u64* mono[2] = {nullptr, nullptr};   // NTT of +X^(N/2) and -X^(N/2), [top_level+1][N]
    std::vector<u64> galois;             // Galois elements a run needs (rotations + conjugation)
    long long mul_c = 1;                 // integer scale-up factor
    double d1 = 0, natural_scale = 0;
    std::vector<u64*> owned;
    // temporaries of a run, kept across runs (hipMalloc / hipFree of GiB-sized buffers cost more than the kernels); a plan
    // belongs to one context and is run on one in-order stream at a time, so reuse is ordered
    std::multimap<size_t, u64*> pool;
    std::vector<u64*> pool_all;

    explicit Bootstrap(Context& ctx) : c(ctx) {}
    ~Bootstrap() {
        (void)hipSetDevice(c.device);
        (void)hipDeviceSynchronize();
        for (u64* p : owned) (void)hipFree(p);
        for (u64* p : pool_all) (void)hipFree(p);
    }

    RowMap rm_limbs(int L) const {
        RowMap rm;
        rm.period = L;
        for (int j = 0; j < L; j++) rm.mod_of[j] = (unsigned char)j;
        return rm;
    }

    // real polynomial coefficients * scale -> NTT-domain plaintext on the device
    // ext: the same integer polynomial at the special primes too, rows level+1 .. level+k (operand of extended ciphertexts)
    u64* upload_plain(const std::vector<double>& coef, double scale, int level, hipStream_t s, bool ext = false) {
        const int L = level + 1 + (ext ? c.np : 0);
        const size_t N = (size_t)c.n;
        std::vector<u64> host((size_t)L * N);
        RowMap rm = rm_limbs(L);
        for (int j = level + 1; j < L; j++) rm.mod_of[j] = (unsigned char)c.p_mod(j - level - 1);
        for (size_t x = 0; x < N; x++) {
            const long long v = round_even(coef[x] * scale);
            for (int j = 0; j < L; j++) {
                const long long q = (long long)c.T.mod[rm.mod_of[j]];
                long long r = v % q;
                if (r < 0) r += q;
                host[(size_t)j * N + x] = (u64)r;
            }
        }
        u64* d = nullptr;
        LSA_HIP(hipMalloc((void**)&d, host.size() * sizeof(u64)));
        owned.push_back(d);
        LSA_HIP(hipMemcpyAsync(d, host.data(), host.size() * sizeof(u64), hipMemcpyHostToDevice, s));
        launch_ntt(c, d, d, 1, (long long)L * N, L, rm, false, s);
        LSA_HIP(hipStreamSynchronize(s));   // `host` goes out of scope
        return d;
    }

    double evalmod_out_scale(int level_in) const {
        int level = level_in - cheb_depth;
        double sc = (double)c.T.mod[level + 1];
        for (int i = 0; i < r; i++) {
            sc = sc * sc / (double)c.T.mod[level];
            level--;
        }
        if (asin_depth) sc = (double)c.T.mod[level - asin_depth + 1];   // a polynomial evaluation ends at the scale of the prime above its result
        return sc;
    }

    void build(hipStream_t s) {
        const int n = c.n / 2;
        top_level = c.nq - 1;
        if (const char* e = std::getenv("LSA_BT_DOUBLE_HOIST")) double_hoist = e[0] != '0';
        LSA_REQUIRE(c.algo == LSA_ALGO_CKKS, "bootstrap: CKKS only");
        LSA_REQUIRE(sine_deg >= 1 && sine_deg <= 63, "bootstrap: sine degree outside 1..63");
        LSA_REQUIRE(arcsine_deg >= 0 && arcsine_deg <= 15 && (arcsine_deg == 0 || (arcsine_deg & 1)), "bootstrap: arcsine degree must be odd and at most 15");
        cheb_depth = 1;
        while ((1 << cheb_depth) < sine_deg + 1) cheb_depth++;
        asin_depth = 0;
        if (arcsine_deg > 0) {
            asin_depth = 1;
            while ((1 << asin_depth) < arcsine_deg + 1) asin_depth++;
        }
        LSA_REQUIRE(top_level - cts_depth - evalmod_depth() - stc_depth >= 0, "bootstrap: modulus chain too short");
        const std::vector<int> rg = rot_group(n);
        const double q0 = (double)c.T.mod[0];
        mul_c = std::max<long long>(1, round_even(q0 / (mr * in_scale)));
        d1 = in_scale * (double)mul_c;
        const int evalmod_level = top_level - cts_depth;
        const int stc_level = evalmod_level - evalmod_depth();
        natural_scale = evalmod_out_scale(evalmod_level) * 2.0 * kPi * d1 / q0;
        int logn_ring = 0;
        while ((1 << logn_ring) < c.n) logn_ring++;
        sparse = log_slots > 0 && log_slots < logn_ring - 1;
        const int ns = sparse ? 1 << log_slots : n;
        std::vector<Diags> mc = merged_matrices(ns, cts_depth, true), ms;
        Diags dp1, dp2;
        if (!sparse) {
            ms = merged_matrices(n, stc_depth, false);
            const double g = 1.0 / (2.0 * (double)n * (double)K);   // 1/n (inverse FFT), 1/2 (t + conj t), 1/K (unit interval)
            for (auto& kv : mc[0])
                for (auto& v : kv.second) v *= g;
        } else {
            // see oracle/ckks_bootstrap.py SparseBootstrapper: last CtS matrix M as y = P1 x + P2 conj(x) on period 2*ns
            // (first half Re(Mx), second half Im(Mx)); StC's first matrix preceded by the repack R = {0, ns}
            const int gap = n / ns;
            Diags last = mc.back();
            mc.pop_back();
            for (auto& kv : last) {
                std::vector<cplx> a(2 * ns), b(2 * ns);
                for (int t = 0; t < ns; t++) {
                    const cplx d = kv.second[t];
                    a[t] = 0.5 * d;
                    a[t + ns] = cplx(0.0, -0.5) * d;
                    b[t] = 0.5 * std::conj(d);
                    b[t + ns] = cplx(0.0, 0.5) * std::conj(d);
                }
                dp1[kv.first] = a;
                dp2[kv.first] = b;
            }
            const double g = 1.0 / ((double)ns * (double)gap * (double)K);   // 1/slots, 1/gap (SubSum), 1/K
            if (!mc.empty()) {
                for (auto& kv : mc[0])
                    for (auto& v : kv.second) v *= g;
            } else {
                for (auto* dd : {&dp1, &dp2})
                    for (auto& kv : *dd)
                        for (auto& v : kv.second) v *= g;
            }
            const std::vector<int> rgs = rot_group(ns);
            const std::vector<int> sizes = group_sizes(log_slots, stc_depth, false);
            Diags m;
            {
                std::vector<cplx> e(2 * ns), f(2 * ns);
                for (int t = 0; t < ns; t++) {
                    e[t] = 1.0;
                    e[t + ns] = cplx(0.0, 1.0);
                    f[t] = cplx(0.0, 1.0);
                    f[t + ns] = 1.0;
                }
                m[0] = e;
                m[ns] = f;
            }
            int pos = 0;
            for (size_t gi = 0; gi < sizes.size(); gi++) {
                bool have = gi == 0;
                for (int i = 0; i < sizes[gi]; i++) {
                    Diags lay = layer_diagonals(ns, 1 << (pos + i + 1), false, rgs);
                    if (gi == 0) {
                        for (auto& kv : lay) kv.second = tiled(kv.second, 2 * ns);
                        m = compose(m, lay, 2 * ns);
                    } else {
                        m = have ? compose(m, lay, ns) : lay;
                        have = true;
                    }
                }
                ms.push_back(m);
                m.clear();
                pos += sizes[gi];
            }
        }
        if (out_scale > 0) {
            const double kappa = out_scale / natural_scale;
            for (auto& kv : ms[0])
                for (auto& v : kv.second) v *= kappa;
            natural_scale = out_scale;
        }
        std::map<u64, bool> gal;
        auto gel = [&](int rot) {
            u64 e = 1;
            const u64 m = 2ULL * c.n;
            for (int i = 0; i < rot % n; i++) e = e * 5 % m;
            return e;
        };
        // one matrix -> plaintexts (diagonals of period `period`, tiled over the N/2 slots) + the rotations it needs
        auto make_one = [&](const Diags& mat, int level, int period) {
            BtMatrix bm;
            bm.level = level;
            bm.period = period;
            for (auto& kv : mat) bm.ks.push_back(kv.first);
            bm.naive = bm.ks.size() < 3;
            bm.n1 = bm.naive ? 1 : bsgs_split(bm.ks, period, 2.0);
            const double pt_scale = (double)c.T.mod[bm.level];
            const bool ext = double_hoist && !bm.naive;
            bm.rows = bm.level + 1 + (ext ? c.np : 0);
            for (int k : bm.ks) {
                const int giant = bm.naive ? 0 : (k / bm.n1) * bm.n1;
                const std::vector<cplx>& d = mat.at(k);
                std::vector<cplx> rolled(n);
                for (int t = 0; t < n; t++) rolled[t] = d[(((t - giant) % period) + period) % period];   // rot_{-giant}(diag), tiled
                bm.plains.push_back(upload_plain(slots_to_coeffs(rolled, rg), pt_scale, bm.level, s, ext));
                const int baby = bm.naive ? k : k - giant;
                if (baby) gal[gel(baby)] = true;
                if (giant) gal[gel(giant)] = true;
            }
            return bm;
        };
        for (size_t i = 0; i < mc.size(); i++) cts.push_back(make_one(mc[i], top_level - (int)i, ns));
        if (sparse) {
            p1 = make_one(dp1, top_level - (int)mc.size(), 2 * ns);
            p2 = make_one(dp2, top_level - (int)mc.size(), 2 * ns);
            for (int i = log_slots; i < logn_ring - 1; i++) gal[gel(1 << i)] = true;   // SubSum
        }
        for (size_t i = 0; i < ms.size(); i++) stc.push_back(make_one(ms[i], stc_level - (int)i, sparse && i == 0 ? 2 * ns : ns));
        gal[2ULL * c.n - 1] = true;
        for (auto& kv : gal) galois.push_back(kv.first);
        // arcsine correction (btp_eval_mod_arcsine_deg > 0): after the double-angle steps y = sin(2 pi v); arcsin(y) = 2 pi eps exactly
        // where sin only approximates it -- Taylor series sum_k C(2k,k) / (4^k (2k+1)) y^(2k+1) up to the requested degree
        asin_coef.clear();
        if (asin_depth) {
            asin_coef.assign((size_t)1 << asin_depth, 0.0);
            double binom = 1.0, pow4 = 1.0;   // C(2k, k), 4^k
            for (int k = 0; 2 * k + 1 <= arcsine_deg; k++) {
                if (k > 0) {
                    binom = binom * (double)(2 * k) * (double)(2 * k - 1) / ((double)k * (double)k);
                    pow4 *= 4.0;
                }
                asin_coef[(size_t)(2 * k + 1)] = binom / (pow4 * (double)(2 * k + 1));
            }
        }
        // Chebyshev interpolant (first-kind nodes) of cos(2 pi (K x - 1/4) / 2^r), 2^cheb_depth coefficients
        const int M = 1 << cheb_depth;
        cheb.assign(M, 0.0);
        std::vector<double> f(M), th(M);
        for (int j = 0; j < M; j++) {
            th[j] = kPi * (j + 0.5) / M;
            f[j] = std::cos(2.0 * kPi * ((double)K * std::cos(th[j]) - 0.25) / (double)(1 << r));
        }
        for (int k = 0; k < M; k++) {
            double acc = 0;
            for (int j = 0; j < M; j++) acc += f[j] * std::cos(k * th[j]);
            cheb[k] = acc * (k == 0 ? 1.0 : 2.0) / M;
        }
        // monomials +-X^(N/2)  (times +-i on the slots)
        for (int sg = 0; sg < 2; sg++) {
            std::vector<double> xn2((size_t)c.n, 0.0);
            xn2[c.n / 2] = sg == 0 ? 1.0 : -1.0;
            mono[sg] = upload_plain(xn2, 1.0, top_level, s);
        }
    }
};

// ------------------------------------------------------------------------------------------------ device evaluator
namespace {

struct DBuf {
    u64* p = nullptr;
    size_t words = 0;
    std::multimap<size_t, u64*>* pool;
    ~DBuf() { pool->emplace(words, p); }
};
struct DCt {
    std::shared_ptr<DBuf> buf;
    int level = 0;
    double scale = 0;
    u64* data() const { return buf->p; }
};

struct Eval {
    Context& c;
    Bootstrap& bt;
    hipStream_t s;
    int m;   // batch
    const Key& rlk;
    const std::map<u64, const Key*>& glk;
    std::multimap<size_t, u64*>& pool;   // released device buffers (single in-order stream: reuse is ordered)
    long long N;

    Eval(Context& c_, Bootstrap& b, hipStream_t s_, int m_, const Key& rlk_, const std::map<u64, const Key*>& g)
        : c(c_), bt(b), s(s_), m(m_), rlk(rlk_), glk(g), pool(b.pool), N(c_.n) {}
    long long stride(int level) const { return 2LL * (level + 1) * N; }
    DCt alloc(int level, double scale) {
        DCt o = alloc_words((size_t)m * stride(level));
        o.level = level;
        o.scale = scale;
        return o;
    }
    DCt alloc_words(size_t words) {
        auto b = std::make_shared<DBuf>();
        b->words = words;
        b->pool = &pool;
        auto it = pool.find(words);
        if (it != pool.end()) {
            b->p = it->second;
            pool.erase(it);
        } else {
            LSA_HIP(hipMalloc((void**)&b->p, words * sizeof(u64)));
            bt.pool_all.push_back(b->p);
        }
        return DCt{b, 0, 0.0};
    }
    // extended ciphertext [2][level+1+k][N] over Q_level u P (ops.hip, ckks_rotate_many_ext); `level` and `scale` as for the
    // ciphertext it will be divided down to
    long long stride_ext(int level) const { return 2LL * (level + 1 + c.np) * N; }
    DCt alloc_ext(int level, double scale) {
        DCt o = alloc_words((size_t)m * stride_ext(level));
        o.level = level;
        o.scale = scale;
        return o;
    }
    RowMap rm_ext(int level) const {   // both polynomials' rows of an extended ciphertext
        RowMap rm;
        rm.period = level + 1 + c.np;
        for (int j = 0; j <= level; j++) rm.mod_of[j] = (unsigned char)j;
        for (int i = 0; i < c.np; i++) rm.mod_of[level + 1 + i] = (unsigned char)c.p_mod(i);
        return rm;
    }
    RowMap rm2(int level) const {   // both polynomials' limbs
        RowMap rm;
        rm.period = level + 1;
        for (int j = 0; j <= level; j++) rm.mod_of[j] = (unsigned char)j;
        return rm;
    }
    double q(int level) const { return (double)c.T.mod[level]; }

    DCt addsub(const DCt& a, const DCt& b, EwOp op) {
        LSA_REQUIRE(a.level == b.level && std::fabs(a.scale / b.scale - 1) < 1e-9, "bootstrap: operands of add/sub differ in level or scale");
        DCt o = alloc(a.level, a.scale);
        launch_elementwise(c, op, a.data(), b.data(), o.data(), m, stride(a.level), stride(a.level), stride(a.level),
                           2 * (a.level + 1), rm2(a.level), s);
        return o;
    }
    DCt add(const DCt& a, const DCt& b) { return addsub(a, b, EW_ADD); }
    DCt sub(const DCt& a, const DCt& b) { return addsub(a, b, EW_SUB); }
    DCt rescale(const DCt& a) {
        DCt o = alloc(a.level - 1, a.scale / q(a.level));
        ckks_rescale(c, a.level, 2, a.data(), o.data(), m, stride(a.level), stride(a.level - 1), s);
        return o;
    }
    DCt mul(const DCt& a0, const DCt& b0) {
        // operands at different levels: the leading rows of each polynomial of the higher one ARE it at the lower level, the
        // tensor kernel takes the rows per polynomial -- no copy (k_copy_rows was 2 % of a bootstrap)
        const int lvl = std::min(a0.level, b0.level);
        DCt o = alloc(lvl - 1, a0.scale * b0.scale / q(lvl));
        ckks_mult_relin_rescale_rpp(c, lvl, a0.data(), b0.data(), rlk, o.data(), m, stride(a0.level), stride(b0.level), stride(lvl - 1), s,
                                    a0.level + 1, b0.level + 1);
        return o;
    }
    const Key& gkey(u64 e) const {
        auto it = glk.find(e);
        LSA_REQUIRE(it != glk.end(), "bootstrap: Galois key for element " + std::to_string(e) + " missing");
        return *it->second;
    }
    DCt rotate(const DCt& a, int r) {
        const int n = c.n / 2;
        r = ((r % n) + n) % n;
        if (r == 0) return a;
        u64 e = 1;
        for (int i = 0; i < r; i++) e = e * 5 % (2ULL * c.n);
        DCt o = alloc(a.level, a.scale);
        ckks_rotate(c, a.level, a.data(), e, gkey(e), o.data(), m, stride(a.level), stride(a.level), s);
        return o;
    }
    // rotations of one ciphertext by several steps with a single decomposition (hoisted); same residues as rotate()
    std::map<int, DCt> rotate_many(const DCt& a, const std::vector<int>& steps) {
        const int n = c.n / 2;
        std::map<int, DCt> out;
        std::vector<u64> els;
        std::vector<const Key*> keys;
        std::vector<u64*> ptrs;
        for (int r0 : steps) {
            const int r = ((r0 % n) + n) % n;
            if (out.count(r)) continue;
            if (r == 0) {
                out[0] = a;
                continue;
            }
            u64 e = 1;
            for (int i = 0; i < r; i++) e = e * 5 % (2ULL * c.n);
            DCt o = alloc(a.level, a.scale);
            els.push_back(e);
            keys.push_back(&gkey(e));
            ptrs.push_back(o.data());
            out[r] = o;
        }
        ckks_rotate_many(c, a.level, a.data(), (int)els.size(), els.data(), keys.data(), ptrs.data(), m, stride(a.level),
                         stride(a.level), s);
        return out;
    }
    u64 galois_of(int r) const {
        u64 e = 1;
        for (int i = 0; i < r; i++) e = e * 5 % (2ULL * c.n);
        return e;
    }
    // the same rotations WITHOUT their division by P: extended ciphertexts (step 0: the ciphertext times P)
    std::map<int, DCt> rotate_many_ext(const DCt& a, const std::vector<int>& steps) {
        const int n = c.n / 2;
        std::map<int, DCt> out;
        std::vector<u64> els;
        std::vector<const Key*> keys;
        std::vector<u64*> ptrs;
        for (int r0 : steps) {
            const int r = ((r0 % n) + n) % n;
            if (out.count(r)) continue;
            DCt o = alloc_ext(a.level, a.scale);
            out[r] = o;
            if (r == 0) {
                ckks_lift_ext(c, a.level, a.data(), o.data(), m, stride(a.level), stride_ext(a.level), s);
                continue;
            }
            const u64 e = galois_of(r);
            els.push_back(e);
            keys.push_back(&gkey(e));
            ptrs.push_back(o.data());
        }
        ckks_rotate_many_ext(c, a.level, a.data(), (int)els.size(), els.data(), keys.data(), ptrs.data(), m, stride(a.level),
                             stride_ext(a.level), s);
        return out;
    }
    DCt moddown(const DCt& a) {   // extended -> ciphertext; `a` is consumed
        DCt o = alloc(a.level, a.scale);
        ckks_moddown_ext(c, a.level, a.data(), o.data(), m, stride_ext(a.level), stride(a.level), s);
        return o;
    }
    DCt conj(const DCt& a) {
        const u64 e = 2ULL * c.n - 1;
        DCt o = alloc(a.level, a.scale);
        ckks_rotate(c, a.level, a.data(), e, gkey(e), o.data(), m, stride(a.level), stride(a.level), s);
        return o;
    }
    // per-limb constant vectors, cached on the context by value
    const u64* kvec(long long k, int level, bool montgomery) {
        std::vector<int> mods(level + 1);
        std::vector<u64> vals(level + 1);
        for (int j = 0; j <= level; j++) {
            mods[j] = j;
            const long long qq = (long long)c.T.mod[j];
            long long r = k % qq;
            if (r < 0) r += qq;
            vals[j] = (u64)r;
        }
        const std::string name = std::string(montgomery ? "btm" : "btr") + std::to_string(level) + "_" + std::to_string(k);
        return montgomery ? c.const_vec(name, mods, vals) : c.raw_vec(name, vals);
    }
    // level < a.level: the product at that lower level, read from a's leading rows (no copy to drop it first)
    DCt mul_int_raw(const DCt& a, long long k, double new_scale, int level = -1) {
        if (level < 0) level = a.level;
        DCt o = alloc(level, new_scale);
        unsigned char lm[LSA_MAX_PERIOD];
        for (int j = 0; j <= level; j++) lm[j] = (unsigned char)j;
        launch_sub_mul_general(c, 2, level + 1, lm, kvec(k, level, true), a.data(), stride(a.level), a.level + 1, nullptr, 0,
                               0, nullptr, 0, 0, 0, o.data(), stride(level), level + 1, m, s);
        return o;
    }
    DCt mul_int(const DCt& a, long long k) { return mul_int_raw(a, k, a.scale); }
    DCt mul_const(const DCt& a, double cst, double const_scale, int level = -1) {
        return mul_int_raw(a, round_even(cst * const_scale), a.scale * const_scale, level);
    }
    // per-row vectors over BOTH polynomials: [value for the L limbs of c0 | `second` for the L limbs of c1]
    const u64* kvec2(long long k0, long long k1, int level, bool montgomery) {
        const int L = level + 1;
        std::vector<int> mods(2 * L);
        std::vector<u64> vals(2 * L);
        for (int p = 0; p < 2; p++)
            for (int j = 0; j < L; j++) {
                mods[p * L + j] = j;
                const long long qq = (long long)c.T.mod[j];
                long long r = (p == 0 ? k0 : k1) % qq;
                if (r < 0) r += qq;
                vals[p * L + j] = (u64)r;
            }
        const std::string name = std::string(montgomery ? "b2m" : "b2r") + std::to_string(level) + "_" + std::to_string(k0) + "_" + std::to_string(k1);
        return montgomery ? c.const_vec(name, mods, vals) : c.raw_vec(name, vals);
    }
    RowMap rm_both(int level) const {
        RowMap rm;
        rm.period = 2 * (level + 1);
        for (int p = 0; p < 2; p++)
            for (int j = 0; j <= level; j++) rm.mod_of[p * (level + 1) + j] = (unsigned char)j;
        return rm;
    }
    // a * factor + cst in every slot, one pass (factor 1: plain add_const)
    DCt mul_int_add_const(const DCt& a, long long factor, double cst) {
        const long long k = round_even(cst * a.scale);
        DCt o = alloc(a.level, a.scale);
        launch_add_const(c, a.data(), stride(a.level), kvec2(k, 0, a.level, false), o.data(), stride(a.level), 2 * (a.level + 1),
                         rm_both(a.level), m, s, factor == 1 ? nullptr : kvec2(factor, factor, a.level, true));
        return o;
    }
    DCt add_const(const DCt& a, double cst) { return mul_int_add_const(a, 1, cst); }
    // every polynomial times a shared plaintext (stride 0 over the batch)
    DCt mul_plain(const DCt& a, const u64* pt, double pt_scale) {
        DCt o = alloc(a.level, a.scale * pt_scale);
        const int L = a.level + 1;
        for (int p = 0; p < 2; p++)
            launch_elementwise(c, EW_MUL, a.data() + (size_t)p * L * N, pt, o.data() + (size_t)p * L * N, m, stride(a.level), 0,
                               stride(a.level), L, rm2(a.level), s);
        return o;
    }
    DCt mul_by_i(const DCt& a, int sign) { return mul_plain_keep_scale(a, bt.mono[sign > 0 ? 0 : 1]); }
    DCt mul_plain_keep_scale(const DCt& a, const u64* pt) {
        DCt o = mul_plain(a, pt, 1.0);
        o.scale = a.scale;
        return o;
    }

    // Baby-step / giant-step with the sums kept over Q u P ("double hoisting", Lattigo v4 ckks/linear_transform.go
    // MultiplyByDiagMatrixBSGS; oracle twin: oracle/ckks_bootstrap.py linear_transform, double_hoist): the baby-step rotations
    // are gadget products without their division by P, the plaintexts carry the special primes' residues, each giant step's
    // inner sum is divided once, rotated without division into the running sum, and that sum is divided once:
    // (giant steps + 1) ModDowns instead of (baby steps + giant steps).
    DCt linear_transform_dh(const DCt& ct, const BtMatrix& mt, bool do_rescale) {
        const double pt_scale = q(ct.level);
        const int T = ct.level + 1 + c.np;
        std::vector<int> steps;
        for (int k : mt.ks) steps.push_back(k % mt.n1);
        std::map<int, DCt> babies = rotate_many_ext(ct, steps);
        std::map<int, std::vector<size_t>> by_giant;
        for (size_t i = 0; i < mt.ks.size(); i++) by_giant[(mt.ks[i] / mt.n1) * mt.n1].push_back(i);
        std::vector<int> bsteps, gsteps;
        std::vector<const u64*> cp;
        std::vector<long long> cs;
        for (auto& kv : babies) {
            bsteps.push_back(kv.first);
            cp.push_back(kv.second.data());
            cs.push_back(stride_ext(ct.level));
        }
        const int nb = (int)bsteps.size(), ng = (int)by_giant.size();
        std::vector<const u64*> pp((size_t)ng * nb, nullptr);
        std::vector<DCt> inner;
        std::vector<u64*> op;
        int gi = 0;
        for (auto& kv : by_giant) {
            for (size_t i : kv.second) {
                const int bi = (int)(std::find(bsteps.begin(), bsteps.end(), mt.ks[i] - kv.first) - bsteps.begin());
                LSA_REQUIRE(bi < nb, "bootstrap: baby step without its rotation");
                pp[(size_t)gi * nb + bi] = mt.plains[i];
            }
            inner.push_back(alloc_ext(ct.level, ct.scale * pt_scale));
            op.push_back(inner.back().data());
            gsteps.push_back(kv.first);
            gi++;
        }
        if (nb <= LSA_MACM_MAX && ng <= LSA_MACM_MAX) {
            launch_mac_plain_multi(c, nb, cp.data(), cs.data(), ng, pp.data(), op.data(), stride_ext(ct.level), m, 2, T, rm_ext(ct.level), s);
        } else {
            for (int g2 = 0; g2 < ng; g2++) {   // wide matrices: one giant step per launch, LSA_MAC_MAX_TERMS products each
                std::vector<int> bi;
                for (int b = 0; b < nb; b++)
                    if (pp[(size_t)g2 * nb + b]) bi.push_back(b);
                for (size_t i0 = 0; i0 < bi.size(); i0 += LSA_MAC_MAX_TERMS) {
                    const int cnt = (int)std::min<size_t>(LSA_MAC_MAX_TERMS, bi.size() - i0);
                    const u64* tc[LSA_MAC_MAX_TERMS];
                    const u64* tp[LSA_MAC_MAX_TERMS];
                    long long ts[LSA_MAC_MAX_TERMS], tz[LSA_MAC_MAX_TERMS];
                    for (int i = 0; i < cnt; i++) {
                        tc[i] = cp[bi[i0 + i]];
                        ts[i] = stride_ext(ct.level);
                        tp[i] = pp[(size_t)g2 * nb + bi[i0 + i]];
                        tz[i] = 0;
                    }
                    launch_mac_plain(c, cnt, tc, ts, tp, tz, i0 ? op[g2] : nullptr, stride_ext(ct.level), op[g2], stride_ext(ct.level), m, 2, T,
                                     rm_ext(ct.level), s);
                }
            }
        }
        babies.clear();
        DCt acc;
        bool have = false;
        const int n = c.n / 2;
        for (int g2 = 0; g2 < ng; g2++) {
            const int r = ((gsteps[g2] % n) + n) % n;
            if (r == 0) {
                LSA_REQUIRE(!have, "bootstrap: giant step 0 must come first");
                acc = inner[g2];
                have = true;
                continue;
            }
            DCt iq = moddown(inner[g2]);
            if (!have) acc = alloc_ext(ct.level, ct.scale * pt_scale);
            const u64 e = galois_of(r);
            ckks_rotate_ext(c, ct.level, iq.data(), e, gkey(e), acc.data(), have, m, stride(ct.level), stride_ext(ct.level), s);
            have = true;
        }
        DCt res = moddown(acc);
        return do_rescale ? rescale(res) : res;
    }

    DCt linear_transform(const DCt& ct, const BtMatrix& mt, bool do_rescale = true) {
        LSA_REQUIRE(ct.level == mt.level, "bootstrap: linear transform applied at an unexpected level");
        if (!mt.naive && mt.rows > ct.level + 1) return linear_transform_dh(ct, mt, do_rescale);
        const double pt_scale = q(ct.level);
        const int L = ct.level + 1;
        // every baby step is a rotation of the SAME ciphertext: one decomposition serves them all
        std::vector<int> steps;
        for (int k : mt.ks) steps.push_back(mt.naive ? k : k % mt.n1);
        std::map<int, DCt> babies = rotate_many(ct, steps);
        auto baby = [&](int b) -> const DCt& { return babies.at(b); };
        // sum of (shared plaintext) x (rotated ciphertext) terms, LSA_MAC_MAX_TERMS per launch
        auto mac = [&](const std::vector<std::pair<const u64*, const DCt*>>& terms) {
            DCt o = alloc(ct.level, ct.scale * pt_scale);
            for (size_t i0 = 0; i0 < terms.size(); i0 += LSA_MAC_MAX_TERMS) {
                const int cnt = (int)std::min<size_t>(LSA_MAC_MAX_TERMS, terms.size() - i0);
                const u64* cp[LSA_MAC_MAX_TERMS];
                const u64* pp[LSA_MAC_MAX_TERMS];
                long long cs[LSA_MAC_MAX_TERMS], ps[LSA_MAC_MAX_TERMS];
                for (int i = 0; i < cnt; i++) {
                    cp[i] = terms[i0 + i].second->data();
                    cs[i] = stride(ct.level);
                    pp[i] = terms[i0 + i].first;
                    ps[i] = 0;
                }
                launch_mac_plain(c, cnt, cp, cs, pp, ps, i0 ? o.data() : nullptr, stride(ct.level), o.data(), stride(ct.level), m, 2,
                                 L, rm2(ct.level), s);
            }
            return o;
        };
        DCt acc;
        bool have = false;
        if (mt.naive) {
            std::vector<std::pair<const u64*, const DCt*>> terms;
            for (size_t i = 0; i < mt.ks.size(); i++) terms.push_back({mt.plains[i], &baby(mt.ks[i])});
            acc = mac(terms);
            return do_rescale ? rescale(acc) : acc;
        }
        std::map<int, std::vector<size_t>> by_giant;
        for (size_t i = 0; i < mt.ks.size(); i++) by_giant[(mt.ks[i] / mt.n1) * mt.n1].push_back(i);
        if (babies.size() <= 8 && by_giant.size() <= 8 && by_giant.size() > 1 && !std::getenv("LSA_BT_NO_MULTI_MAC")) {
            // all inner sums in one launch: every baby-step ciphertext is read once, not once per giant step (same sums)
            std::vector<int> bsteps;
            std::vector<const u64*> cp;
            std::vector<long long> cs;
            for (auto& kv : babies) {
                bsteps.push_back(kv.first);
                cp.push_back(kv.second.data());
                cs.push_back(stride(ct.level));
            }
            const int nb = (int)bsteps.size(), ng = (int)by_giant.size();
            std::vector<const u64*> pp((size_t)ng * nb, nullptr);
            std::vector<DCt> inner;
            std::vector<u64*> op;
            std::vector<int> gsteps;
            int gi = 0;
            for (auto& kv : by_giant) {
                for (size_t i : kv.second) {
                    const int bs = mt.ks[i] - kv.first;
                    const int bi = (int)(std::find(bsteps.begin(), bsteps.end(), bs) - bsteps.begin());
                    LSA_REQUIRE(bi < nb, "bootstrap: baby step without its rotation");
                    pp[(size_t)gi * nb + bi] = mt.plains[i];
                }
                inner.push_back(alloc(ct.level, ct.scale * pt_scale));
                op.push_back(inner.back().data());
                gsteps.push_back(kv.first);
                gi++;
            }
            launch_mac_plain_multi(c, nb, cp.data(), cs.data(), ng, pp.data(), op.data(), stride(ct.level), m, 2, L, rm2(ct.level), s);
            for (int g2 = 0; g2 < ng; g2++) {
                DCt r = rotate(inner[g2], gsteps[g2]);
                acc = have ? add(acc, r) : r;
                have = true;
            }
            return do_rescale ? rescale(acc) : acc;
        }
        for (auto& kv : by_giant) {
            std::vector<std::pair<const u64*, const DCt*>> terms;
            for (size_t i : kv.second) terms.push_back({mt.plains[i], &baby(mt.ks[i] - kv.first)});
            DCt inner = rotate(mac(terms), kv.first);
            acc = have ? add(acc, inner) : inner;
            have = true;
        }
        return do_rescale ? rescale(acc) : acc;
    }

    DCt eval_chebyshev(const DCt& u, const std::vector<double>& coeffs) {
        int k = 0;
        while ((1u << k) < coeffs.size()) k++;
        std::map<int, DCt> powers;
        powers[1] = u;
        for (int j = 1; j < k; j++) {   // T_{2^j} = 2 T_{2^(j-1)}^2 - 1
            const DCt& p = powers[1 << (j - 1)];
            powers[1 << j] = mul_int_add_const(mul(p, p), 2, -1.0);
        }
        std::function<DCt(const std::vector<double>&, int, double)> rec = [&](const std::vector<double>& cf, int level_out,
                                                                              double scale_out) -> DCt {
            if (cf.size() == 2) {
                const double cs = scale_out * q(level_out + 1) / u.scale;
                DCt rr = rescale(mul_const(u, cf[1], cs, level_out + 1));
                rr.scale = scale_out;
                return add_const(rr, cf[0]);
            }
            const size_t half = cf.size() / 2;
            std::vector<double> hi(half, 0.0), lo(cf.begin(), cf.begin() + half);
            hi[0] = cf[half];
            for (size_t j = 1; j < half; j++) {   // T_{half+j} = 2 T_half T_j - T_{half-j}
                hi[j] = 2 * cf[half + j];
                lo[half - j] -= cf[half + j];
            }
            const DCt& th = powers[(int)half];   // at or above level_out + 1: mul() reads its leading rows
            DCt h = rec(hi, level_out + 1, scale_out * q(level_out + 1) / th.scale);
            DCt prod = mul(h, th);
            prod.scale = scale_out;
            return add(prod, rec(lo, level_out, scale_out));
        };
        const int level_out = u.level - k;
        return rec(coeffs, level_out, q(level_out + 1));
    }

    // sum_k coeffs[k] u^k, len(coeffs) a power of two: the same binary splitting in the monomial basis (p = hi * u^half + lo)
    DCt eval_monomial(const DCt& u, const std::vector<double>& coeffs) {
        int k = 0;
        while ((1u << k) < coeffs.size()) k++;
        std::map<int, DCt> powers;
        powers[1] = u;
        for (int j = 1; j < k; j++) {
            const DCt& p = powers[1 << (j - 1)];
            powers[1 << j] = mul(p, p);
        }
        std::function<DCt(const std::vector<double>&, int, double)> rec = [&](const std::vector<double>& cf, int level_out,
                                                                              double scale_out) -> DCt {
            if (cf.size() == 2) {
                const double cs = scale_out * q(level_out + 1) / u.scale;
                DCt rr = rescale(mul_const(u, cf[1], cs, level_out + 1));
                rr.scale = scale_out;
                return add_const(rr, cf[0]);
            }
            const size_t half = cf.size() / 2;
            std::vector<double> hi(cf.begin() + half, cf.end()), lo(cf.begin(), cf.begin() + half);
            const DCt& th = powers[(int)half];   // at or above level_out + 1: mul() reads its leading rows
            DCt h = rec(hi, level_out + 1, scale_out * q(level_out + 1) / th.scale);
            DCt prod = mul(h, th);
            prod.scale = scale_out;
            return add(prod, rec(lo, level_out, scale_out));
        };
        const int level_out = u.level - k;
        return rec(coeffs, level_out, q(level_out + 1));
    }

    DCt eval_mod(const DCt& u) {
        DCt y = eval_chebyshev(u, bt.cheb);
        for (int i = 0; i < bt.r; i++) y = mul_int_add_const(mul(y, y), 2, -1.0);
        if (!bt.asin_coef.empty()) y = eval_monomial(y, bt.asin_coef);
        return y;
    }

    // level-0 ciphertext -> the same polynomials (centred mod q_0) over Q_top
    DCt mod_raise(const DCt& a, int top) {
        DCt o = alloc(top, q(0));
        DCt tmp = alloc(0, a.scale);   // [m][2][1][N] coefficient-domain copy
        RowMap r0;
        r0.period = 1;
        r0.mod_of[0] = 0;
        launch_ntt(c, a.data(), tmp.data(), m, stride(0), stride(0), 2, r0, true, s);
        // each polynomial is one "ring-t plaintext": centred lift from q_0 to every limb, then NTT
        launch_lift_ringt(c, 0, top, tmp.data(), N, o.data(), (long long)(top + 1) * N, 2 * m, s);
        launch_ntt(c, o.data(), o.data(), m, stride(top), 2 * (top + 1), rm2(top), false, s);
        return o;
    }
};

}  // namespace

Bootstrap* bootstrap_create(Context& c, int cts_depth, int stc_depth, int K, int double_angle, double message_ratio,
                            double in_scale, double out_scale, int log_slots, hipStream_t s, int sine_deg, int arcsine_deg) {
    auto b = std::make_unique<Bootstrap>(c);
    b->sine_deg = sine_deg;
    b->arcsine_deg = arcsine_deg;
    b->log_slots = log_slots;
    b->cts_depth = cts_depth;
    b->stc_depth = stc_depth;
    b->K = K;
    b->r = double_angle;
    b->mr = message_ratio;
    b->in_scale = in_scale;
    b->out_scale = out_scale;
    c.use_device();
    b->build(s);
    return b.release();
}
void bootstrap_destroy(Bootstrap* b) { delete b; }

// in: [batch][2][1][N] level-0 ciphertexts at the plan's input scale; out: [batch][2][out_level+1][N]
void bootstrap_run(Bootstrap& bt, const u64* in, long long sin, u64* out, long long sout, int batch, const Key& rlk,
                   const std::map<u64, const Key*>& glk, const Key* swk_dts, const Key* swk_std, hipStream_t s) {
    Context& c = bt.c;
    c.use_device();
    Eval ev(c, bt, s, batch, rlk, glk);
    const long long N = c.n;
    // diagnostic: LSA_BT_STOP=<step> returns the first out_level+1 limbs of that step's intermediate instead
    const char* stop_env = getenv("LSA_BT_STOP");
    const int stop = stop_env ? atoi(stop_env) : -1;
    const int out_level = bootstrap_out_level(bt);
    int step = 0;
    auto emit = [&](const DCt& v) {
        std::vector<int> rr;
        const int lv = std::min(v.level, out_level);
        for (int p = 0; p < 2; p++)
            for (int j = 0; j <= out_level; j++) rr.push_back(p * (v.level + 1) + std::min(j, lv));
        launch_copy_rows(c, v.data(), ev.stride(v.level), out, sout, (int)rr.size(), rr.data(), batch, s);
    };
#define LSA_BT_CHECK(v)            \
    if (++step == stop) {          \
        emit(v);                   \
        return;                    \
    }
    DCt x = ev.alloc(0, bt.in_scale);
    std::vector<int> rows = {0, 1};
    launch_copy_rows(c, in, sin, x.data(), 2 * N, 2, rows.data(), batch, s);
    x = ev.mul_int(x, bt.mul_c);
    LSA_BT_CHECK(x)   // 1
    if (swk_dts) {
        DCt y = ev.alloc(0, x.scale);
        ckks_switch_key(c, 0, x.data(), *swk_dts, y.data(), batch, 2 * N, 2 * N, s);
        x = y;
    }
    x = ev.mod_raise(x, bt.top_level);
    LSA_BT_CHECK(x)   // 2
    if (swk_std) {
        DCt y = ev.alloc(bt.top_level, x.scale);
        ckks_switch_key(c, bt.top_level, x.data(), *swk_std, y.data(), batch, ev.stride(bt.top_level), ev.stride(bt.top_level), s);
        x = y;
    }
    if (bt.sparse) {   // SubSum: trace onto the subring of the sparse packing
        int logn_ring = 0;
        while ((1 << logn_ring) < c.n) logn_ring++;
        for (int i = bt.log_slots; i < logn_ring - 1; i++) x = ev.add(x, ev.rotate(x, 1 << i));
    }
    for (auto& mt : bt.cts) {
        x = ev.linear_transform(x, mt);
        LSA_BT_CHECK(x)   // 3 .. 2+cts_depth
    }
    DCt y;
    if (bt.sparse) {
        DCt a = ev.linear_transform(x, bt.p1, false);
        DCt b = ev.linear_transform(ev.conj(x), bt.p2, false);
        DCt u = ev.rescale(ev.add(a, b));          // [Re(t)/K | Im(t)/K], period 2*slots
        LSA_BT_CHECK(u)
        y = ev.eval_mod(u);
        LSA_BT_CHECK(y)
    } else {
        DCt xc = ev.conj(x);
        DCt u_re = ev.add(x, xc);
        LSA_BT_CHECK(u_re)
        DCt u_im = ev.mul_by_i(ev.sub(x, xc), -1);
        LSA_BT_CHECK(u_im)
        // both halves go through EvalMod as ONE batch of 2m ciphertexts (ciphertexts of a batch are independent: same
        // residues as two separate evaluations, half the launches -- what a single-ciphertext bootstrap is bound by)
        Eval ev2(c, bt, s, 2 * batch, rlk, glk);
        DCt uu = ev2.alloc(u_re.level, u_re.scale);
        const size_t half_bytes = (size_t)batch * ev.stride(u_re.level) * sizeof(u64);
        LSA_HIP(hipMemcpyAsync(uu.data(), u_re.data(), half_bytes, hipMemcpyDeviceToDevice, s));
        LSA_HIP(hipMemcpyAsync(uu.data() + (size_t)batch * ev.stride(u_re.level), u_im.data(), half_bytes, hipMemcpyDeviceToDevice, s));
        DCt yy = ev2.eval_mod(uu);
        DCt y_re = ev.alloc(yy.level, yy.scale), y_im = ev.alloc(yy.level, yy.scale);
        const size_t out_bytes = (size_t)batch * ev.stride(yy.level) * sizeof(u64);
        LSA_HIP(hipMemcpyAsync(y_re.data(), yy.data(), out_bytes, hipMemcpyDeviceToDevice, s));
        LSA_HIP(hipMemcpyAsync(y_im.data(), yy.data() + (size_t)batch * ev.stride(yy.level), out_bytes, hipMemcpyDeviceToDevice, s));
        LSA_BT_CHECK(y_re)
        y = ev.add(y_re, ev.mul_by_i(y_im, 1));
        LSA_BT_CHECK(y)
    }
    for (auto& mt : bt.stc) y = ev.linear_transform(y, mt);
#undef LSA_BT_CHECK
    std::vector<int> all(2 * (y.level + 1));
    for (size_t i = 0; i < all.size(); i++) all[i] = (int)i;
    launch_copy_rows(c, y.data(), ev.stride(y.level), out, sout, (int)all.size(), all.data(), batch, s);
}

int bootstrap_out_level(const Bootstrap& bt) { return bt.top_level - bt.cts_depth - bt.evalmod_depth() - bt.stc_depth; }

// read-only views for the C API (constants are exported so that the oracle can replay the program with the same integers)
double bootstrap_out_scale(const Bootstrap& bt) { return bt.natural_scale; }
const std::vector<u64>& bootstrap_galois(const Bootstrap& bt) { return bt.galois; }
const std::vector<double>& bootstrap_chebyshev(const Bootstrap& bt) { return bt.cheb; }
const std::vector<double>& bootstrap_arcsine(const Bootstrap& bt) { return bt.asin_coef; }
// matrix order: the leading CoeffsToSlots matrices, (sparse packing: P1, P2,) the SlotsToCoeffs matrices
int bootstrap_matrices(const Bootstrap& bt) { return (int)(bt.cts.size() + bt.stc.size()) + (bt.sparse ? 2 : 0); }
int bootstrap_cts_matrices(const Bootstrap& bt) { return (int)bt.cts.size(); }
bool bootstrap_is_sparse(const Bootstrap& bt) { return bt.sparse; }
void bootstrap_matrix(const Bootstrap& bt, int i, int* level, int* n1, const std::vector<int>** ks, const std::vector<u64*>** plains,
                      int* rows) {
    LSA_REQUIRE(i >= 0 && i < bootstrap_matrices(bt), "bootstrap: matrix index out of range");
    const int nc = (int)bt.cts.size(), extra = bt.sparse ? 2 : 0;
    const BtMatrix& m = i < nc ? bt.cts[i] : (i < nc + extra ? (i == nc ? bt.p1 : bt.p2) : bt.stc[i - nc - extra]);
    *level = m.level;
    *n1 = m.naive ? 0 : m.n1;
    *ks = &m.ks;
    *plains = &m.plains;
    if (rows) *rows = m.rows;
}

}  // namespace lsa
