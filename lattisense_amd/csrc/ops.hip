// ops.hip — operator pipelines: the HEArithmeticOperator surface the reference's executors call
// (mega_ag_runners/gpu/mega_ag_executors_gpu.cu:71-426), built from the kernels in kernels.hip.
//
// Algorithms follow the CPU path the results must agree with (Lattigo v4, restated in oracle/ls_oracle.c):
//   key-switch  = per-digit exact ModUp (DecomposeSingleNTT) + gadget MAC + centred ModDown (ModDownQPtoQNTT)
//   CKKS rescale = divide-and-round by the last prime (DivRoundByLastModulusNTT)
//   rotate      = key-switch c1, add c0, then apply the automorphism (Evaluator.Automorphism)
//   BFV mult    = centred extension Q->QMul, tensor in Q u QMul, round(./Q), centred return to Q, times t
#include "lsa_internal.h"

namespace lsa {

static RowMap rm_seq(int count, int first = 0) {
    RowMap r;
    LSA_REQUIRE(count >= 1 && count <= LSA_MAX_PERIOD, "row map too long");
    r.period = count;
    for (int i = 0; i < count; i++) r.mod_of[i] = (unsigned char)(first + i);
    return r;
}

static int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------------------------------------ key switch
static size_t ks_ws_rows(const Context& c, int level) {
    const int L = level + 1, T = L + c.np, beta = ceil_div(L, c.np);
    return (size_t)L + (size_t)beta * T + 2 * (size_t)T + 2 * (size_t)L;
}

// Merged tail: the key-switch result is rescaled straight away (CKKS HMult+relin+rescale).  ModDown ends with
// c_j = (acc_j - NTT(conv_j)) * P^-1 + base_j and the rescale continues with (c_j - NTT(lift_j(t))) * q_l^-1, t = INTT(c_l).
// Modular arithmetic being exact and the transforms linear, residue for residue
//   t   = (INTT(acc_l) - conv_l) * P^-1 + INTT(base_l)                      (two extra rows in transforms that run anyway)
//   out = (acc_j * P^-1 - NTT(conv_j * P^-1 + lift_j(t)) + base_j) * q_l^-1  (ONE forward transform per remaining limb; the
//                                                                            base conversion emits conv_j * P^-1 directly)
// 104 limb transforms per operation instead of 128, and 4 fewer launches.
struct KsRescale {
    u64* out;          // [2][level][N], level - 1 result
    long long sout;
};

// p[h][i] = (h < base_polys ? base[h][i] : 0) + ModDown( sum_d ModUp_d(cx) * key_d[h] )[i]   (all NTT domain)
// With `rs` (needs fused tails): p is scratch of the same shape and rs->out receives rescale(p).
static void ks_finish(Context& c, int level, const u64* cx, long long scx, const Key& key, u64* p, long long sp,
                      const u64* base, long long sbase, int base_rpp, int base_polys, int nb, u64* ws, hipStream_t s,
                      const KsRescale* rs, bool coeff_out = false, bool ext_first_pass_only = false, const u32* scatter = nullptr);

// steps 1-3: cx out of the NTT domain, every digit converted to the other limbs of Q u P, extended limbs back into the
// NTT domain (workspace layout: cxi | ext | acc | conv).  cx_coef: the same polynomial in the coefficient domain if the
// caller has it (BFV): the inverse transform is skipped.
// second_pass = false: the extended limbs are left in first-pass form for the fused second pass + key MAC (ks_finish)
static void ks_decompose(Context& c, int level, const u64* cx, long long scx, int nb, u64* ws, hipStream_t s,
                         const u64* cx_coef = nullptr, long long s_coef = 0, bool second_pass = true) {
    LSA_REQUIRE(c.np >= 1, "key switching needs at least one special prime");
    LSA_REQUIRE(level >= 0 && level < c.nq, "level out of range");
    const long long N = c.n;
    const int L = level + 1, np = c.np, T = L + np, beta = ceil_div(L, np);
    u64* cxi = ws;
    u64* ext = cxi + (size_t)nb * L * N;
    const long long s_cxi = (long long)L * N, s_ext = (long long)beta * T * N;

    // 1. cx out of the NTT domain
    const u64* conv_src = cxi;
    long long s_src = s_cxi;
    if (cx_coef) {
        conv_src = cx_coef;
        s_src = s_coef;
    } else {
        launch_ntt(c, cx, cxi, nb, scx, s_cxi, L, rm_seq(L), true, s);
    }
    // 2. per digit: exact conversion of the digit's limbs to every other limb of Q u P
    auto tl_mod = [&](int tl) { return tl < L ? tl : c.p_mod(tl - L); };
    for (int d = 0; d < beta; d++) {
        const int d0 = d * np, d1 = std::min(d0 + np, L);
        std::vector<int> src, dst;
        BaseConvRows rows{};
        for (int i = d0; i < d1; i++) {
            rows.src_row[i - d0] = i;
            src.push_back(i);
        }
        for (int tl = 0; tl < T; tl++) {
            if (tl >= d0 && tl < d1) continue;
            rows.dst_row[dst.size()] = d * T + tl;
            dst.push_back(tl_mod(tl));
        }
        launch_baseconv(c, c.baseconv(src, dst, false), rows, conv_src, ext, nb, s_src, s_ext, s);
    }
    // 3. extended limbs into the NTT domain (the digit's own limbs are taken from cx directly by the MAC)
    if (beta * T <= LSA_MAX_PERIOD) {
        RowMap rm;
        rm.period = beta * T;
        for (int d = 0; d < beta; d++)
            for (int tl = 0; tl < T; tl++) {
                const bool own = tl >= d * np && tl < std::min((d + 1) * np, L);
                rm.mod_of[d * T + tl] = own ? LSA_ROW_SKIP : (unsigned char)tl_mod(tl);
            }
        launch_ntt(c, ext, ext, nb, s_ext, s_ext, beta * T, rm, false, s, nullptr, second_pass ? 3 : 1);
        if (!second_pass && ks_fused_engines(c) != 3) {
            // the target limbs whose engine does not take the fused kernel get their second pass here (stand-alone MAC later)
            RowMap r2 = rm;
            bool any = false;
            for (int i = 0; i < beta * T; i++) {
                if (r2.mod_of[i] == LSA_ROW_SKIP) continue;
                const bool fp = c.fp64_ntt && (c.T.mod[r2.mod_of[i]] >> LSA_FP64_MAX_BITS) == 0;
                if ((ks_fused_engines(c) >> (fp ? 1 : 0)) & 1) r2.mod_of[i] = LSA_ROW_SKIP;
                else any = true;
            }
            if (any) launch_ntt(c, ext, ext, nb, s_ext, s_ext, beta * T, r2, false, s, nullptr, 2);
        }
    } else {
        for (int d = 0; d < beta; d++) {
            RowMap rm;
            rm.period = T;
            for (int tl = 0; tl < T; tl++) {
                const bool own = tl >= d * np && tl < std::min((d + 1) * np, L);
                rm.mod_of[tl] = own ? LSA_ROW_SKIP : (unsigned char)tl_mod(tl);
            }
            launch_ntt(c, ext + (size_t)d * T * N, ext + (size_t)d * T * N, nb, s_ext, s_ext, T, rm, false, s, nullptr, second_pass ? 3 : 1);
        }
    }
}

struct KsWorkspace {   // where ks_decompose / ks_mac / ks_moddown keep their intermediates inside one tile's workspace
    u64 *cxi, *ext, *acc, *conv;
    long long s_ext, s_acc, s_conv;
};
static KsWorkspace ks_layout(const Context& c, int level, int nb, u64* ws) {
    const long long N = c.n;
    const int L = level + 1, T = L + c.np, beta = ceil_div(L, c.np);
    KsWorkspace w;
    w.cxi = ws;
    w.ext = w.cxi + (size_t)nb * L * N;
    w.acc = w.ext + (size_t)nb * beta * T * N;
    w.conv = w.acc + (size_t)nb * 2 * T * N;
    w.s_ext = (long long)beta * T * N;
    w.s_acc = 2LL * T * N;
    w.s_conv = 2LL * L * N;
    return w;
}

// step 4 of a key switch on the digits that ks_decompose left in the workspace: the gadget inner product with the key (both
// halves) -> w.acc, [2][L+k][N] over Q_level u P, NTT domain; fused with the extension transform's second pass when
// ks_decompose stopped after the first one
static void ks_mac(Context& c, int level, const u64* cx, long long scx, const Key& key, int nb, u64* ws, hipStream_t s,
                   bool ext_first_pass_only) {
    const KsWorkspace w = ks_layout(c, level, nb, ws);
    if (ext_first_pass_only) {
        const int eng = ks_fused_engines(c);
        LSA_REQUIRE(launch_ntt_ksmac(c, level, cx, scx, w.ext, w.s_ext, key, w.acc, w.s_acc, nb, s, eng), "fused key MAC: shape not covered");
        for (int e = 0; e < 2; e++)
            if (!((eng >> e) & 1)) launch_ks_mac(c, level, cx, scx, w.ext, w.s_ext, key, w.acc, w.s_acc, nb, s, e);
    } else {
        launch_ks_mac(c, level, cx, scx, w.ext, w.s_ext, key, w.acc, w.s_acc, nb, s);
    }
}

static void ks_moddown(Context& c, int level, u64* acc, long long s_acc, u64* conv, u64* p, long long sp, const u64* base,
                       long long sbase, int base_rpp, int base_polys, int nb, hipStream_t s, const KsRescale* rs, bool coeff_out,
                       const u32* scatter = nullptr);

// steps 4-5 of a key switch on the digits that ks_decompose left in the workspace: they depend on the key, the decomposition
// does not -- rotations of ONE ciphertext by several Galois elements share it ("hoisting"; with the automorphism applied
// after the switch, as here, every rotation's residues are the same as if it had been computed on its own)
static void ks_finish(Context& c, int level, const u64* cx, long long scx, const Key& key, u64* p, long long sp,
                      const u64* base, long long sbase, int base_rpp, int base_polys, int nb, u64* ws, hipStream_t s,
                      const KsRescale* rs, bool coeff_out, bool ext_first_pass_only, const u32* scatter) {
    const KsWorkspace w = ks_layout(c, level, nb, ws);
    ks_mac(c, level, cx, scx, key, nb, ws, s, ext_first_pass_only);
    ks_moddown(c, level, w.acc, w.s_acc, w.conv, p, sp, base, sbase, base_rpp, base_polys, nb, s, rs, coeff_out, scatter);
}

// step 5, the division by P of a polynomial pair over Q_level u P (acc: [2][L+k][N] per batch item, NTT domain; its P rows --
// and, for the merged rescale, its last Q row -- are transformed in place), conv: 2L rows of scratch per batch item.
// scatter (fused tails only): every result row is written through the index map, p[row][scatter[x]] = value(x) -- the
// NTT-domain automorphism of a rotation applied by the last pass's store instead of a permutation kernel afterwards
static void ks_moddown(Context& c, int level, u64* acc, long long s_acc, u64* conv, u64* p, long long sp, const u64* base,
                       long long sbase, int base_rpp, int base_polys, int nb, hipStream_t s, const KsRescale* rs, bool coeff_out,
                       const u32* scatter) {
    LSA_REQUIRE(!scatter || (c.fuse_tails && !rs && !coeff_out), "scattered ModDown store: fused tails, no rescale, NTT-domain output");
    const long long N = c.n;
    const int L = level + 1, np = c.np, T = L + np;
    const long long s_conv = 2LL * L * N;
    // 5. ModDown: P-part out of NTT, centred exact conversion P -> Q, back to NTT, (accQ - conv) * P^-1 (+ base).
    //    coeff_out (BFV: the result is wanted in the coefficient domain and `base` is given there): every row of acc leaves
    //    the NTT domain once and the tail runs on coefficients -- INTT((acc - NTT(conv)) * P^-1) == (INTT(acc) - conv) * P^-1
    //    residue for residue, 2(L+k) transforms instead of 2k + 2L + 2L.
    {
        RowMap rm;
        rm.period = 2 * T;
        for (int h = 0; h < 2; h++)
            for (int tl = 0; tl < T; tl++)
                rm.mod_of[h * T + tl] = tl >= L ? (unsigned char)c.p_mod(tl - L)
                                                : (coeff_out ? (unsigned char)tl
                                                             : (rs && tl == level ? (unsigned char)level : LSA_ROW_SKIP));
        launch_ntt(c, acc, acc, nb, s_acc, 2 * T, rm, true, s);
    }
    {
        std::vector<int> src, dst;
        BaseConvRows rows{};
        for (int i = 0; i < np; i++) {
            rows.src_row[i] = L + i;
            src.push_back(c.p_mod(i));
        }
        for (int j = 0; j < L; j++) {
            rows.dst_row[j] = j;
            dst.push_back(j);
        }
        const BaseConvPlan* k = c.baseconv(src, dst, true, rs != nullptr);
        for (int h = 0; h < 2; h++)
            launch_baseconv(c, k, rows, acc + (size_t)h * T * N, conv + (size_t)h * L * N, nb, s_acc, s_conv, s);
    }
    if (coeff_out) {
        launch_moddown_final(c, level, acc, s_acc, T, conv, s_conv, base, sbase, base_rpp, base_polys, p, sp, nb, s);
        return;
    }
    if (rs) {
        LSA_REQUIRE(c.fuse_tails && level >= 1 && base && base_polys == 2, "merged ModDown+rescale: unsupported shape");
        const int l = level;
        // t[h] = (INTT(acc[h][l]) - conv[h][l]) * P^-1 + INTT(base[h][l]) -> p[h][l].  base is the caller's scratch here
        // (the tensor output): its last limbs are transformed in place, nothing reads them in NTT form afterwards.
        u64* base_rw = const_cast<u64*>(base);
        RowMap rb;
        rb.period = 1;
        rb.mod_of[0] = (unsigned char)l;
        rb.row0 = l;
        rb.row_step = base_rpp;
        launch_ntt(c, base_rw, base_rw, nb, sbase, sbase, 2, rb, true, s);
        const unsigned char lm[1] = {(unsigned char)l};
        launch_sub_mul_general(c, 2, 1, lm, c.pinv_vec(level) + l, acc + (long long)l * N, s_acc, T, conv + (long long)l * N,
                               s_conv, L, base + (long long)l * N, sbase, base_rpp, 2, p + (long long)l * N, sp, L, nb, s);
        // every other limb: in = conv_j*P^-1 + lift_j(t), out = (acc_j*P^-1 - NTT(in) + base_j) * q_l^-1
        RowMap rmo;
        rmo.period = 2 * L;
        for (int h = 0; h < 2; h++)
            for (int j = 0; j < L; j++) rmo.mod_of[h * L + j] = j == l ? LSA_ROW_SKIP : (unsigned char)j;
        NttFusion fb;
        fb.pro = 2;
        fb.epi = 2;
        fb.limbs = L;
        fb.ql_mod = l;
        fb.last = p + (long long)l * N;
        fb.last_stride = sp;
        fb.last_rpp = L;
        fb.a = acc;
        fb.a_stride = s_acc;
        fb.a_rpp = T;
        fb.base = base;
        fb.base_stride = sbase;
        fb.base_rpp = base_rpp;
        fb.base_polys = base_polys;
        fb.k = c.pinv_vec(level);
        fb.k2 = c.qlinv_vec(level);
        fb.out = rs->out;
        fb.out_stride = rs->sout;
        fb.out_rpp = level;
        launch_ntt(c, conv, conv, nb, s_conv, s_conv, 2 * L, rmo, false, s, &fb);
    } else if (c.fuse_tails) {
        // forward NTT of conv with the ModDown tail fused into its last-pass store: the transformed conv is consumed in
        // registers ((acc_Q - conv) * P^-1 + base) and never written
        NttFusion fz;
        fz.epi = 1;
        fz.limbs = L;
        fz.a = acc;
        fz.a_stride = s_acc;
        fz.a_rpp = T;
        fz.base = base;
        fz.base_stride = sbase;
        fz.base_rpp = base_rpp;
        fz.base_polys = base_polys;
        fz.k = c.pinv_vec(level);
        fz.out = p;
        fz.out_stride = sp;
        fz.out_rpp = L;
        fz.scatter = scatter;
        launch_ntt(c, conv, conv, nb, s_conv, s_conv, 2 * L, rm_seq(L), false, s, &fz);
    } else {
        launch_ntt(c, conv, conv, nb, s_conv, 2 * L, rm_seq(L), false, s);
        launch_moddown_final(c, level, acc, s_acc, T, conv, s_conv, base, sbase, base_rpp, base_polys, p, sp, nb, s);
    }
}

// one key per decomposition (relinearisation, a single rotation, a generic switch): the extension transform's second pass
// and the key MAC run as one kernel where the shape allows (k_ntt_r16_ksmac); hoisted rotations (several keys on one
// decomposition) keep the two steps apart
static bool ks_fuse_mac(const Context& c, int level, const Key& key) {
    const int T = level + 1 + c.np, beta = ceil_div(level + 1, c.np);
    if (!ks_fused_enabled(c) || !ks_fused_engines(c) || (c.fp64_ntt && !key.fp) || beta * T > LSA_MAX_PERIOD || T > 64) return false;
    for (int tl = 0; tl < T; tl++) {   // worth it only if some target limb takes the fused kernel
        const u64 q = c.T.mod[tl <= level ? tl : c.p_mod(tl - level - 1)];
        const bool fp = c.fp64_ntt && (q >> LSA_FP64_MAX_BITS) == 0;
        if ((ks_fused_engines(c) >> (fp ? 1 : 0)) & 1) return true;
    }
    return false;
}
static void key_switch(Context& c, int level, const u64* cx, long long scx, const Key& key, u64* p, long long sp,
                       const u64* base, long long sbase, int base_rpp, int base_polys, int nb, u64* ws, hipStream_t s,
                       const KsRescale* rs = nullptr, const u32* scatter = nullptr) {
    const bool fuse = ks_fuse_mac(c, level, key);
    ks_decompose(c, level, cx, scx, nb, ws, s, nullptr, 0, !fuse);
    ks_finish(c, level, cx, scx, key, p, sp, base, sbase, base_rpp, base_polys, nb, ws, s, rs, false, fuse, scatter);
}

// the automorphism X -> X^g of a rotation as the SCATTER map of the key switch's last store: out[i] = in[perm_g[i]] is
// out[perm_{g^-1}[x]] = in[x] (the maps of g and g^-1 are inverse permutations).  Null when the store cannot take it
// (unfused tails, LSA_ROT_SCATTER=0): the caller then permutes afterwards.
static bool rotation_scatter_on() {   // read per call (not cached): the parity tests flip it inside one process
    const char* e = std::getenv("LSA_ROT_SCATTER");
    return !(e && e[0] == '0');
}
static const u32* inverse_perm(Context& c, u64 g) {
    const u64 mask = 2 * (u64)c.n - 1;
    u64 inv = g;   // Newton iteration for the inverse modulo a power of two: doubles the correct low bits each step
    for (int i = 0; i < 6; i++) inv = (inv * (2 - g * inv)) & mask;
    LSA_REQUIRE(((inv * g) & mask) == 1, "Galois element without an inverse");
    return c.ntt_perm(inv);
}
static const u32* rotation_scatter(Context& c, u64 g) {
    return rotation_scatter_on() && c.fuse_tails ? inverse_perm(c, g) : nullptr;
}

// ------------------------------------------------------------------------------------------------ rescale
static size_t rescale_ws_rows(int level, int polys) { return (size_t)polys * (1 + level); }

static void rescale(Context& c, int level, int polys, const u64* in, long long sin, u64* out, long long sout, int nb,
                    bool ntt_domain, u64* ws, hipStream_t s) {
    LSA_REQUIRE(level >= 1 && level < c.nq, "rescale needs level >= 1");
    const long long N = c.n;
    const int L = level + 1;
    u64* last = ws;
    u64* tmp = last + (size_t)nb * polys * N;
    const long long s_last = (long long)polys * N, s_tmp = (long long)polys * level * N;
    std::vector<int> rows(polys);
    for (int p = 0; p < polys; p++) rows[p] = p * L + level;
    launch_copy_rows(c, in, sin, last, s_last, polys, rows.data(), nb, s);
    if (ntt_domain) {
        RowMap rm;
        rm.period = 1;
        rm.mod_of[0] = (unsigned char)level;
        launch_ntt(c, last, last, nb, s_last, polys, rm, true, s);
    }
    if (ntt_domain && c.fuse_tails) {
        // one forward NTT over the level limbs of every polynomial: its first-pass load derives the tile from the last limb
        // (centred remainder, reduced to the target prime), its last-pass store applies (c - t) * q_l^-1 -> out
        NttFusion fz;
        fz.pro = 1;
        fz.epi = 1;
        fz.limbs = level;
        fz.ql_mod = level;
        fz.last = last;
        fz.last_stride = s_last;
        fz.a = in;
        fz.a_stride = sin;
        fz.a_rpp = L;
        fz.k = c.qlinv_vec(level);
        fz.out = out;
        fz.out_stride = sout;
        fz.out_rpp = level;
        launch_ntt(c, tmp, tmp, nb, s_tmp, s_tmp, polys * level, rm_seq(level), false, s, &fz);
        return;
    }
    launch_rescale_prep(c, level, polys, last, s_last, tmp, s_tmp, nb, s);
    if (ntt_domain) launch_ntt(c, tmp, tmp, nb, s_tmp, polys * level, rm_seq(level), false, s);
    launch_rescale_final(c, level, polys, in, sin, tmp, s_tmp, out, sout, nb, s);
}

// ------------------------------------------------------------------------------------------------ tiling
static int pick_tile(const Context& c, size_t rows_per_ct, int batch) {
    if (c.tile_batch > 0) return std::min(c.tile_batch, batch);
    // measured on MI355X (profiles/r01/tile_sweep*.log): launches need >= ~2k workgroups each to fill 256 CUs (~16
    // ciphertexts per wave at N=2^16); past that the gain is the shrinking tail of each launch: +3 % (HMult) / +8 %
    // (rotate) from 19 to 64 ciphertexts, for 6.6 GiB of workspace out of 288.
    const size_t bytes_per_ct = rows_per_ct * (size_t)c.n * sizeof(u64);
    size_t tb = (8ull << 30) / std::max<size_t>(bytes_per_ct, 1);
    if (tb < 1) tb = 1;
    // the cap scales with 1/N (same work per launch): 64 at N=2^16, 256 at N=2^14 (BFV mult+relin +12 % over 64)
    const size_t cap = std::min<size_t>(512, std::max<size_t>(64, (64ull << 16) / (size_t)c.n));
    if (tb > cap) tb = cap;
    return (int)std::min<size_t>(tb, (size_t)batch);
}

// Runs fn(nb, b0, ws, tb, stream) for every tile of `tb` ciphertexts.  Tiles alternate between the caller's stream and
// the context's auxiliary stream (each with its own workspace half): every kernel of the pipeline uses only part of the
// chip's VALU / HBM / latency budget, so two independent tiles in flight fill each other's gaps.
template <typename F>
static void for_tiles(Context& c, size_t rows_per_ct, int batch, hipStream_t s, F&& fn) {
    if (batch <= 0) return;
    const int tb = pick_tile(c, rows_per_ct, batch);
    const int ntiles = ceil_div(batch, tb);
    const bool dual = c.dual_stream && ntiles >= 2;
    const size_t words = rows_per_ct * (size_t)c.n * tb;
    u64* ws = c.workspace(words * (dual ? 2 : 1), s);
    if (dual) c.fork_aux(s);
    for (int t = 0; t < ntiles; t++) {
        const int b0 = t * tb, nb = std::min(tb, batch - b0);
        const bool on_aux = dual && (t & 1);
        fn(nb, b0, ws + (on_aux ? words : 0), tb, on_aux ? c.aux_stream : s);
    }
    if (dual) c.join_aux(s);
}

// ================================================================================================ CKKS
void ckks_mult(Context& c, int level, const u64* a, const u64* b, u64* d3, int batch, long long sa, long long sb,
               long long sd, hipStream_t s) {
    LSA_REQUIRE(level >= 0 && level < c.nq, "level out of range");
    launch_tensor(c, a, b, d3, batch, sa, sb, sd, level + 1, rm_seq(level + 1), s);
}

void ckks_relin(Context& c, int level, const u64* d3, const Key& rlk, u64* out, int batch, long long sd, long long so,
                hipStream_t s) {
    const long long N = c.n;
    const int L = level + 1;
    for_tiles(c, ks_ws_rows(c, level), batch, s, [&](int nb, int b0, u64* ws, int, hipStream_t st) {
        const u64* d = d3 + (size_t)b0 * sd;
        key_switch(c, level, d + 2LL * L * N, sd, rlk, out + (size_t)b0 * so, so, d, sd, L, 2, nb, ws, st);
    });
}

void ckks_rescale(Context& c, int level, int polys, const u64* in, u64* out, int batch, long long sin, long long sout,
                  hipStream_t s) {
    for_tiles(c, rescale_ws_rows(level, polys), batch, s, [&](int nb, int b0, u64* ws, int, hipStream_t st) {
        rescale(c, level, polys, in + (size_t)b0 * sin, sin, out + (size_t)b0 * sout, sout, nb, true, ws, st);
    });
}

void ckks_rotate(Context& c, int level, const u64* in, u64 g, const Key& glk, u64* out, int batch, long long sin,
                 long long sout, hipStream_t s) {
    const long long N = c.n;
    const int L = level + 1;
    const size_t ks_rows = ks_ws_rows(c, level);
    const long long sp = 2LL * L * N;
    // (an in-place rotation keeps the two-step form: the tail reads c0 from `in` while other workgroups already store)
    const bool apart = out + (size_t)batch * sout <= in || in + (size_t)batch * sin <= out;
    if (const u32* scatter = apart ? rotation_scatter(c, g) : nullptr) {
        // the permutation rides on the ModDown tail's store: no intermediate, no permutation kernel (2L reads + 2L writes less)
        for_tiles(c, ks_rows, batch, s, [&](int nb, int b0, u64* ws, int, hipStream_t st) {
            const u64* ct = in + (size_t)b0 * sin;
            key_switch(c, level, ct + (long long)L * N, sin, glk, out + (size_t)b0 * sout, sout, ct, sin, L, 1, nb, ws, st, nullptr, scatter);
        });
        return;
    }
    const u32* perm = c.ntt_perm(g);
    for_tiles(c, ks_rows + 2 * (size_t)L, batch, s, [&](int nb, int b0, u64* ws, int tb, hipStream_t st) {
        u64* p = ws + ks_rows * N * tb;
        const u64* ct = in + (size_t)b0 * sin;
        key_switch(c, level, ct + (long long)L * N, sin, glk, p, sp, ct, sin, L, 1, nb, ws, st);
        launch_permute_ntt(c, perm, p, sp, out + (size_t)b0 * sout, sout, 2 * L, nb, st);
    });
}

// rotations of the same ciphertexts by several Galois elements with ONE decomposition (hoisting); outs[i] = rotate(in, g[i]),
// each identical to ckks_rotate's result
void ckks_rotate_many(Context& c, int level, const u64* in, int n_rot, const u64* g, const Key* const* glk, u64* const* outs,
                      int batch, long long sin, long long sout, hipStream_t s) {
    if (n_rot <= 0) return;
    const long long N = c.n;
    const int L = level + 1;
    const size_t ks_rows = ks_ws_rows(c, level);
    const long long sp = 2LL * L * N;
    std::vector<const u32*> perms(n_rot), scatters(n_rot);
    for (int i = 0; i < n_rot; i++) {
        const bool apart = outs[i] + (size_t)batch * sout <= in || in + (size_t)batch * sin <= outs[i];
        scatters[i] = apart ? rotation_scatter(c, g[i]) : nullptr;
        perms[i] = scatters[i] ? nullptr : c.ntt_perm(g[i]);
    }
    for_tiles(c, ks_rows + 2 * (size_t)L, batch, s, [&](int nb, int b0, u64* ws, int tb, hipStream_t st) {
        u64* p = ws + ks_rows * N * tb;
        const u64* ct = in + (size_t)b0 * sin;
        ks_decompose(c, level, ct + (long long)L * N, sin, nb, ws, st);
        for (int i = 0; i < n_rot; i++) {
            if (scatters[i]) {   // the permutation rides on the ModDown tail's store
                ks_finish(c, level, ct + (long long)L * N, sin, *glk[i], outs[i] + (size_t)b0 * sout, sout, ct, sin, L, 1, nb, ws, st, nullptr,
                          false, false, scatters[i]);
                continue;
            }
            ks_finish(c, level, ct + (long long)L * N, sin, *glk[i], p, sp, ct, sin, L, 1, nb, ws, st, nullptr);
            launch_permute_ntt(c, perms[i], p, sp, outs[i] + (size_t)b0 * sout, sout, 2 * L, nb, st);
        }
    });
}

// ---- extended ciphertexts: (c0, c1) times P over Q_level u P, [2][L+k][N] in the NTT domain -- what a key switch holds before
// its division by P.  Sums of them are exact, so a baby-step / giant-step linear transform divides once per giant step and
// once at the end instead of once per rotation ("double hoisting": Lattigo v4 ckks/linear_transform.go
// MultiplyByDiagMatrixBSGS over rlwe GadgetProductNoModDown / ModDownQPtoQNTT; bootstrap.hip, Eval::linear_transform).
void ckks_lift_ext(Context& c, int level, const u64* in, u64* out, int batch, long long sin, long long sout, hipStream_t s) {
    launch_permute_ext(c, level, nullptr, nullptr, 0, in, sin, 2, out, sout, false, batch, s);
}

// outs[i] = automorphism_g[i]( (P c0 + ks0, ks1) ), ks = gadget product of c1 with glk[i]; one decomposition for all
void ckks_rotate_many_ext(Context& c, int level, const u64* in, int n_rot, const u64* g, const Key* const* glk, u64* const* outs,
                          int batch, long long sin, long long sout, hipStream_t s) {
    if (n_rot <= 0) return;
    const long long N = c.n;
    const int L = level + 1;
    const bool one_pass = rotation_scatter_on();   // the MAC writes the rotated extended ciphertext itself (LSA_ROT_SCATTER=0: MAC, then k_permute_ext)
    std::vector<const u32*> perms(n_rot);
    for (int i = 0; i < n_rot; i++) perms[i] = one_pass ? inverse_perm(c, g[i]) : c.ntt_perm(g[i]);
    for_tiles(c, ks_ws_rows(c, level), batch, s, [&](int nb, int b0, u64* ws, int, hipStream_t st) {
        const u64* ct = in + (size_t)b0 * sin;
        const KsWorkspace w = ks_layout(c, level, nb, ws);
        ks_decompose(c, level, ct + (long long)L * N, sin, nb, ws, st);
        for (int i = 0; i < n_rot; i++) {
            if (one_pass) {
                launch_ks_mac(c, level, ct + (long long)L * N, sin, w.ext, w.s_ext, *glk[i], outs[i] + (size_t)b0 * sout, sout, nb, st, -1,
                              perms[i], ct, sin);
                continue;
            }
            ks_mac(c, level, ct + (long long)L * N, sin, *glk[i], nb, ws, st, false);
            launch_permute_ext(c, level, perms[i], w.acc, w.s_acc, ct, sin, 1, outs[i] + (size_t)b0 * sout, sout, false, nb, st);
        }
    });
}

// out (+)= automorphism_g( (P c0 + ks0, ks1) ): one rotation without its division by P, optionally added to `out`
void ckks_rotate_ext(Context& c, int level, const u64* in, u64 g, const Key& glk, u64* out, bool accumulate, int batch,
                     long long sin, long long sout, hipStream_t s) {
    const long long N = c.n;
    const int L = level + 1;
    const u32* perm = c.ntt_perm(g);
    const bool fuse = ks_fuse_mac(c, level, glk);
    for_tiles(c, ks_ws_rows(c, level), batch, s, [&](int nb, int b0, u64* ws, int, hipStream_t st) {
        const u64* ct = in + (size_t)b0 * sin;
        const KsWorkspace w = ks_layout(c, level, nb, ws);
        ks_decompose(c, level, ct + (long long)L * N, sin, nb, ws, st, nullptr, 0, !fuse);
        ks_mac(c, level, ct + (long long)L * N, sin, glk, nb, ws, st, fuse);
        launch_permute_ext(c, level, perm, w.acc, w.s_acc, ct, sin, 1, out + (size_t)b0 * sout, sout, accumulate, nb, st);
    });
}

// the rounded division by P: extended ciphertext -> ciphertext [2][L][N].  `in` is clobbered (its P rows leave the NTT domain).
void ckks_moddown_ext(Context& c, int level, u64* in, u64* out, int batch, long long sin, long long sout, hipStream_t s) {
    const int L = level + 1;
    for_tiles(c, 2 * (size_t)L, batch, s, [&](int nb, int b0, u64* ws, int, hipStream_t st) {
        ks_moddown(c, level, in + (size_t)b0 * sin, sin, ws, out + (size_t)b0 * sout, sout, nullptr, 0, 0, 0, nb, st, nullptr, false);
    });
}

// (c0, c1) under s_in -> (c0 + ks0, ks1) under s_out with a generic switching key (bootstrapping's sparse-secret
// encapsulation keys swk_dts / swk_std, reference: custom_task.py:1989-1996): the rotation pipeline without the permutation
void ckks_switch_key(Context& c, int level, const u64* in, const Key& swk, u64* out, int batch, long long sin, long long sout,
                     hipStream_t s) {
    const long long N = c.n;
    const int L = level + 1;
    for_tiles(c, ks_ws_rows(c, level), batch, s, [&](int nb, int b0, u64* ws, int, hipStream_t st) {
        const u64* ct = in + (size_t)b0 * sin;
        key_switch(c, level, ct + (long long)L * N, sin, swk, out + (size_t)b0 * sout, sout, ct, sin, L, 1, nb, ws, st);
    });
}

void ckks_mult_relin_rescale_rpp(Context& c, int level, const u64* a, const u64* b, const Key& rlk, u64* out, int batch,
                                 long long sa, long long sb, long long so, hipStream_t s, int a_rpp, int b_rpp);
void ckks_mult_relin_rescale(Context& c, int level, const u64* a, const u64* b, const Key& rlk, u64* out, int batch,
                             long long sa, long long sb, long long so, hipStream_t s) {
    ckks_mult_relin_rescale_rpp(c, level, a, b, rlk, out, batch, sa, sb, so, s, 0, 0);
}
// a_rpp / b_rpp: rows per polynomial of a / b when an operand sits at a higher level than `level` (0: level + 1) -- its leading
// rows ARE the operand at this level, so callers with operands at mixed levels (polynomial evaluation) need no copies
void ckks_mult_relin_rescale_rpp(Context& c, int level, const u64* a, const u64* b, const Key& rlk, u64* out, int batch,
                                 long long sa, long long sb, long long so, hipStream_t s, int a_rpp, int b_rpp) {
    LSA_REQUIRE(level >= 1, "mult+relin+rescale needs level >= 1");
    const long long N = c.n;
    const int L = level + 1;
    const size_t r_d3 = 3 * (size_t)L, r_r2 = 2 * (size_t)L;
    const size_t r_shared = std::max(ks_ws_rows(c, level), rescale_ws_rows(level, 2));
    const long long sd = 3LL * L * N, sr = 2LL * L * N;
    for_tiles(c, r_d3 + r_r2 + r_shared, batch, s, [&](int nb, int b0, u64* ws, int tb, hipStream_t st) {
        u64* d3 = ws;
        u64* r2 = d3 + r_d3 * N * tb;
        u64* sub = r2 + r_r2 * N * tb;
        launch_tensor(c, a + (size_t)b0 * sa, b + (size_t)b0 * sb, d3, nb, sa, sb, sd, L, rm_seq(L), st, a_rpp, b_rpp);
        if (c.fuse_tails) {
            const KsRescale rs{out + (size_t)b0 * so, so};
            key_switch(c, level, d3 + 2LL * L * N, sd, rlk, r2, sr, d3, sd, L, 2, nb, sub, st, &rs);
            return;
        }
        key_switch(c, level, d3 + 2LL * L * N, sd, rlk, r2, sr, d3, sd, L, 2, nb, sub, st);
        rescale(c, level, 2, r2, sr, out + (size_t)b0 * so, so, nb, true, sub, st);
    });
}

void drop_level(Context& c, int level, int polys, const u64* in, u64* out, int batch, long long sin, long long sout,
                hipStream_t s) {
    LSA_REQUIRE(level >= 1, "drop_level needs level >= 1");
    std::vector<int> rows;
    for (int p = 0; p < polys; p++)
        for (int i = 0; i < level; i++) rows.push_back(p * (level + 1) + i);
    launch_copy_rows(c, in, sin, out, sout, (int)rows.size(), rows.data(), batch, s);
}

void poly_addsub(Context& c, int op, int level, int polys, const u64* a, const u64* b, u64* out, int batch, long long sa,
                 long long sb, long long so, hipStream_t s) {
    LSA_REQUIRE(op >= 0 && op <= 2, "op must be 0 add, 1 sub, 2 neg");
    LSA_REQUIRE(level >= 0 && level < c.nq, "level out of range");
    launch_elementwise(c, (EwOp)op, a, b, out, batch, sa, sb, so, polys * (level + 1), rm_seq(level + 1), s);
}

// ================================================================================================ BFV
static int bfv_aux_limbs(const Context& c, int level) { return bfv_aux_count(c.T.mod.data(), level + 1, c.logn); }

void bfv_mult(Context& c, int level, const u64* a, const u64* b, u64* d3, int batch, long long sa, long long sb,
              long long sd, hipStream_t s0) {
    LSA_REQUIRE(c.algo == LSA_ALGO_BFV, "context is not BFV");
    LSA_REQUIRE(level >= 0 && level < c.nq, "level out of range");
    const long long N = c.n;
    const int L = level + 1, M = bfv_aux_limbs(c, level), T2 = L + M;
    LSA_REQUIRE(M <= c.nmul, "auxiliary basis too small");
    const size_t rows = 2 * 2 * (size_t)T2 + 3 * (size_t)T2 + 3 * (size_t)M;
    const long long s_e = 2LL * T2 * N, s_d = 3LL * T2 * N, s_x = 3LL * M * N;
    std::vector<int> qmods, amods;
    RowMap rmT;
    rmT.period = T2;
    for (int i = 0; i < L; i++) {
        qmods.push_back(i);
        rmT.mod_of[i] = (unsigned char)i;
    }
    for (int i = 0; i < M; i++) {
        amods.push_back(c.aux_mod(i));
        rmT.mod_of[L + i] = (unsigned char)c.aux_mod(i);
    }
    // folded (default; LSA_BFV_FOLD=0: the separate element-wise steps): the Q limbs are transformed straight from the operands into
    // the extended buffer (no copy), and the two element-wise steps around the last conversion -- (aux - ext) * Q^-1 before it,
    // * t after it -- live in its source load and its constants (Context::BaseConvFold)
    const char* fold_env = std::getenv("LSA_BFV_FOLD");   // read per call: the parity tests flip it inside one process
    const bool fold_on = !(fold_env && fold_env[0] == '0');
    const BaseConvPlan* kQA = c.baseconv(qmods, amods, true);
    const BaseConvPlan* kAQ = c.baseconv(amods, qmods, true);
    BaseConvRows rQA{}, rAQ{};
    for (int i = 0; i < L; i++) rQA.src_row[i] = i, rAQ.dst_row[i] = i;
    for (int i = 0; i < M; i++) rQA.dst_row[i] = i, rAQ.src_row[i] = i;
    std::vector<int> cp(2 * L);
    // Q^-1 mod aux_i and t mod q_i
    std::vector<u64> qinv(M), tq(L);
    unsigned char lmA[LSA_MAX_PERIOD], lmQ[LSA_MAX_PERIOD];
    for (int i = 0; i < M; i++) {
        const u64 p = c.T.mod[c.aux_mod(i)];
        u64 pr = 1;
        for (int l = 0; l < L; l++) pr = mul_mod_host(pr, c.T.mod[l] % p, p);
        qinv[i] = inv_mod(pr, p);
        lmA[i] = (unsigned char)c.aux_mod(i);
    }
    for (int i = 0; i < L; i++) {
        tq[i] = c.t % c.T.mod[i];
        lmQ[i] = (unsigned char)i;
    }
    const u64* kQinv = c.const_vec("bfv_qinv" + std::to_string(L), amods, qinv);
    const u64* kT = c.const_vec("bfv_t" + std::to_string(L), qmods, tq);
    Context::BaseConvFold fold{"bfv_mul" + std::to_string(L), qinv, tq};
    const BaseConvPlan* kAQf = fold_on ? c.baseconv(amods, qmods, true, false, &fold) : nullptr;
    std::vector<int> sub_rows(M);
    for (int i = 0; i < M; i++) sub_rows[i] = i;
    RowMap rmAux = rmT;   // the auxiliary rows only
    for (int i = 0; i < L; i++) rmAux.mod_of[i] = LSA_ROW_SKIP;

    for_tiles(c, rows, batch, s0, [&](int nb, int b0, u64* ws, int tb, hipStream_t s) {
        u64* ea = ws;
        u64* eb = ea + (size_t)tb * 2 * T2 * N;
        u64* d = eb + (size_t)tb * 2 * T2 * N;
        u64* ext = d + (size_t)tb * 3 * T2 * N;
        const u64* srcs[2] = {a + (size_t)b0 * sa, b + (size_t)b0 * sb};
        const long long ss[2] = {sa, sb};
        u64* es[2] = {ea, eb};
        const int nops = (a == b && sa == sb) ? 1 : 2;
        for (int o = 0; o < nops; o++) {
            for (int p = 0; p < 2; p++) {
                // Q limbs copied (folded: transformed from where they are), aux limbs by centred exact extension
                std::vector<int> rr(L);
                for (int i = 0; i < L; i++) rr[i] = p * L + i;
                if (fold_on) launch_ntt(c, srcs[o] + (size_t)p * L * N, es[o] + (size_t)p * T2 * N, nb, ss[o], s_e, L, rm_seq(L), false, s);
                else launch_copy_rows(c, srcs[o], ss[o], es[o] + (size_t)p * T2 * N, s_e, L, rr.data(), nb, s);
                launch_baseconv(c, kQA, rQA, srcs[o] + (size_t)p * L * N, es[o] + ((size_t)p * T2 + L) * N, nb, ss[o],
                                s_e, s);
            }
            launch_ntt(c, es[o], es[o], nb, s_e, 2 * T2, fold_on ? rmAux : rmT, false, s);
        }
        launch_tensor(c, ea, nops == 1 ? ea : eb, d, nb, s_e, s_e, s_d, T2, rmT, s);
        launch_ntt(c, d, d, nb, s_d, 3 * T2, rmT, true, s);
        for (int k = 0; k < 3; k++)
            launch_baseconv(c, kQA, rQA, d + (size_t)k * T2 * N, ext + (size_t)k * M * N, nb, s_d, s_x, s);
        u64* o3 = d3 + (size_t)b0 * sd;
        if (fold_on) {
            // out = t * conv_{A->Q}((aux - ext) * Q^-1): the subtraction on the conversion's source load, both factors in its constants
            for (int k = 0; k < 3; k++)
                launch_baseconv(c, kAQf, rAQ, d + ((size_t)k * T2 + L) * N, o3 + (size_t)k * L * N, nb, s_d, sd, s, ext + (size_t)k * M * N,
                                s_x, sub_rows.data());
            return;
        }
        // aux part <- (aux - ext) * Q^-1     (= round(d/Q) in basis QMul)
        launch_sub_mul_general(c, 3, M, lmA, kQinv, d + (size_t)L * N, s_d, T2, ext, s_x, M, nullptr, 0, 0, 0,
                               d + (size_t)L * N, s_d, T2, nb, s);
        for (int k = 0; k < 3; k++)
            launch_baseconv(c, kAQ, rAQ, d + ((size_t)k * T2 + L) * N, o3 + (size_t)k * L * N, nb, s_d, sd, s);
        launch_sub_mul_general(c, 3, L, lmQ, kT, o3, sd, L, nullptr, 0, 0, nullptr, 0, 0, 0, o3, sd, L, nb, s);
    });
}

// key switch of a coefficient-domain polynomial: NTT in, INTT out
// cx in the coefficient domain; p[h] = (h < base_polys ? base[h] : 0) + KeySwitch(cx)[h], all in the coefficient domain.
// Only the MAC's own-digit operand needs cx in the NTT domain; the decomposition starts from the coefficients the caller
// already has and the ModDown tail runs on coefficients (ks_finish, coeff_out).
static void bfv_key_switch(Context& c, int level, const u64* cx, long long scx, const Key& key, u64* p, long long sp,
                           const u64* base, long long sbase, int base_rpp, int base_polys, int nb, u64* ws, hipStream_t s) {
    const long long N = c.n;
    const int L = level + 1;
    u64* cxn = ws;
    u64* sub = ws + (size_t)nb * L * N;
    launch_ntt(c, cx, cxn, nb, scx, (long long)L * N, L, rm_seq(L), false, s);
    const bool fuse = ks_fuse_mac(c, level, key);
    ks_decompose(c, level, cxn, (long long)L * N, nb, sub, s, cx, scx, !fuse);
    ks_finish(c, level, cxn, (long long)L * N, key, p, sp, base, sbase, base_rpp, base_polys, nb, sub, s, nullptr, true, fuse);
}

void bfv_relin(Context& c, int level, const u64* d3, const Key& rlk, u64* out, int batch, long long sd, long long so,
               hipStream_t s) {
    const long long N = c.n;
    const int L = level + 1;
    const size_t ks_rows = ks_ws_rows(c, level) + L;
    for_tiles(c, ks_rows, batch, s, [&](int nb, int b0, u64* ws, int, hipStream_t st) {
        const u64* d = d3 + (size_t)b0 * sd;
        bfv_key_switch(c, level, d + 2LL * L * N, sd, rlk, out + (size_t)b0 * so, so, d, sd, L, 2, nb, ws, st);
    });
}

void bfv_rotate(Context& c, int level, const u64* in, u64 g, const Key& glk, u64* out, int batch, long long sin,
                long long sout, hipStream_t s) {
    const long long N = c.n;
    const int L = level + 1;
    const u32* perm = c.coeff_perm(g);
    const size_t ks_rows = ks_ws_rows(c, level) + L;
    const long long sp = 2LL * L * N;
    for_tiles(c, ks_rows + 2 * (size_t)L, batch, s, [&](int nb, int b0, u64* ws, int tb, hipStream_t st) {
        u64* p = ws + ks_rows * N * tb;
        const u64* ct = in + (size_t)b0 * sin;
        bfv_key_switch(c, level, ct + (long long)L * N, sin, glk, p, sp, ct, sin, L, 1, nb, ws, st);   // p0 = c0 + ks0
        launch_permute_coeff(c, perm, p, sp, out + (size_t)b0 * sout, sout, 2 * L, rm_seq(L), nb, st);
    });
}

void bfv_rescale(Context& c, int level, int polys, const u64* in, u64* out, int batch, long long sin, long long sout,
                 hipStream_t s) {
    for_tiles(c, rescale_ws_rows(level, polys), batch, s, [&](int nb, int b0, u64* ws, int, hipStream_t st) {
        rescale(c, level, polys, in + (size_t)b0 * sin, sin, out + (size_t)b0 * sout, sout, nb, false, ws, st);
    });
}

}  // namespace lsa
