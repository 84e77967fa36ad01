// ntt_r16.h — the 8-stage passes of a two-pass transform as 16 x 16 register butterflies ("radix-16 squared").
//
// Both passes of N = 2^16 (and the second pass of N = 2^15, the first of N = 2^17) have mu = 8 stages on a 4096-point tile:
// the tile is SIXTEEN independent 256-point transforms (first pass: 16 adjacent columns of the 256 x N/256 matrix view;
// second pass: 16 consecutive 256-point chunks), and a 256-point transform is two radix-16 groups per point.  ntt_core.h's
// generic pass stages the whole tile through LDS around every radix group (three LDS round trips, three barriers).  Here
// a 16-lane group owns one 256-point transform: lane (k, i) holds the 16 points i + 16 e in registers for the first four
// stages and 16 i + e for the last four, and the two register images are exchanged through LDS ONCE:
//   * the first group's operands are loaded from global memory straight into the registers that use them -- 8 bytes per
//     lane, 16 lanes = one 128-byte line, so every line is still fetched whole by one instruction -- and (first pass) the
//     second group's results are stored the same way: one LDS exchange, one barrier per tile;
//   * in the second pass the sixteen lanes of a transform sit in ONE wavefront, so its exchanges (and the final transposition
//     that turns 16 consecutive points per lane back into 16-byte coalesced stores) are wave-local: no workgroup barrier at
//     all, the LDS is used as a wavefront shuffle network;
//   * no separate load / store phases: ~35 % fewer vector instructions per point than the staged pass (FP64 engine).
// The butterfly arithmetic, the twiddle tables and the load / store conversions (incl. the fused prologue / epilogue) are the
// ones of ntt_core.h (ntt_group_int / ntt_group_fp, ntt_load_fix / ntt_store_fix): same residues bit for bit.
//
// Inverse transforms run the mirror image (Gentleman-Sande, last stages first).
#pragma once
#include "ntt_core.h"

#define LSA_R16_THREADS 256
#define LSA_R16_LDS_WORDS (272 * 16)

// PASS 0: first pass of a two-pass plan  {s_lo = 0, mu = 8, lambda = 4, tau = 12}: local l = (i << 4) | k   (k = column)
// PASS 1: second pass                    {s_lo = logn - 8, mu = 8, lambda = 0, tau = 12}: l = (k << 8) | i  (k = 256-point chunk)
// MU = 8: a pass of 8 stages = two radix-16 groups per point (N = 2^16 both passes, 2^15 second, 2^17 first);
// MU = 7: 7 stages = a radix-16 group and a radix-8 group, two of the latter per thread (N = 2^14 both passes, 2^15 first).
// The twiddle tables are laid out for the plan's radix split (ntt_split): 4 + (MU - 4) for exactly these two.
// MU = 9 (second pass only: N = 2^17, 2^18): three radix-8 groups per point, see "nine-stage second pass" below.
LSA_HD bool ntt_r16_shape_ok(const NttPassArgs& a, int npass) {
    if (npass != 2 || a.tau != 12) return false;
    if (a.mu == 9) return a.s_lo == a.logn - 9 && a.lambda == 0;
    if (a.mu != 8 && a.mu != 7) return false;
    return (a.s_lo == 0 && a.lambda == 12 - a.mu) || (a.s_lo == a.logn - a.mu && a.lambda == 0);
}

// LDS word of point i (0..255) of the tile's transform k (0..15): one padding word per 16 points; transform-minor in the
// first pass (a wave's lanes are 16 columns x 4 points), transform-major in the second (16 points x 4 chunks)
template <int PASS, int MU>
LSA_HD int r16_lds(int k, int i) {
    const int p = i + (i >> 4);
    return PASS == 0 ? (p << (12 - MU)) + k : k * ((1 << MU) + (1 << (MU - 4))) + p;
}
// element index inside the limb
template <int PASS, int MU>
LSA_HD long long r16_x(const NttPassArgs& a, int tile, int k, int i) {
    return PASS == 0 ? ((long long)tile << (12 - MU)) + ((long long)i << (a.logn - MU)) + k : ((long long)tile << 12) + (k << MU) + i;
}
// the e-th point a thread holds: i + 2^(MU-4) e in the strided image (first four stages), 16 i + e in the contiguous one
template <int MU, int IMAGE>
LSA_HD int r16_pt(int i, int e) {
    return IMAGE == 0 ? i + (e << (MU - 4)) : 16 * i + e;
}
LSA_HD u64 r16_load1(const u64* p) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LSA_NTT_NO_NT)
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
LSA_HD void r16_store1(u64* p, u64 v) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LSA_NTT_NO_NT)
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

struct R16Limb {   // per-block engine constants
    ModDev md;
    const u64* tw;       // integer engine: the limb's {w, ws} pairs
    const double* twd;   // FP64 engine: the limb's doubles
    double q, qinv;
    int s_lo;
};
LSA_HD R16Limb r16_limb(const NttPassArgs& a, const NttBlockCtx& bc) {
    R16Limb L;
    L.md = a.mods[bc.mod];
    L.tw = a.tw + (((long long)bc.mod << a.logn) << 1);
    L.twd = a.twd + ((long long)bc.mod << a.logn);
    L.q = (double)L.md.q;
    L.qinv = 1.0 / L.q;
    L.s_lo = a.s_lo;
    return L;
}

// the radix groups of one image on either engine, in place on the raw 64-bit images (doubles' bits on the FP64 engine).
// HI = 0: ONE radix-16 group, stages s_lo .. s_lo+3 (twiddle group G = G1); HI = 1: the last MU-4 stages on the thread's 16
// contiguous points = 2^(8-MU) groups of radix 2^(MU-4), group g with twiddle group (G1 << 4) + (i << (8-MU)) + g (G = that of
// g = 0).  The first group of a first pass has G = 0 for every lane of the grid: its 15 twiddles are scalar loads.
template <int PASS, int HI, int MU>
LSA_HD void r16_group(u64 (&v)[16], const NttPassArgs& a, const NttBlockCtx& bc, const R16Limb& L, unsigned G) {
    constexpr int R = HI ? MU - 4 : 4, NG = 16 >> R, E = 1 << R;
    const int s_base = (PASS == 0 ? 0 : L.s_lo) + 4 * HI;
    constexpr bool TWU = PASS == 0 && HI == 0;
    const bool scale_here = PASS == 0 && HI == 0 && a.inverse && a.apply_scale;   // (the transform's stage 0 is in the first pass)
    if (bc.fp) {
        const double sc0 = scale_here ? a.scaled[2 * bc.mod] : 0.0, sc1 = scale_here ? a.scaled[2 * bc.mod + 1] : 0.0;
#pragma unroll
        for (int g = 0; g < NG; g++) {
            double d[E];
#pragma unroll
            for (int e = 0; e < E; e++) d[e] = d_from_bits(v[g * E + e]);
            ntt_group_fp<R, TWU>(d, a.inverse != 0, L.twd, s_base, G + (unsigned)g, L.q, L.qinv, scale_here, sc0, sc1, false);
#pragma unroll
            for (int e = 0; e < E; e++) v[g * E + e] = d_to_bits(d[e]);
        }
    } else {
        const u64* sc = a.scale + 4 * bc.mod;
        const u64 sc0 = scale_here ? sc[0] : 0, sc0s = scale_here ? sc[1] : 0, sc1 = scale_here ? sc[2] : 0, sc1s = scale_here ? sc[3] : 0;
#pragma unroll
        for (int g = 0; g < NG; g++) {
            u64 w[E];
#pragma unroll
            for (int e = 0; e < E; e++) w[e] = v[g * E + e];
            ntt_group_int<R, TWU>(w, a.inverse != 0, L.tw, s_base, G + (unsigned)g, L.md.q, scale_here, sc0, sc0s, sc1, sc1s);
#pragma unroll
            for (int e = 0; e < E; e++) v[g * E + e] = w[e];
        }
    }
}

// thread -> (transform k, index inside the group's complement) for the "stride-16" image (points i + 16 e) and the
// "contiguous" image (points 16 i + e): the SAME split of the thread index serves both images of a pass
template <int PASS, int MU>
LSA_HD void r16_lane(int tid, int& k, int& i) {
    if (PASS == 0) {   // 2^(12-MU) transforms (columns) side by side: consecutive lanes = consecutive columns
        k = tid & ((1 << (12 - MU)) - 1);
        i = tid >> (12 - MU);
    } else {           // 2^(MU-4) lanes per transform
        i = tid & ((1 << (MU - 4)) - 1);
        k = tid >> (MU - 4);
    }
}
// twiddle group of transform k at the pass's first stage
template <int PASS, int MU>
LSA_HD unsigned r16_G1(const NttBlockCtx& bc, int k) {
    return PASS == 0 ? 0u : ((unsigned)bc.tile << (12 - MU)) + (unsigned)k;
}

// ---------------------------------------------------------------------------------------------- phases
// A pass is three (first pass: two) phases separated by LDS exchanges.  `stride16` image: lane holds points i + 16 e;
// `contig` image: points 16 i + e.  Forward: stride16 first (stages s_lo..s_lo+3), then contig.  Inverse: contig first.

// global -> registers (stride-16 or contiguous-by-rows image; 8 bytes per lane, 16 lanes = one 128-byte line) with the
// load-side conversions, IMAGE = 0 stride16, 1 contig (contig only ever direct in the first pass)
template <int PASS, bool FZ, int IMAGE, int MU>
LSA_HD void r16_load_direct(const NttPassArgs& a, const NttBlockCtx& bc, int tid, u64 (&v)[16]) {
    const u64* g;
    const u64* gl;
    const NttLoadFix f = ntt_make_load_fix<FZ>(a, bc, g, gl);
    int k, i;
    r16_lane<PASS, MU>(tid, k, i);
    u64 t[16];
    if (FZ && f.add) {
        // two operands per point (merged ModDown + rescale prologue): two rounds of eight so that 16, not 32, loads are in flight
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int e = 8 * h; e < 8 * h + 8; e++) {
                const long long x = r16_x<PASS, MU>(a, bc.tile, k, r16_pt<MU, IMAGE>(i, e));
                v[e] = r16_load1(g + x);
                t[e] = r16_load1(gl + x);
            }
            // the block-uniform cases of ntt_load_fix as branches around the loop (inside it the compiler evaluates both the
            // FP64 lift and the integer lift -- a general reduction -- for every element and selects)
            if (f.fp_lift) {
#pragma unroll
                for (int e = 8 * h; e < 8 * h + 8; e++) {
                    const double td = u52_to_double(t[e]);
                    const double r = td > f.hd ? td - f.qld : td;
                    v[e] = d_to_bits(u52_to_double(v[e]) + r);
                }
            } else {
#pragma unroll
                for (int e = 8 * h; e < 8 * h + 8; e++) v[e] = ntt_load_fix(f, v[e], t[e]);
            }
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = r16_load1(g + r16_x<PASS, MU>(a, bc.tile, k, r16_pt<MU, IMAGE>(i, e)));
    if (!(FZ && f.head)) {   // plain load: the only conversion is u64 -> double on FP64-engine limbs that are not handed over raw
        if (f.fp && !f.raw) {
#pragma unroll
            for (int e = 0; e < 16; e++) v[e] = d_to_bits(u52_to_double(v[e]));
        }
        return;
    }
    NttLoadFix f1 = f;
    f1.add = false;
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = ntt_load_fix(f1, v[e], v[e]);
}
// registers -> global, the mirror image (never the last pass of a forward transform in the second pass: that one needs
// 16-byte coalesced stores and goes through r16_store_coalesced)
template <int PASS, bool FZ, int IMAGE, int MU>
LSA_HD void r16_store_direct(const NttPassArgs& a, const NttBlockCtx& bc, int tid, const u64 (&v)[16]) {
    u64* g;
    const u64* pa;
    const u64* pb;
    const NttStoreFix f = ntt_make_store_fix<FZ>(a, bc, g, pa, pb);
    int k, i;
    r16_lane<PASS, MU>(tid, k, i);
    if (FZ && f.tail) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            u64 va[8], vb[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const long long x = r16_x<PASS, MU>(a, bc.tile, k, r16_pt<MU, IMAGE>(i, 8 * h + e));
                va[e] = r16_load1(pa + x);
                vb[e] = f.with_base ? r16_load1(pb + x) : 0;
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const long long x = r16_x<PASS, MU>(a, bc.tile, k, r16_pt<MU, IMAGE>(i, 8 * h + e));
                r16_store1(g + (a.fz_scatter ? (long long)a.fz_scatter[x] : x), ntt_store_fix(f, v[8 * h + e], va[e], vb[e]));
            }
        }
        return;
    }
    // plain store: the block-uniform cases of ntt_store_fix as branches around the loop instead of selects inside it
    u64 w[16];
    if (f.fp && f.raw && f.skip_reduce) {
#pragma unroll
        for (int e = 0; e < 16; e++) w[e] = v[e];
    } else if (f.fp && f.raw) {
#pragma unroll
        for (int e = 0; e < 16; e++) w[e] = d_to_bits(fp_reduce(d_from_bits(v[e]), f.qd, f.qinvd));
    } else {
#pragma unroll
        for (int e = 0; e < 16; e++) w[e] = ntt_store_fix(f, v[e], 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; e++) r16_store1(g + r16_x<PASS, MU>(a, bc.tile, k, r16_pt<MU, IMAGE>(i, e)), w[e]);
}
// LDS exchange
template <int PASS, int IMAGE, int MU>
LSA_HD void r16_lds_put(int tid, u64* lds, const u64 (&v)[16]) {
    int k, i;
    r16_lane<PASS, MU>(tid, k, i);
#pragma unroll
    for (int e = 0; e < 16; e++) lds[r16_lds<PASS, MU>(k, r16_pt<MU, IMAGE>(i, e))] = v[e];
}
template <int PASS, int IMAGE, int MU>
LSA_HD void r16_lds_get(int tid, const u64* lds, u64 (&v)[16]) {
    int k, i;
    r16_lane<PASS, MU>(tid, k, i);
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = lds[r16_lds<PASS, MU>(k, r16_pt<MU, IMAGE>(i, e))];
}
// second pass: the tile's 4096 consecutive points as 16-byte pairs, wave w owning chunks 4w .. 4w+3 (the ones its lanes
// transform): pair p = lane + 64 m of the wave's 512 pairs
template <int MU>
LSA_HD void r16_pair_pos(int tid, int m, int& k, int& i) {
    const int wave = tid >> 6, lane = tid & 63;
    const int c = 2 * (lane + 64 * m);   // element inside the wave's 1024
    k = (wave << (10 - MU)) + (c >> MU);
    i = c & ((1 << MU) - 1);
}
// second pass, inverse: 16-byte coalesced loads -> LDS (contiguous image is read back from there)
template <bool FZ, int MU>
LSA_HD void r16_load_coalesced(const NttPassArgs& a, const NttBlockCtx& bc, int tid, u64* lds) {
    const u64* g;
    const u64* gl;
    NttLoadFix f = ntt_make_load_fix<FZ>(a, bc, g, gl);
    f.add = false;   // (a fused prologue lives in a forward first pass: never here)
    u64 st[16];
#pragma unroll
    for (int m = 0; m < 8; m++) {
        int k, i;
        r16_pair_pos<MU>(tid, m, k, i);
        ntt_load_data_pair(g + r16_x<1, MU>(a, bc.tile, k, i), st[2 * m], st[2 * m + 1]);
    }
#pragma unroll
    for (int m = 0; m < 8; m++) {
        int k, i;
        r16_pair_pos<MU>(tid, m, k, i);
        lds[r16_lds<1, MU>(k, i)] = ntt_load_fix(f, st[2 * m], st[2 * m]);
        lds[r16_lds<1, MU>(k, i + 1)] = ntt_load_fix(f, st[2 * m + 1], st[2 * m + 1]);
    }
}
// second pass, forward: LDS -> 16-byte coalesced stores with the store-side conversions and the fused epilogue
template <bool FZ, int MU>
LSA_HD void r16_store_coalesced(const NttPassArgs& a, const NttBlockCtx& bc, int tid, const u64* lds) {
    u64* g;
    const u64* pa;
    const u64* pb;
    const NttStoreFix f = ntt_make_store_fix<FZ>(a, bc, g, pa, pb);
#pragma unroll
    for (int m0 = 0; m0 < 8; m0 += LSA_NTT_STORE_CHUNK) {
        u64 v[2 * LSA_NTT_STORE_CHUNK], va[2 * LSA_NTT_STORE_CHUNK], vb[2 * LSA_NTT_STORE_CHUNK];
        long long xs[LSA_NTT_STORE_CHUNK];
#pragma unroll
        for (int m = 0; m < LSA_NTT_STORE_CHUNK; m++) {
            int k, i;
            r16_pair_pos<MU>(tid, m0 + m, k, i);
            xs[m] = r16_x<1, MU>(a, bc.tile, k, i);
            v[2 * m] = lds[r16_lds<1, MU>(k, i)];
            v[2 * m + 1] = lds[r16_lds<1, MU>(k, i + 1)];
            va[2 * m] = va[2 * m + 1] = vb[2 * m] = vb[2 * m + 1] = 0;
            if (FZ && f.tail) ntt_load_data_pair(pa + xs[m], va[2 * m], va[2 * m + 1]);
            if (FZ && f.with_base) ntt_load_data_pair(pb + xs[m], vb[2 * m], vb[2 * m + 1]);
        }
        // the block-uniform cases of ntt_store_fix as branches around the loop: FP64 limb with the merged tail (the headline's
        // epilogue), FP64 limb plain, everything else through the general function
        u64 w[2 * LSA_NTT_STORE_CHUNK];
        if (f.fp && !f.raw && FZ && f.tail && f.merged) {
#pragma unroll
            for (int j = 0; j < 2 * LSA_NTT_STORE_CHUNK; j++) {
                double r = f.skip_reduce ? d_from_bits(v[j]) : fp_reduce(d_from_bits(v[j]), f.qd, f.qinvd);
                const double ad = u52_to_double(va[j]), bd = f.with_base ? u52_to_double(vb[j]) : 0.0;
                r = fp_modmul(fp_modmul(ad, f.kd, f.qd, f.qinvd) - r + bd, f.k2d, f.qd, f.qinvd);
                r = fp_reduce(r, f.qd, f.qinvd);
                if (r < 0) r += f.qd;
                w[j] = double_to_u52(r);
            }
        } else if (f.fp && !f.raw && !(FZ && f.tail)) {
#pragma unroll
            for (int j = 0; j < 2 * LSA_NTT_STORE_CHUNK; j++) {
                double r = f.skip_reduce ? d_from_bits(v[j]) : fp_reduce(d_from_bits(v[j]), f.qd, f.qinvd);
                if (r < 0) r += f.qd;
                w[j] = double_to_u52(r);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2 * LSA_NTT_STORE_CHUNK; j++) w[j] = ntt_store_fix(f, v[j], va[j], vb[j]);
        }
        if (FZ && f.tail && a.fz_scatter) {   // the rotation's index map applied by the store (block-uniform branch)
#pragma unroll
            for (int m = 0; m < LSA_NTT_STORE_CHUNK; m++) {
                g[a.fz_scatter[xs[m]]] = w[2 * m];
                g[a.fz_scatter[xs[m] + 1]] = w[2 * m + 1];
            }
            continue;
        }
#pragma unroll
        for (int m = 0; m < LSA_NTT_STORE_CHUNK; m++) ntt_store_pair(g + xs[m], w[2 * m], w[2 * m + 1]);
    }
}

// ---- A/B build (-DLSA_AB_DPP_LO): the first four stages of a second pass (FP64 engine, forward, MU = 8) as CROSS-LANE
// butterflies -- the wavefront-shuffle formulation: every lane keeps its 16 consecutive points for the whole pass, the
// partner of a stage sits 8 / 4 / 2 / 1 lanes away in the 16-lane row (DPP row_ror / quad_perm moves), both lanes of a pair
// compute the twiddle product.  It needs the consecutive-points image from the start, i.e. a coalesced load staged through
// LDS instead of the direct strided load, and ~14 instead of 4 vector instructions per point and stage.  Measured against the
// register/LDS formulation in profiles/r03/ab_ntt_dpp_last_stages.log; not part of the product build.
#if defined(LSA_AB_DPP_LO) && defined(__HIP_DEVICE_COMPILE__)
template <int D>
__device__ __forceinline__ int r16_dpp_xor32(int v) {
    if (D == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false);   // row_ror:8
    if (D == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
    if (D == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    int r = __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xA, false);              // banks 1,3 (bit 2 set): from lane - 4 (row_ror:4)
    return __builtin_amdgcn_update_dpp(r, v, 0x12C, 0xF, 0x5, false);               // banks 0,2: from lane + 4 (row_ror:12)
}
template <int D>
__device__ __forceinline__ double r16_dpp_xor(double x) {
    const u64 b = d_to_bits(x);
    const u32 lo = (u32)r16_dpp_xor32<D>((int)(u32)b), hi = (u32)r16_dpp_xor32<D>((int)(u32)(b >> 32));
    return d_from_bits(((u64)hi << 32) | lo);
}
template <int J>
__device__ __forceinline__ void r16_dpp_stage(double (&v)[16], const double* twd, int s_lo, unsigned G1, int i_hi, double q, double qinv) {
    constexpr int D = 8 >> J;
    const int s = s_lo + J, kk = i_hi >> (4 - J);
    const double w = (twd + (1LL << s))[ntt_tw_pos_fp(s, J, G1, kk)];
    const bool upper = (i_hi >> (3 - J)) & 1;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const double X = v[e], P = r16_dpp_xor<D>(X);
        const double U = upper ? P : X, V = upper ? X : P;
        const double T = fp_modmul(V, w, q, qinv);
        v[e] = upper ? U - T : U + T;
    }
}
#endif

// The phases of a pass, in execution order; `sync` between them is a workgroup barrier in the first pass and a wavefront-local
// ordering point in the second (kernels.hip); the CPU replay runs each phase for every thread in turn.
//   forward: [0] load stride16, group LO, put stride16   | [1] get contig, group HI, (PASS 0: store contig) (PASS 1: put contig) | [2] PASS 1: coalesced store
//   inverse: [0] (PASS 0: load contig) (PASS 1: coalesced load -> LDS) | [1] (PASS 1: get contig) group HI, put contig | [2] get stride16, group LO, store stride16
template <int PASS, int FZ, int MU>
LSA_HD void r16_phase(const NttPassArgs& a, const NttBlockCtx& bc, int tid, u64* lds, int phase, u64 (&v)[16]) {
    const R16Limb L = r16_limb(a, bc);
    int k, i;
    r16_lane<PASS, MU>(tid, k, i);
    const unsigned G1 = r16_G1<PASS, MU>(bc, k), G2 = (G1 << 4) + ((unsigned)i << (8 - MU));
#if defined(LSA_AB_DPP_LO) && defined(__HIP_DEVICE_COMPILE__)
    if (PASS == 1 && MU == 8 && !a.inverse && bc.fp) {   // A/B: cross-lane first four stages (see above)
        if (phase == 0) {
            r16_load_coalesced<false, MU>(a, bc, tid, lds);
        } else if (phase == 1) {
            r16_lds_get<PASS, 1, MU>(tid, lds, v);
            double d[16];
#pragma unroll
            for (int e = 0; e < 16; e++) d[e] = d_from_bits(v[e]);
            r16_dpp_stage<0>(d, L.twd, L.s_lo, G1, i, L.q, L.qinv);
            r16_dpp_stage<1>(d, L.twd, L.s_lo, G1, i, L.q, L.qinv);
            r16_dpp_stage<2>(d, L.twd, L.s_lo, G1, i, L.q, L.qinv);
            r16_dpp_stage<3>(d, L.twd, L.s_lo, G1, i, L.q, L.qinv);
#pragma unroll
            for (int e = 0; e < 16; e++) v[e] = d_to_bits(d[e]);
            r16_group<PASS, 1, MU>(v, a, bc, L, G2);
            r16_lds_put<PASS, 1, MU>(tid, lds, v);
        } else {
            r16_store_coalesced<(FZ & 2) != 0, MU>(a, bc, tid, lds);
        }
        return;
    }
#endif
    if (!a.inverse) {
        if (phase == 0) {
            r16_load_direct<PASS, (FZ & 1) != 0, 0, MU>(a, bc, tid, v);
            r16_group<PASS, 0, MU>(v, a, bc, L, G1);
            r16_lds_put<PASS, 0, MU>(tid, lds, v);
        } else if (phase == 1) {
            r16_lds_get<PASS, 1, MU>(tid, lds, v);
            r16_group<PASS, 1, MU>(v, a, bc, L, G2);
            if (PASS == 0) r16_store_direct<PASS, (FZ & 2) != 0, 1, MU>(a, bc, tid, v);
            else r16_lds_put<PASS, 1, MU>(tid, lds, v);
        } else if (PASS == 1) {
            r16_store_coalesced<(FZ & 2) != 0, MU>(a, bc, tid, lds);
        }
    } else {
        if (phase == 0) {
            if (PASS == 0) r16_load_direct<PASS, false, 1, MU>(a, bc, tid, v);
            else r16_load_coalesced<false, MU>(a, bc, tid, lds);
        } else if (phase == 1) {
            if (PASS == 1) r16_lds_get<PASS, 1, MU>(tid, lds, v);
            r16_group<PASS, 1, MU>(v, a, bc, L, G2);
            r16_lds_put<PASS, 1, MU>(tid, lds, v);
        } else {
            r16_lds_get<PASS, 0, MU>(tid, lds, v);
            r16_group<PASS, 0, MU>(v, a, bc, L, G1);
            r16_store_direct<PASS, false, 0, MU>(a, bc, tid, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------- nine-stage second pass
// N = 2^17 / 2^18: the second pass has 9 stages on 512-point chunks, eight per tile.  Same idea with THREE radix-8 groups per
// point (the plan's split of 9 stages is 3 + 3 + 3: the twiddle tables are laid out for it): lane (k, i), k = tid >> 5 the chunk,
// i = tid & 31, holds 16 points = two radix-8 groups in each of three register images, exchanged through LDS twice --
//   image A (stages 0-2, bits 8..6 of the point index vary): points i + 32 E; group g = E & 1 is v[g], v[g + 2], ..  The 32 lanes of a
//            chunk read 256 contiguous bytes per load instruction, straight from global memory as in the 8-stage pass;
//   image B (stages 3-5, bits 5..3): points (hi << 6) | (e << 3) | lo with hi = 2 (i >> 3) + g, lo = i & 7, held as v[8 g + e] -- with
//            the one-word-per-16 padding the 32 lanes of a chunk hit 32 different banks for every (g, e);
//   image C (stages 6-8, bits 2..0): points 16 i + E, group g = E >> 3 -- the contiguous image the coalesced store / load expects.
// Both lanes' halves of a chunk sit in one wavefront, so the exchanges stay wave-local (no workgroup barrier).
template <int IMAGE>
LSA_HD int r8x3_pt(int i, int E) {
    if (IMAGE == 0) return i + (E << 5);
    if (IMAGE == 2) return 16 * i + E;
    return ((2 * (i >> 3) + (E >> 3)) << 6) | ((E & 7) << 3) | (i & 7);
}
// v-index of element e of group g in each image
template <int IMAGE>
LSA_HD constexpr int r8x3_slot(int g, int e) {
    return IMAGE == 0 ? g + 2 * e : 8 * g + e;
}
template <int IMAGE>
LSA_HD void r8x3_group(u64 (&v)[16], const NttPassArgs& a, const NttBlockCtx& bc, const R16Limb& L, unsigned G1, int i) {
    const int s_base = L.s_lo + 3 * IMAGE;
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const unsigned G = IMAGE == 0 ? G1 : IMAGE == 1 ? (G1 << 3) + 2u * (unsigned)(i >> 3) + (unsigned)g : (G1 << 6) + 2u * (unsigned)i + (unsigned)g;
        if (bc.fp) {
            double d[8];
#pragma unroll
            for (int e = 0; e < 8; e++) d[e] = d_from_bits(v[r8x3_slot<IMAGE>(g, e)]);
            ntt_group_fp<3, false>(d, a.inverse != 0, L.twd, s_base, G, L.q, L.qinv, false, 0.0, 0.0, false);
#pragma unroll
            for (int e = 0; e < 8; e++) v[r8x3_slot<IMAGE>(g, e)] = d_to_bits(d[e]);
        } else {
            u64 w[8];
#pragma unroll
            for (int e = 0; e < 8; e++) w[e] = v[r8x3_slot<IMAGE>(g, e)];
            ntt_group_int<3, false>(w, a.inverse != 0, L.tw, s_base, G, L.md.q, false, 0, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 8; e++) v[r8x3_slot<IMAGE>(g, e)] = w[e];
        }
    }
}
template <int IMAGE>
LSA_HD void r8x3_put(int k, int i, u64* lds, const u64 (&v)[16]) {
#pragma unroll
    for (int E = 0; E < 16; E++) lds[r16_lds<1, 9>(k, r8x3_pt<IMAGE>(i, E))] = v[E];
}
template <int IMAGE>
LSA_HD void r8x3_get(int k, int i, const u64* lds, u64 (&v)[16]) {
#pragma unroll
    for (int E = 0; E < 16; E++) v[E] = lds[r16_lds<1, 9>(k, r8x3_pt<IMAGE>(i, E))];
}
// phases (wave-local ordering points between them):
//   forward: [0] load image A, group A, put | [1] get B, group B, put | [2] get C, group C, put | [3] coalesced store (+ fused epilogue)
//   inverse: [0] coalesced load -> LDS      | [1] get C, group C, put | [2] get B, group B, put | [3] get A, group A, store image A
template <int FZ>
LSA_HD void r8x3_phase(const NttPassArgs& a, const NttBlockCtx& bc, int tid, u64* lds, int phase, u64 (&v)[16]) {
    const R16Limb L = r16_limb(a, bc);
    const int k = tid >> 5, i = tid & 31;
    const unsigned G1 = r16_G1<1, 9>(bc, k);
    if (!a.inverse) {
        if (phase == 0) {
            r16_load_direct<1, false, 0, 9>(a, bc, tid, v);
            r8x3_group<0>(v, a, bc, L, G1, i);
            r8x3_put<0>(k, i, lds, v);
        } else if (phase == 1) {
            r8x3_get<1>(k, i, lds, v);
            r8x3_group<1>(v, a, bc, L, G1, i);
            r8x3_put<1>(k, i, lds, v);
        } else if (phase == 2) {
            r8x3_get<2>(k, i, lds, v);
            r8x3_group<2>(v, a, bc, L, G1, i);
            r8x3_put<2>(k, i, lds, v);
        } else {
            r16_store_coalesced<(FZ & 2) != 0, 9>(a, bc, tid, lds);
        }
    } else {
        if (phase == 0) {
            r16_load_coalesced<false, 9>(a, bc, tid, lds);
        } else if (phase == 1) {
            r8x3_get<2>(k, i, lds, v);
            r8x3_group<2>(v, a, bc, L, G1, i);
            r8x3_put<2>(k, i, lds, v);
        } else if (phase == 2) {
            r8x3_get<1>(k, i, lds, v);
            r8x3_group<1>(v, a, bc, L, G1, i);
            r8x3_put<1>(k, i, lds, v);
        } else {
            r8x3_get<0>(k, i, lds, v);
            r8x3_group<0>(v, a, bc, L, G1, i);
            r16_store_direct<1, false, 0, 9>(a, bc, tid, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------- second pass + key MAC
// The extension transform's second pass fused with the gadget inner product of a key switch (SURVEY K1 + K7): one workgroup
// owns (ciphertext b, target limb tl, tile) and walks the beta digits -- for every digit but the one that contains tl it runs
// the pass on that digit's extended limb (pass-one output in `ext`) exactly as k_ntt_r16<1, 0, MU> does, the limb's own digit
// reads the ciphertext's NTT-domain limb as it is -- and instead of storing the transformed tile it multiplies it with the key's
// two polynomials at the same positions and keeps the sums in registers (the coalesced 16-byte-pair image of the tile, the one
// the plain pass stores from).  The transformed extension (55 of the headline's 727 limb streams per ciphertext) is never
// written, the stand-alone MAC (its 68 reads) never runs; the key tile is re-read per ciphertext from L2 / Infinity Cache
// (consecutive workgroups are consecutive ciphertexts at one (tl, tile): the 64 of a tile share it).
// FP64-engine limbs multiply with the key's double copy (Key::fp): 7 double operations per product-and-add; integer limbs with
// the Montgomery-form key (one lazy REDC per product).  Same canonical residues as launch_ks_mac: the sums are sums mod q.
struct KsFusedArgs {
    const u64* ext;   // [batch][beta * T][N]: pass-one output of the extension transform (own-digit rows unused)
    const u64* cx;    // [batch][L][N]: the switched polynomial, NTT domain (the digits' own limbs)
    const u64* key;   // Montgomery form, compact [beta][2][kcomp][N]
    const double* keyd;
    u64* acc;         // [batch][2][T][N]
    long long sext, scx, sacc;
    const ModDev* mods;
    const u64* tw;
    const double* twd;
    int logn, L, np, nq, beta, kcomp, klvl, batch, allow_fp64, fp_raw_in;
    int xcd_deal;                  // 1: the workgroups of one (tile, limb) -- one key tile -- are dealt to one XCD: one key fetch instead of eight (LSA_KSMAC_XCD=0 off)
    int n_tl;                      // target limbs of this launch (one launch per butterfly engine: each has its own register budget)
    unsigned char tl_list[64];
};

// the MAC of one digit on the pair image: vin = the operand pair values (FP: doubles' bits), acc0/acc1 the two halves' sums
template <int MU, bool FP>
LSA_HD void r16_mac_digit(const KsFusedArgs& g, const NttPassArgs& a, const NttBlockCtx& bc, const R16Limb& L, int tid, int d, int kj,
                          bool own, long long b, int tl, const u64* lds, u64 (&acc0)[16], u64 (&acc1)[16]) {
    const long long N = 1LL << g.logn;
    const u64* k0 = g.key + ((long long)(2 * d) * g.kcomp + kj) * N;
    const u64* k1 = k0 + (long long)g.kcomp * N;
    const double* k0d = g.keyd + ((long long)(2 * d) * g.kcomp + kj) * N;
    const double* k1d = k0d + (long long)g.kcomp * N;
    const u64* own_src = g.cx + b * g.scx + (long long)tl * N;
    const u64 q = L.md.q, qinv = L.md.qinv;
    const bool lazy = ntt_int_lazy(q);
    const u64 one_s = L.tw[1];   // floor(2^64 / q): the Shoup quotient of 1 (entry 0 of the limb's table)
#ifndef LSA_KSMAC_CHUNK
#define LSA_KSMAC_CHUNK 4   // 16-byte pairs whose operand and key loads are in flight together
#endif
    constexpr int CH = LSA_KSMAC_CHUNK;
#pragma unroll
    for (int m0 = 0; m0 < 8; m0 += CH) {
        u64 v[2 * CH], ka[2 * CH], kb[2 * CH];
        long long xs[CH];
#pragma unroll
        for (int m = 0; m < CH; m++) {
            int k, i;
            r16_pair_pos<MU>(tid, m0 + m, k, i);
            xs[m] = r16_x<1, MU>(a, bc.tile, k, i);
            if (own) {
                ntt_load_data_pair(own_src + xs[m], v[2 * m], v[2 * m + 1]);
            } else {
                v[2 * m] = lds[r16_lds<1, MU>(k, i)];
                v[2 * m + 1] = lds[r16_lds<1, MU>(k, i + 1)];
            }
            if (FP) {
                ntt_load_pair(reinterpret_cast<const u64*>(k0d + xs[m]), ka[2 * m], ka[2 * m + 1]);
                ntt_load_pair(reinterpret_cast<const u64*>(k1d + xs[m]), kb[2 * m], kb[2 * m + 1]);
            } else {
                ntt_load_pair(k0 + xs[m], ka[2 * m], ka[2 * m + 1]);
                ntt_load_pair(k1 + xs[m], kb[2 * m], kb[2 * m + 1]);
            }
        }
#pragma unroll
        for (int j = 0; j < 2 * CH; j++) {
            const int e = 2 * m0 + j;
            if (FP) {
                // the transformed value is an integer-valued double of magnitude < 11 q (unreduced forward pass), the own limb a
                // canonical residue: both within fp_modmul's operand range
                const double x = own ? u52_to_double(v[j]) : d_from_bits(v[j]);
                acc0[e] = d_to_bits(d_from_bits(acc0[e]) + fp_modmul(x, d_from_bits(ka[j]), L.q, L.qinv));
                acc1[e] = d_to_bits(d_from_bits(acc1[e]) + fp_modmul(x, d_from_bits(kb[j]), L.q, L.qinv));
            } else {
                u64 x = v[j];
                if (!own) {   // unreduced transform output -> below 4q (lazy limbs: anything below 2^64; others: below 8q)
#if !defined(LSA_NTT_EXACT_BFLY)
                    x = lazy ? shoup_mul_approx(x, 1, one_s, 0 - q) : csub_sign(x, 0 - 4 * q);
#else
                    x = csub(x, 2 * q);
#endif
                }
                // key in Montgomery form: x * k * R^-1 in [0, 2q); the running sums stay below 2q
                acc0[e] = csub_sign(acc0[e] + mont_mul_lazy(x, ka[j], q, qinv), 0 - 2 * q);
                acc1[e] = csub_sign(acc1[e] + mont_mul_lazy(x, kb[j], q, qinv), 0 - 2 * q);
            }
        }
    }
}
template <int MU, bool FP>
LSA_HD void r16_mac_store(const KsFusedArgs& g, const NttPassArgs& a, const NttBlockCtx& bc, const R16Limb& L, int tid, long long b, int tl,
                          const u64 (&acc0)[16], const u64 (&acc1)[16]) {
    const long long N = 1LL << g.logn;
    const int T = g.L + g.np;
    u64* o0 = g.acc + b * g.sacc + (long long)tl * N;
    u64* o1 = o0 + (long long)T * N;
#pragma unroll
    for (int m = 0; m < 8; m++) {
        int k, i;
        r16_pair_pos<MU>(tid, m, k, i);
        const long long x = r16_x<1, MU>(a, bc.tile, k, i);
        u64 r[4] = {acc0[2 * m], acc0[2 * m + 1], acc1[2 * m], acc1[2 * m + 1]};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (FP) {
                double t = fp_reduce(d_from_bits(r[j]), L.q, L.qinv);
                if (t < 0) t += L.q;
                r[j] = double_to_u52(t);
            } else {
                r[j] = csub(r[j], L.md.q);
            }
        }
        ntt_store_pair(o0 + x, r[0], r[1]);
        ntt_store_pair(o1 + x, r[2], r[3]);
    }
}
