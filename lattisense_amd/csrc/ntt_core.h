// ntt_core.h — batched negacyclic NTT / INTT passes over 64-bit RNS limbs for gfx950.
//
// Replaces the NTT kernels of the absent HEonGPU library behind every call site of
// mega_ag_runners/gpu/mega_ag_executors_gpu.cu:185-289.  Ordering and roots follow the ABI's
// canonical (Lattigo) convention: forward = Cooley-Tukey, natural in -> bit-reversed out, twiddle psi^{brv(m+i)};
// inverse = Gentleman-Sande with N^-1 folded into the last stage.
//
// Structure (MI355X-first, not a port of a CUDA warp-shuffle NTT):
//   * a limb of N = 2^logn points is transformed in one or two PASSES; a pass covers `mu` consecutive radix-2 stages
//     [s_lo, s_lo+mu) on a TILE of 2^tau points staged through LDS (<= 34 KiB so 4 workgroups share a CU's 160 KiB
//     and their HBM loads / butterflies / stores overlap);
//     N = 2^13 / 2^14 also have a whole-limb plan (one pass, 512 / 1024 threads, 69 / 136 KiB of LDS), picked per launch;
//   * inside a pass every thread owns radix-2^rho groups (rho <= 4): 16 points live in VGPRs for 4 stages, so a point
//     crosses LDS once per 4 stages instead of once per stage;
//   * all global accesses are 16 B/lane and contiguous per wave (pass A gathers >=128 B column segments);
//   * LDS layout pads one 8-byte word per 16 so that both the stride-16 and the unit-stride sub-passes are
//     conflict-free for ds_read_b64/ds_write_b64 (bank = (addr/4) mod 64, lane groups of 32).  The stride-256 sub-pass of a
//     first pass (consecutive words per lane: 32 lanes span 34 padded words) and the pair accesses of the load / store phases
//     pay one extra cycle per 32 lanes: SQ_LDS_BANK_CONFLICT is 9 % of the LDS-active cycles of a first pass, 0 of a second
//     (profiles/r02/hmult_b256_rocprofv3_summary.txt); a padding that serves those patterns breaks the other two, and the
//     LDS is 2.6 % of the kernel's waits;
//   * blockIdx -> (limb, tile, batch) with batch fastest: co-resident workgroups share one prime's twiddle slice, which
//     therefore stays in L1/L2 (twiddles are excluded from the algorithmic byte count for exactly that reason).
//
// Two butterfly engines, chosen per workgroup from the limb's modulus (results are the same canonical residues):
//   INT  (any q < 2^61): Shoup/Harvey butterflies on v_mad_u64_u32 chains -- every twiddle w is stored with its quotient
//         w' = floor(w*2^64/q); w*v mod q = w*v + t*(2^64-q) with t the quotient estimated from the three high partial
//         products of w'*v, in [0,4q) -- lazy values in [0,8q) forward / [0,4q) inverse (modarith.h, shoup_mul_approx;
//         -DLSA_NTT_EXACT_BFLY builds Harvey's exact-quotient [0,4q) form for A/B);
//   FP64 (q < 2^47, i.e. most CKKS chain primes): every butterfly is 6 double-precision ops
//         h = v*w ; l = fma(v,w,-h) ; c = rndne(h/q) ; d = fma(-c,q,h) ; t = d + l     (all exact, |t| < 1.1 q)
//     — ~9 VALU issue slots per butterfly against ~30 plus VCC hazards on the integer engine (both the 32-bit integer
//     multiplies and the double operations issue at full rate on gfx950, tools/probe_mul.py: the integer engine is bound by
//     its instruction count -- carries, compare/select chains, register-pair moves -- not by the multiplier).  Exactness: operands are integers of magnitude < 2^51,
//     h+l is the exact product (FMA error-free transformation), |h/q| < 2^51 keeps c within 1 of the true quotient, so
//     d and d+l are integers below 2^53 and therefore exact.  A forward pass of <= 9 stages needs no reduction
//     (growth <= 1.1q per stage; longer single-pass transforms reduce after every sub-pass); the inverse reduces once per
//     sub-pass (sums double per stage).
#pragma once
#include "modarith.h"

#ifndef LSA_NTT_THREADS
#define LSA_NTT_THREADS 256
#endif
#define LSA_MAX_PERIOD 192
#if defined(LSA_EMULATE)
#include <cstdio>
#include <cstdlib>
#define LSA_EMU_CHECK(c) do { if (!(c)) { std::fprintf(stderr, "ntt replay: range invariant violated: %s (%s:%d)\n", #c, __FILE__, __LINE__); std::abort(); } } while (0)
#else
#define LSA_EMU_CHECK(c) do { } while (0)
#endif
#define LSA_ROW_SKIP 0xFF
#define LSA_FP64_MAX_BITS 47
#ifndef LSA_NTT_MAX_RHO
#define LSA_NTT_MAX_RHO 4   // largest radix exponent of a sub-pass (2^rho points per thread in VGPRs)
#endif

struct NttPassArgs {
    const u64* src;       // batch base (may equal dst)
    u64* dst;
    long long src_stride;    // elements between consecutive batch items of src
    long long dst_stride;    // ... of dst
    int batch;            // number of batch items
    long long total_tiles;   // batch * rows * tiles-per-limb
    int rows;             // limb-polynomials per batch item, each N contiguous elements
    const ModDev* mods;   // [nmod]
    const u64* tw;        // [nmod][N][2] {w, Shoup quotient of w}, w = psi^{brv(x)} (forward) or psi^{-brv(x)} (inverse)
    const u64* scale;     // [nmod][2][2] inverse only: the same pairs for N^-1 and psiinv[1] * N^-1
    const double* twd;    // [nmod][N] the same twiddles as plain doubles (FP64 engine)
    const double* scaled; // [nmod][2]
    // FP64-engine limbs of a two-pass transform travel between the passes as integer-valued doubles, |v| <= q/2 + 1 (the
    // first pass stores the reduced value's bits, the second loads them as they are): no sign fix and no conversions at the
    // hand-off -- 6 (forward) / 9 (inverse) of the ~94 double-precision operations a point costs per transform
    int fp_raw_in, fp_raw_out;
    int logn, s_lo, mu, lambda, tau;
    int inverse;          // 0 forward, 1 inverse
    int apply_scale;      // inverse: this pass contains global stage 0 -> fold N^-1
    int final_reduce;     // store fully reduced [0,q)
    int allow_fp64;       // 0 forces the integer engine for every limb
    int period;           // row r uses modulus mod_of[r % period]
    int row0, row_step;   // the launch's i-th row is row r = row0 + i * row_step of the batch item
    int compact;          // 1: the grid covers only the ACTIVE rows; the launch's i-th row is row_tbl[i] (rows whose modulus is
                          // LSA_ROW_SKIP get no workgroups at all: the digit's own limbs in the extension transform and the Q
                          // limbs in the ModDown inverse were 19 % / 70 % of those grids, ~3 ns of dispatch each)
    unsigned short row_tbl[LSA_MAX_PERIOD];
    int row_inner;        // workgroup order: 0 = (row, tile, batch), 1 = (tile, row, batch), batch fastest in both
    // ---- fused element-wise tails (rows of the transformed buffer are [poly][fz_limbs]):
    // epilogue of the LAST pass of a forward transform, replaces the plain store of the transformed value v:
    //   fz_epi = 1:  out[poly][limb] = (fz_a[poly][limb] - v) * fz_k[limb]  (+ fz_base[poly][limb] if poly < fz_base_polys)
    //                (ModDown tail: (acc_Q - conv) * P^-1 + d;   rescale tail: (c - t) * q_l^-1)
    // prologue of the FIRST pass, replaces the plain load:
    //   fz_pro = 1:  in = lift = ((fz_last[poly] + h) mod q_l) mod q_limb - (h mod q_limb),  h = (q_l - 1)/2  (rescale head)
    //   fz_pro = 2:  in = src + lift   (merged ModDown + rescale: one transform of conv*P^-1 + lift serves both steps)
    //   fz_epi = 2:  out = (fz_a * fz_k - v + fz_base) * fz_k2   (its tail: (acc*P^-1 - NTT(in) + base) * q_l^-1)
    int fz_epi, fz_pro, fz_limbs, fz_base_polys, fz_ql_mod;
    int fz_a_rpp, fz_base_rpp, fz_out_rpp, fz_last_rpp;
    const u64* fz_a;
    const u64* fz_base;
    const u64* fz_k;
    const u64* fz_k2;
    u64* fz_out;
    // with an epilogue: out[perm^-1 ...] -- the result row is written through this index map, out[fz_scatter[x]] = value(x): the
    // NTT-domain automorphism of a rotation applied by the store of the key switch's last pass (null: plain store)
    const unsigned* fz_scatter;
    const u64* fz_last;
    long long fz_a_stride, fz_base_stride, fz_out_stride, fz_last_stride;
    unsigned long long* diag;   // diagnostic builds (LSA_NTT_DIAG_STAMPS): per-workgroup phase time stamps, else unused
    unsigned char mod_of[LSA_MAX_PERIOD];
};

LSA_HD int lds_addr(int l) { return l + (l >> 4); }
LSA_HD int lds_words(int tau) { return (1 << tau) + (1 << (tau - 4)) + 16; }

// ---- twiddle table layout.  Stage s owns the 2^s entries [2^s, 2^(s+1)) of a limb's table; a butterfly group G of a
// radix-2^rho sub-pass uses, at its j-th stage, the 2^j entries of natural index (G << j) + k, k < 2^j.  In natural order a
// wave's lanes (consecutive G) read entries 2^j apart: at j = 3 every lane touches its own 128-byte line, 64 lines per load
// instruction, 680 line visits per wave and sub-pass on the integer engine -- 3..6 times what the tile's own data costs the
// address unit.  The tables are therefore stored "k-major" per stage, for the j that stage has in the plan the table is
// built for (ntt_plan.h): position (k << (s - j)) + G for the integer engine's 16-byte {w, w'} pairs -- a load instruction
// of the wave reads 1 KiB contiguous -- and, for the FP64 engine's 8-byte entries, two k side by side:
// ((k >> 1) << (s - j + 1)) + 2 G + (k & 1), again one 16-byte load per lane.
// -DLSA_NTT_TW_NATURAL keeps the natural order (A/B builds).
// position = (wave-uniform part, from s, j, k) + (per-lane part, from G): the kernel adds the first to the table's scalar
// base and keeps the second as the load's 32-bit vector offset
LSA_HD long long ntt_tw_u_int(int s, int j, int k) {
#if defined(LSA_NTT_TW_NATURAL)
    return k;
#else
    return (long long)k << (s - j);
#endif
}
LSA_HD unsigned ntt_tw_v_int(int j, unsigned G) {
#if defined(LSA_NTT_TW_NATURAL)
    return G << j;
#else
    return G;
#endif
}
LSA_HD long long ntt_tw_u_fp(int s, int j, int k) {
#if defined(LSA_NTT_TW_NATURAL)
    return k;
#else
    return j == 0 ? 0 : ((long long)(k >> 1) << (s - j + 1)) + (k & 1);
#endif
}
LSA_HD unsigned ntt_tw_v_fp(int j, unsigned G) {
#if defined(LSA_NTT_TW_NATURAL)
    return G << j;
#else
    return j == 0 ? G : 2 * G;
#endif
}
LSA_HD long long ntt_tw_pos_int(int s, int j, long long G, int k) { return ntt_tw_u_int(s, j, k) + ntt_tw_v_int(j, (unsigned)G); }
LSA_HD long long ntt_tw_pos_fp(int s, int j, long long G, int k) { return ntt_tw_u_fp(s, j, k) + ntt_tw_v_fp(j, (unsigned)G); }
// The tile's own data and the fused tails' operands are read once and written once per pass: they go through the memory
// pipeline with the NON-TEMPORAL policy (`nt`: served by L2, no L1 allocation), which leaves the CU's L1 to the per-point
// constants.  FP64-engine transforms +5-8 % (N = 2^14 whole-limb plan +10-15 %), integer +4-7 %, headline +2.4 %, rotate
// +2.7 %, BFV +3 % (profiles/r02/ab_ntt_nontemporal_tile_data.log).  The same policy on the element-wise kernels' streams
// changes nothing.  -DLSA_NTT_NO_NT builds the default-policy accesses (A/B).
#if !defined(LSA_NTT_NO_NT)
typedef u64 lsa_v2u64 __attribute__((ext_vector_type(2)));
#endif
LSA_HD void ntt_load_data_pair(const u64* p, u64& x, u64& y) {   // 16-byte aligned; the tile's own data
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LSA_NTT_NO_NT)
    const lsa_v2u64 v = __builtin_nontemporal_load(reinterpret_cast<const lsa_v2u64*>(p));
    x = v.x;
    y = v.y;
#elif defined(__HIP_DEVICE_COMPILE__)
    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(p);
    x = v.x;
    y = v.y;
#else
    x = p[0];
    y = p[1];
#endif
}
LSA_HD void ntt_load_pair(const u64* p, u64& x, u64& y) {   // 16-byte aligned
#if defined(__HIP_DEVICE_COMPILE__)
    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(p);
    x = v.x;
    y = v.y;
#else
    x = p[0];
    y = p[1];
#endif
}
// Twiddles whose position is the same for every lane of the wavefront (the first radix group of a first pass: G = 0) are read
// through the CONSTANT address space: wave-uniform loads of plain global memory stay on the scalar unit only until the
// kernel's first store, reads of constant memory always (the tables are never written while a kernel runs).
#if defined(__HIP_DEVICE_COMPILE__)
#define LSA_CONST_PTR(T, p) ((const __attribute__((address_space(4))) T*)(p))
#else
#define LSA_CONST_PTR(T, p) (p)
#endif
// the 2^j FP64-engine twiddles of stage s for butterfly group G (tws = the stage's first entry)
LSA_HD void ntt_load_tw_fp(const double* tws, int s, int j, unsigned G, double* w) {
    const unsigned gv = ntt_tw_v_fp(j, G);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LSA_NTT_TW_NATURAL)
    if (j == 0) {
        w[0] = tws[gv];
        return;
    }
#pragma unroll
    for (int k = 0; k < (1 << j); k += 2) {
        const double2 v = *reinterpret_cast<const double2*>(tws + ntt_tw_u_fp(s, j, k) + gv);
        w[k] = v.x;
        w[k + 1] = v.y;
    }
#else
#pragma unroll
    for (int k = 0; k < (1 << j); k++) w[k] = (tws + ntt_tw_u_fp(s, j, k))[gv];
#endif
}

struct NttBlockCtx {
    long long base_src;   // element offset of this (batch,row) limb in src
    long long base_dst;   // ... in dst
    int tile;             // tile index inside the limb
    int mod;              // modulus index
    int fp;               // 1: FP64 engine
    int b, row;           // batch item and row inside it
    const u64* tw_l;      // sub-passes instantiated with TWL: this limb's twiddles staged in LDS ({w, w'} pairs or doubles)
};

LSA_HD NttBlockCtx ntt_decode_block(const NttPassArgs& a, long long bid) {
    NttBlockCtx c;
    int tiles = 1 << (a.logn - a.tau);
    int b = (int)(bid % a.batch);
    long long rt = bid / a.batch;
    int row;
    int ri;   // index of the row within the launch
    if (a.row_inner) {
        // rows in the middle: consecutive batch-sized runs of workgroups walk over the launch's limbs at one tile position,
        // so the workgroups resident on a CU mix integer-engine (multiply-bound) and FP64-engine (traffic-bound) limbs
        c.tile = (int)(rt / a.rows);
        ri = (int)(rt % a.rows);
    } else {
        c.tile = (int)(rt % tiles);
        ri = (int)(rt / tiles);
    }
    row = a.compact ? (int)a.row_tbl[ri] : a.row0 + ri * a.row_step;
    c.base_src = (long long)b * a.src_stride + ((long long)row << a.logn);
    c.base_dst = (long long)b * a.dst_stride + ((long long)row << a.logn);
    c.mod = a.mod_of[row % a.period];
    c.b = b;
    c.row = row;
    c.fp = 0;
    c.tw_l = nullptr;
    if (c.mod != LSA_ROW_SKIP) c.fp = a.allow_fp64 && (a.mods[c.mod].q >> LSA_FP64_MAX_BITS) == 0;
    return c;
}

// local tile index l -> element index inside the limb (all extents are powers of two: shifts only)
LSA_HD int ntt_global_index(const NttPassArgs& a, int tile, int l) {
    if (a.lambda == 0) return (tile << a.tau) + l;  // contiguous tile
    const int lo_bits = a.logn - a.s_lo - a.mu;     // >= lambda
    const int bph_bits = lo_bits - a.lambda;        // log2(column blocks per hi index)
    const int hi = tile >> bph_bits, lob = tile & ((1 << bph_bits) - 1);
    const int r = l >> a.lambda, c = l & ((1 << a.lambda) - 1);
    return (hi << (a.logn - a.s_lo)) + (r << lo_bits) + (lob << a.lambda) + c;
}
// the same map with the per-tile part hoisted: x(l) = base + ((l >> lambda) << rshift) + (l & cmask)
struct NttTileMap {
    int base, lambda, rshift, cmask;
};
LSA_HD NttTileMap ntt_tile_map(const NttPassArgs& a, int tile) {
    NttTileMap m;
    m.base = ntt_global_index(a, tile, 0);
    m.lambda = a.lambda;
    m.rshift = a.lambda ? a.logn - a.s_lo - a.mu : 0;
    m.cmask = (1 << a.lambda) - 1;
    return m;
}
LSA_HD int ntt_tile_index(const NttTileMap& m, int l) { return m.base + ((l >> m.lambda) << m.rshift) + (l & m.cmask); }

// "hi" index (the s_lo leading bits of the element index) of local element l
LSA_HD int ntt_hi_index(const NttPassArgs& a, int tile, int l) {
    if (a.lambda == 0) return ((tile << a.tau) + l) >> a.mu;
    int lo_bits = a.logn - a.s_lo - a.mu;
    return tile >> (lo_bits - a.lambda);
}

// forward transforms of integer-engine limbs below 2^57 run without per-butterfly conditional subtractions (ntt_group_int)
#if !defined(LSA_NTT_EXACT_BFLY) && !defined(LSA_NTT_NO_LAZY)
LSA_HD bool ntt_int_lazy(u64 q) { return (q >> 57) == 0; }
#else
LSA_HD bool ntt_int_lazy(u64) { return false; }
#endif

// ---- FP64 engine primitives (all results exact, see header)
LSA_HD double d_from_bits(u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)b);
#else
    double d;
    __builtin_memcpy(&d, &b, 8);
    return d;
#endif
}
LSA_HD u64 d_to_bits(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (u64)__double_as_longlong(d);
#else
    u64 b;
    __builtin_memcpy(&b, &d, 8);
    return b;
#endif
}
// exact u64 <-> double for integers below 2^52 through the 2^52 offset trick (2 instructions each way instead of the
// generic 64-bit conversions)
LSA_HD double u52_to_double(u64 v) { return d_from_bits(v | 0x4330000000000000ull) - 4503599627370496.0; }
LSA_HD u64 double_to_u52(double r) { return d_to_bits(r + 4503599627370496.0) & 0x000FFFFFFFFFFFFFull; }
// v*w mod q as an integer-valued double in (-1.1q, 1.1q); |v| < 2^51, 0 <= w < q < 2^47
LSA_HD double fp_modmul(double v, double w, double q, double qinv) {
    const double h = v * w;
    const double l = __builtin_fma(v, w, -h);
    const double c = __builtin_rint(h * qinv);
    const double d = __builtin_fma(-c, q, h);
    return d + l;
}
// x mod q into [-q/2-1, q/2+1] for |x| < 2^53
LSA_HD double fp_reduce(double x, double q, double qinv) { return __builtin_fma(-__builtin_rint(x * qinv), q, x); }

// phase 0: global -> LDS (2 elements = 16 B per lane per step).  For a full tile (every thread owns exactly
// LSA_NTT_STAGE_PAIRS pairs) all of a thread's loads are issued before the first one is consumed: a rolled
// load->convert->ds_write loop serialises one HBM latency per step (measured with the LSA_NTT_DIAG_STAMPS build: 16.3k of
// a workgroup's 34.6k cycles).
#define LSA_NTT_STAGE_PAIRS 8   // 16-byte pairs per thread: a workgroup of NT threads stages tiles up to 2 * 8 * NT points
#ifndef LSA_NTT_HEAD_ROUNDS
#define LSA_NTT_HEAD_ROUNDS 2   // two-operand prologue: load rounds per tile (1 = all 16 operand pairs in flight at once)
#endif
struct NttLoadFix {   // per-block constants of the load-side conversions
    bool head, add, fp, near, raw, fp_lift;
    u64 ql, h, hq;
    double hd, qld;
    ModDev mi;
};
// v: the tile's own element; t: the last limb's element at the same position (head modes only)
LSA_HD u64 ntt_load_fix(const NttLoadFix& f, u64 v, u64 t) {
    if (f.fp_lift) {
        // FP64-engine limb, q_l below 2^48: the engine takes any integer-valued double of small magnitude that is congruent
        // to the input, so the centred remainder of the last limb (t if t <= h, else t - q_l) is used as it is -- no
        // reduction modulo this limb's prime, 9 double-precision / select operations instead of four 64-bit modular ones.
        // |in| < q + q_l / 2 < 2^48.6; a pass adds at most 1.1 q per stage: far below the engine's 2^51.
        const double td = u52_to_double(t);
        const double r = td > f.hd ? td - f.qld : td;
        return d_to_bits(f.add ? u52_to_double(v) + r : r);
    }
    if (f.head) {
        const u64 c = add_mod(t, f.h, f.ql);   // centred remainder + h, in [0, q_l)
        // chain primes are within a factor two of each other almost always: one conditional subtraction then replaces
        // the general reduction
        const u64 lift = sub_mod(f.near ? csub(c, f.mi.q) : reduce_u64(c, f.mi), f.hq, f.mi.q);
        v = f.add ? add_mod(v, lift, f.mi.q) : lift;
    }
    if (f.fp && !f.raw) v = d_to_bits(u52_to_double(v));  // inputs of an FP64-engine limb are canonical or lazy (< 4q < 2^49): exact
    return v;
}
// FZ = false compiles the fused prologue out (plain launches: fewer live constants, smaller code)
// the load-side constants of a block; g / gl: where the tile's own elements and the last limb's elements are read from
template <bool FZ>
LSA_HD NttLoadFix ntt_make_load_fix(const NttPassArgs& a, const NttBlockCtx& bc, const u64*& g, const u64*& gl) {
    g = a.src + bc.base_src;
    gl = g;   // last-limb source of the head modes
    NttLoadFix f;
    f.head = FZ && a.fz_pro && a.s_lo == 0;   // fused rescale head: the tile is derived from the (coefficient-domain) last limb
    f.add = f.head && a.fz_pro == 2;
    f.fp = bc.fp != 0;
    f.raw = f.fp && a.fp_raw_in;   // (never together with a fused prologue: that belongs to the first pass)
    f.mi = a.mods[bc.mod];
    f.ql = f.h = f.hq = 0;
    f.hd = f.qld = 0.0;
    f.near = f.fp_lift = false;
    if (f.head) {
        gl = a.fz_last + (long long)bc.b * a.fz_last_stride + ((long long)(bc.row / a.fz_limbs) * a.fz_last_rpp << a.logn);
        if (!f.add) g = gl;
        f.ql = a.mods[a.fz_ql_mod].q;
        f.h = (f.ql - 1) >> 1;
        f.hq = reduce_u64(f.h, f.mi);
        f.near = f.ql <= 2 * f.mi.q;
        f.fp_lift = f.fp && (f.ql >> 48) == 0;
        f.hd = (double)f.h;
        f.qld = (double)f.ql;
    }
    return f;
}
template <bool FZ, int NT>
LSA_HD void ntt_phase_load(const NttPassArgs& a, const NttBlockCtx& bc, int tid, u64* lds) {
    const u64* g;
    const u64* gl;
    const int half = 1 << (a.tau - 1);
    const NttLoadFix f = ntt_make_load_fix<FZ>(a, bc, g, gl);
    const NttTileMap tm = ntt_tile_map(a, bc.tile);
    if (half == LSA_NTT_STAGE_PAIRS * NT && !f.add) {
        NttLoadFix f1 = f;
        f1.add = false;   // known here: keeps the two-operand arithmetic out of this path's 16 unrolled copies
        u64 st[2 * LSA_NTT_STAGE_PAIRS];
#pragma unroll
        for (int p = 0; p < LSA_NTT_STAGE_PAIRS; p++) {
            const int x = ntt_tile_index(tm, 2 * (tid + p * NT));
            ntt_load_data_pair(g + x, st[2 * p], st[2 * p + 1]);
        }
#pragma unroll
        for (int p = 0; p < LSA_NTT_STAGE_PAIRS; p++) {
            const int l = 2 * (tid + p * NT);
            lds[lds_addr(l)] = ntt_load_fix(f1, st[2 * p], st[2 * p]);
            lds[lds_addr(l + 1)] = ntt_load_fix(f1, st[2 * p + 1], st[2 * p + 1]);
        }
        return;
    }
    if (half == LSA_NTT_STAGE_PAIRS * NT) {   // two operands per element: half the pairs per round
        constexpr int CH = LSA_NTT_STAGE_PAIRS / LSA_NTT_HEAD_ROUNDS;
#pragma unroll 1
        for (int p0 = 0; p0 < LSA_NTT_STAGE_PAIRS; p0 += CH) {
            u64 st[2 * CH], sl[2 * CH];
#pragma unroll
            for (int p = 0; p < CH; p++) {
                const int x = ntt_tile_index(tm, 2 * (tid + (p0 + p) * NT));
                ntt_load_data_pair(g + x, st[2 * p], st[2 * p + 1]);
                ntt_load_data_pair(gl + x, sl[2 * p], sl[2 * p + 1]);
            }
            if (f.fp_lift) {   // (block-uniform: a branch around the loop, not a select per element between this and the integer lift)
#pragma unroll
                for (int p = 0; p < CH; p++) {
                    const int l = 2 * (tid + (p0 + p) * NT);
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const double td = u52_to_double(sl[2 * p + h]);
                        const double r = td > f.hd ? td - f.qld : td;
                        lds[lds_addr(l + h)] = d_to_bits(u52_to_double(st[2 * p + h]) + r);
                    }
                }
            } else {
#pragma unroll
                for (int p = 0; p < CH; p++) {
                    const int l = 2 * (tid + (p0 + p) * NT);
                    lds[lds_addr(l)] = ntt_load_fix(f, st[2 * p], sl[2 * p]);
                    lds[lds_addr(l + 1)] = ntt_load_fix(f, st[2 * p + 1], sl[2 * p + 1]);
                }
            }
        }
        return;
    }
    for (int i = tid; i < half; i += NT) {   // partial tiles (small rings)
        const int x = ntt_tile_index(tm, 2 * i);
        lds[lds_addr(2 * i)] = ntt_load_fix(f, g[x], gl[x]);
        lds[lds_addr(2 * i + 1)] = ntt_load_fix(f, g[x + 1], gl[x + 1]);
    }
}

// final phase: LDS -> global.  Full tiles go in chunks of LSA_NTT_STORE_CHUNK pairs per thread: a chunk's LDS reads and
// (fused tail) operand loads are all issued before the first is consumed, for the same reason as in the load phase.
#ifndef LSA_NTT_STORE_CHUNK
#define LSA_NTT_STORE_CHUNK 4
#endif
struct NttStoreFix {   // per-block constants of the store-side conversions
    bool fp, final_reduce, tail, with_base, merged, raw, skip_reduce, lazy;
    u64 q, qinv, k, k2, one_s;   // one_s = floor(2^64 / q): the Shoup quotient of w = 1 (entry 0 of the limb's twiddle table)
    double qd, qinvd, kd, k2d;   // kd/k2d: the tail factors as plain doubles (FP64-engine limbs)
};
LSA_HD u64 ntt_store_fix(const NttStoreFix& f, u64 v, u64 va, u64 vb) {
    if (f.fp) {
        // the store's own reduction is skipped where it changes nothing that matters: inverse passes end every sub-pass
        // reduced already, and the hand-off of a forward transform whose modulus is below 2^46 stays below 2^51 unreduced
        // (|in| < 4q, + 1.1q per stage over <= 17 stages: 22.7q < 2^50.6)
        double r = f.skip_reduce ? d_from_bits(v) : fp_reduce(d_from_bits(v), f.qd, f.qinvd);   // |r| <= q/2 (+ rounding slack)
        if (f.raw) return d_to_bits(r);   // first pass of a two-pass transform: the second pass's butterflies start from this
        if (f.tail) {   // the fused tail of an FP64-engine limb stays on the FP64 engine: exact, 6 operations per product
            const double ad = u52_to_double(va), bd = f.with_base ? u52_to_double(vb) : 0.0;
            if (f.merged) r = fp_modmul(fp_modmul(ad, f.kd, f.qd, f.qinvd) - r + bd, f.k2d, f.qd, f.qinvd);
            else r = fp_modmul(ad - r, f.kd, f.qd, f.qinvd) + bd;
            r = fp_reduce(r, f.qd, f.qinvd);
        }
        if (r < 0) r += f.qd;   // one conditional add lands in [0, q): canonical on store
        return double_to_u52(r);
    }
    if (f.final_reduce) {
#if !defined(LSA_NTT_EXACT_BFLY)
        if (f.lazy) v = shoup_mul_approx(v, 1, f.one_s, 0 - f.q);   // lazy forward transform: anything below 2^64 -> [0, 4q)
        else v = csub_sign(v, 0 - 4 * f.q);   // forward transforms end below 8q
        v = csub_sign(csub_sign(v, 0 - 2 * f.q), 0 - f.q);
#else
        v = csub(csub(v, 2 * f.q), f.q);
#endif
    }
    if (f.tail) {   // fused tail: the transformed value is consumed here and never stored
        if (f.merged) {
            v = sub_mod(mont_mul(va, f.k, f.q, f.qinv), v, f.q);
            if (f.with_base) v = add_mod(v, vb, f.q);
            v = mont_mul(v, f.k2, f.q, f.qinv);
        } else {
            v = mont_mul(sub_mod(va, v, f.q), f.k, f.q, f.qinv);
            if (f.with_base) v = add_mod(v, vb, f.q);
        }
    }
    return v;
}
LSA_HD void ntt_store_pair(u64* gp, u64 v0, u64 v1) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LSA_NTT_NO_NT)
    lsa_v2u64 w;
    w.x = v0;
    w.y = v1;
    __builtin_nontemporal_store(w, reinterpret_cast<lsa_v2u64*>(gp));
#elif defined(__HIP_DEVICE_COMPILE__)
    ulonglong2 w;
    w.x = v0;
    w.y = v1;
    *reinterpret_cast<ulonglong2*>(gp) = w;
#else
    gp[0] = v0;
    gp[1] = v1;
#endif
}
// the store-side constants of a block; g: where results go, pa / pb: the fused tail's operands (placeholders without one)
template <bool FZ>
LSA_HD NttStoreFix ntt_make_store_fix(const NttPassArgs& a, const NttBlockCtx& bc, u64*& g, const u64*& pa, const u64*& pb) {
    g = a.dst + bc.base_dst;
    const ModDev md = a.mods[bc.mod];
    NttStoreFix f;
    f.fp = bc.fp != 0;
    f.final_reduce = a.final_reduce != 0;
    f.raw = f.fp && a.fp_raw_out;
    f.skip_reduce = f.fp && (a.inverse || (f.raw && (md.q >> 46) == 0));
    f.tail = FZ && a.fz_epi && a.final_reduce;
    f.q = md.q;
    f.qinv = md.qinv;
    f.qd = (double)md.q;
    f.qinvd = 1.0 / f.qd;
    f.k = f.k2 = 0;
    f.kd = f.k2d = 0.0;
    f.merged = false;
    f.lazy = !f.fp && !a.inverse && ntt_int_lazy(md.q);
    f.one_s = f.lazy ? (a.tw + (((long long)bc.mod << a.logn) << 1))[1] : 0;
    const int poly = f.tail ? bc.row / a.fz_limbs : 0, limb = f.tail ? bc.row % a.fz_limbs : 0;
    f.with_base = f.tail && a.fz_base && poly < a.fz_base_polys;
    pa = g;   // placeholders when there is no fused tail (never dereferenced)
    pb = g;
    if (f.tail) {
        pa = a.fz_a + (long long)bc.b * a.fz_a_stride + (((long long)poly * a.fz_a_rpp + limb) << a.logn);
        f.k = a.fz_k[limb];
        f.merged = a.fz_epi == 2;
        if (f.merged) f.k2 = a.fz_k2[limb];
        if (f.fp) {   // Montgomery form -> plain value (k*R * 1 * R^-1), exact as a double below 2^47
            f.kd = (double)mont_mul(f.k, 1, f.q, f.qinv);
            f.k2d = f.merged ? (double)mont_mul(f.k2, 1, f.q, f.qinv) : 0.0;
        }
        g = a.fz_out + (long long)bc.b * a.fz_out_stride + (((long long)poly * a.fz_out_rpp + limb) << a.logn);
        if (f.with_base) pb = a.fz_base + (long long)bc.b * a.fz_base_stride + (((long long)poly * a.fz_base_rpp + limb) << a.logn);
    }
    return f;
}
template <bool FZ, int NT>
LSA_HD void ntt_phase_store(const NttPassArgs& a, const NttBlockCtx& bc, int tid, const u64* lds) {
    u64* g;
    const u64* pa;
    const u64* pb;
    const int half = 1 << (a.tau - 1);
    const NttStoreFix f = ntt_make_store_fix<FZ>(a, bc, g, pa, pb);
    const NttTileMap tm = ntt_tile_map(a, bc.tile);
    if (half == LSA_NTT_STAGE_PAIRS * NT) {
        for (int p0 = 0; p0 < LSA_NTT_STAGE_PAIRS; p0 += LSA_NTT_STORE_CHUNK) {
            u64 v[2 * LSA_NTT_STORE_CHUNK], va[2 * LSA_NTT_STORE_CHUNK], vb[2 * LSA_NTT_STORE_CHUNK];
            int xs[LSA_NTT_STORE_CHUNK];
#pragma unroll
            for (int p = 0; p < LSA_NTT_STORE_CHUNK; p++) {
                const int l = 2 * (tid + (p0 + p) * NT);
                const int x = ntt_tile_index(tm, l);
                xs[p] = x;
                v[2 * p] = lds[lds_addr(l)];
                v[2 * p + 1] = lds[lds_addr(l + 1)];
                va[2 * p] = va[2 * p + 1] = vb[2 * p] = vb[2 * p + 1] = 0;
                if (f.tail) {
                    ntt_load_data_pair(pa + x, va[2 * p], va[2 * p + 1]);
                }
                if (f.with_base) {
                    ntt_load_data_pair(pb + x, vb[2 * p], vb[2 * p + 1]);
                }
            }
            if (FZ && f.tail && a.fz_scatter) {   // block-uniform: a branch around the stores
#pragma unroll
                for (int p = 0; p < LSA_NTT_STORE_CHUNK; p++) {
                    g[a.fz_scatter[xs[p]]] = ntt_store_fix(f, v[2 * p], va[2 * p], vb[2 * p]);
                    g[a.fz_scatter[xs[p] + 1]] = ntt_store_fix(f, v[2 * p + 1], va[2 * p + 1], vb[2 * p + 1]);
                }
                continue;
            }
#pragma unroll
            for (int p = 0; p < LSA_NTT_STORE_CHUNK; p++)
                ntt_store_pair(g + xs[p], ntt_store_fix(f, v[2 * p], va[2 * p], vb[2 * p]),
                               ntt_store_fix(f, v[2 * p + 1], va[2 * p + 1], vb[2 * p + 1]));
        }
        return;
    }
    for (int i = tid; i < half; i += NT) {   // partial tiles (small rings)
        const int x = ntt_tile_index(tm, 2 * i);
        const u64 a0 = f.tail ? pa[x] : 0, a1 = f.tail ? pa[x + 1] : 0;
        const u64 b0 = f.with_base ? pb[x] : 0, b1 = f.with_base ? pb[x + 1] : 0;
        const u64 w0 = ntt_store_fix(f, lds[lds_addr(2 * i)], a0, b0), w1 = ntt_store_fix(f, lds[lds_addr(2 * i + 1)], a1, b1);
        if (FZ && f.tail && a.fz_scatter) {
            g[a.fz_scatter[x]] = w0;
            g[a.fz_scatter[x + 1]] = w1;
        } else {
            ntt_store_pair(g + x, w0, w1);
        }
    }
}

// The 2^RHO-point butterfly group of a sub-pass -- integer (Shoup) engine.  v[e] are the group's points in local order, s_base
// the global index of the group's first stage (j = 0), G the group's index at that stage (twiddle table layout above).
// Forward: stages j = 0..RHO-1 (Cooley-Tukey); inverse: j = RHO-1..0 (Gentleman-Sande); scale_here: the inverse transform's
// last stage is in this group and folds N^-1 (sc = {N^-1, quotient, psi^-1 N^-1, quotient}).
template <int RHO, bool TWU, bool lazy>
LSA_HD void ntt_group_int_impl(u64 (&v)[1 << RHO], bool inverse, const u64* tw /* limb's {w, ws} pairs */, int s_base, unsigned G, u64 q_,
                               bool scale_here, u64 sc0, u64 sc0s, u64 sc1, u64 sc1s) {
    constexpr int E = 1 << RHO;
    const u64 q = q_;
#if defined(LSA_NTT_EXACT_BFLY)
    const u64 q2 = 2 * q;
#else
    const u64 q4 = 4 * q, nq = 0 - q, nq4 = 0 - q4;
#endif
    if (!inverse) {
#pragma unroll
        for (int j = 0; j < RHO; j++) {
            const int half = E >> (j + 1);
            const int s = s_base + j;
            const u64* tws = tw + 2 * (1LL << s);
            u64 w[E / 2], ws[E / 2];   // this stage's 2^j twiddle pairs, one 16-byte load each
#if defined(LSA_NTT_DIAG_TW8)   // diagnostic build: 8-byte twiddles in the FP64 table's layout (wrong results, same arithmetic)
            if (!TWU) {
                double wd[E / 2];
                ntt_load_tw_fp(reinterpret_cast<const double*>(tw) + (1LL << s), s, j, G, wd);
#pragma unroll
                for (int k = 0; k < (1 << j); k++) {
                    w[k] = (u64)__builtin_bit_cast(long long, wd[k]);
                    ws[k] = (w[k] << 3) | 1;
                }
            } else
#endif
#pragma unroll
            for (int k = 0; k < (1 << j); k++) {
                if (TWU) {   // wave-uniform position: scalar loads
                    w[k] = LSA_CONST_PTR(u64, tws)[2 * ntt_tw_u_int(s, j, k) + 2u * ntt_tw_v_int(j, G)];
                    ws[k] = LSA_CONST_PTR(u64, tws)[2 * ntt_tw_u_int(s, j, k) + 2u * ntt_tw_v_int(j, G) + 1];
                } else {
                    ntt_load_pair(tws + 2 * ntt_tw_u_int(s, j, k) + 2u * ntt_tw_v_int(j, G), w[k], ws[k]);
                }
            }
#pragma unroll
            for (int e = 0; e < E; e++) {
                if ((e & half) == 0) {
                    const int k = e >> (RHO - j);
#if defined(LSA_NTT_DIAG_NO_TWIDDLE_LOADS)   // diagnostic build: constant twiddle (wrong results, same arithmetic)
                    const u64 wk = (u64)(G + 2 * k + 3), wsk = (u64)(G + 2 * k + 5) << 40;
#else
                    const u64 wk = w[k], wsk = ws[k];
#endif
#if defined(LSA_NTT_EXACT_BFLY)   // Harvey's form: values in [0, 4q), exact quotient
                    u64 U = csub(v[e], q2);
                    u64 T = shoup_mul_lazy(v[e + half], wk, wsk, q);
                    v[e] = U + T;
                    v[e + half] = sub64(U + q2, T);
#else                             // values in [0, 8q), product in [0, 4q) (modarith.h, shoup_mul_approx)
                    u64 T = shoup_mul_approx(v[e + half], wk, wsk, nq);
                    u64 U;
                    if (lazy) {
                        // q < 2^57: no conditional subtraction at all.  The product takes ANY 64-bit operand, both outputs are
                        // below U + 4q: from inputs below 8q a transform of s stages ends below (8 + 4s) q <= 76 q < 2^64 for
                        // s <= 17; ONE reduction (the same product with w = 1) in the final store.
                        U = v[e];
                        LSA_EMU_CHECK(U < 0 - 2 * q4 && T < q4);
                    } else {
                        U = csub_sign(v[e], nq4);
                        LSA_EMU_CHECK(v[e] < 2 * q4 && U < q4 && T < q4);   // the range invariants, checked by the CPU replay
                    }
                    v[e] = U + T;
                    v[e + half] = sub64(U + q4, T);
#endif
                }
            }
        }
    } else {
#pragma unroll
        for (int j = RHO - 1; j >= 0; j--) {
            const int half = E >> (j + 1);
            const int s = s_base + j;
            if (j == 0 && scale_here) {   // last stage of the whole transform: N^-1 folded into both outputs
#pragma unroll
                for (int e = 0; e < half; e++) {
                    u64 U = v[e], V = v[e + half];   // both in [0,2q) (exact form) / [0,4q); the exact product lands in [0,2q)
                    v[e] = shoup_mul_lazy(U + V, sc0, sc0s, q);
#if defined(LSA_NTT_EXACT_BFLY)
                    v[e + half] = shoup_mul_lazy(sub64(U + q2, V), sc1, sc1s, q);
#else
                    v[e + half] = shoup_mul_lazy(sub64(U + q4, V), sc1, sc1s, q);
#endif
                }
                continue;
            }
            const u64* tws = tw + 2 * (1LL << s);
            u64 w[E / 2], ws[E / 2];
#if defined(LSA_NTT_DIAG_TW8)   // diagnostic build: 8-byte twiddles in the FP64 table's layout (wrong results, same arithmetic)
            if (!TWU) {
                double wd[E / 2];
                ntt_load_tw_fp(reinterpret_cast<const double*>(tw) + (1LL << s), s, j, G, wd);
#pragma unroll
                for (int k = 0; k < (1 << j); k++) {
                    w[k] = (u64)__builtin_bit_cast(long long, wd[k]);
                    ws[k] = (w[k] << 3) | 1;
                }
            } else
#endif
#pragma unroll
            for (int k = 0; k < (1 << j); k++) {
                if (TWU) {   // wave-uniform position: scalar loads
                    w[k] = LSA_CONST_PTR(u64, tws)[2 * ntt_tw_u_int(s, j, k) + 2u * ntt_tw_v_int(j, G)];
                    ws[k] = LSA_CONST_PTR(u64, tws)[2 * ntt_tw_u_int(s, j, k) + 2u * ntt_tw_v_int(j, G) + 1];
                } else {
                    ntt_load_pair(tws + 2 * ntt_tw_u_int(s, j, k) + 2u * ntt_tw_v_int(j, G), w[k], ws[k]);
                }
            }
#pragma unroll
            for (int e = 0; e < E; e++) {
                if ((e & half) == 0) {
                    const int k = e >> (RHO - j);
                    u64 U = v[e], V = v[e + half];
#if defined(LSA_NTT_DIAG_NO_TWIDDLE_LOADS)
                    const u64 wk = (u64)(G + 2 * k + 3), wsk = (u64)(G + 2 * k + 5) << 40;
#else
                    const u64 wk = w[k], wsk = ws[k];
#endif
#if defined(LSA_NTT_EXACT_BFLY)   // both in [0,2q)
                    v[e] = csub(U + V, q2);
                    v[e + half] = shoup_mul_lazy(sub64(U + q2, V), wk, wsk, q);
#else                             // both in [0,4q)
                    LSA_EMU_CHECK(U < q4 && V < q4);
                    v[e] = csub_sign(U + V, nq4);
                    v[e + half] = shoup_mul_approx(sub64(U + q4, V), wk, wsk, nq);
                    LSA_EMU_CHECK(v[e] < q4 && v[e + half] < q4);
#endif
                }
            }
        }
    }
}

// (the lazy form is its own instantiation behind a block-uniform branch: as a run-time flag inside one body the compiler
// turned it into selects and kept computing the conditional subtractions)
template <int RHO, bool TWU = false>
LSA_HD void ntt_group_int(u64 (&v)[1 << RHO], bool inverse, const u64* tw, int s_base, unsigned G, u64 q, bool scale_here, u64 sc0, u64 sc0s,
                          u64 sc1, u64 sc1s) {
    if (!inverse && ntt_int_lazy(q)) ntt_group_int_impl<RHO, TWU, true>(v, false, tw, s_base, G, q, false, 0, 0, 0, 0);
    else ntt_group_int_impl<RHO, TWU, false>(v, inverse, tw, s_base, G, q, scale_here, sc0, sc0s, sc1, sc1s);
}

// One radix-2^RHO sub-pass over local stages [sig0, sig0+RHO) of the pass — integer (Shoup) engine.
// LIN: the padded LDS addresses of a group's 2^RHO elements are an arithmetic progression (beta0 == 0 or >= 4), so one
// add per element replaces the shift/add padding arithmetic.
// TWL: the twiddles come from the block's LDS copy (bc.tw_l, indexed like the limb's global slice) instead of global memory
template <int RHO, bool LIN, int NT, bool TWL = false>
LSA_HD void ntt_phase_sub(const NttPassArgs& a, const NttBlockCtx& bc, int tid, u64* lds, int sig0) {
    constexpr int E = 1 << RHO;
    const ModDev md = a.mods[bc.mod];
    const u64* tw = TWL ? bc.tw_l : a.tw + (((long long)bc.mod << a.logn) << 1);   // {w, ws} pairs
    const int beta0 = a.lambda + a.mu - sig0 - RHO;  // lowest active bit of this sub-pass in l
    const int ngroups = 1 << (a.tau - RHO);
    const bool scale_here = a.inverse && a.apply_scale && a.s_lo + sig0 == 0;
    const u64* sc = a.scale + 4 * bc.mod;
    const u64 sc0 = scale_here ? sc[0] : 0, sc0s = scale_here ? sc[1] : 0, sc1 = scale_here ? sc[2] : 0, sc1s = scale_here ? sc[3] : 0;
    for (int gid = tid; gid < ngroups; gid += NT) {
        int lbase = ((gid >> beta0) << (beta0 + RHO)) | (gid & ((1 << beta0) - 1));
        const int ad0 = lds_addr(lbase), adst = !LIN ? 0 : beta0 ? (1 << beta0) + (1 << (beta0 - 4)) : 1;
        u64 v[E];
#pragma unroll
        for (int e = 0; e < E; e++) v[e] = lds[LIN ? ad0 + e * adst : lds_addr(lbase + (e << beta0))];
        // G = H * 2^sig0 + r_high ; r = (l >> lambda) & (2^mu - 1)
        int r = (lbase >> a.lambda) & ((1 << a.mu) - 1);
        int H = ntt_hi_index(a, bc.tile, lbase);
        int G = (H << sig0) + (r >> (a.mu - sig0));
        ntt_group_int<RHO>(v, a.inverse != 0, tw, a.s_lo + sig0, (unsigned)G, md.q, scale_here, sc0, sc0s, sc1, sc1s);
#pragma unroll
        for (int e = 0; e < E; e++) lds[LIN ? ad0 + e * adst : lds_addr(lbase + (e << beta0))] = v[e];
    }
}

// The same butterfly group on the FP64 engine: integer-valued doubles.  long_pass: a (single-pass) transform of more than 9
// stages reduces after every forward group; the inverse always does.
template <int RHO, bool TWU = false>
LSA_HD void ntt_group_fp(double (&v)[1 << RHO], bool inverse, const double* tw /* limb's table */, int s_base, unsigned G, double q, double qinv,
                         bool scale_here, double sc0, double sc1, bool long_pass) {
    constexpr int E = 1 << RHO;
    if (!inverse) {
#pragma unroll
        for (int j = 0; j < RHO; j++) {
            const int half = E >> (j + 1);
            const int s = s_base + j;
            double w[E / 2];   // this stage's 2^j twiddles: one 8-byte (j = 0) or 2^(j-1) 16-byte loads
            if (TWU) {   // wave-uniform position: scalar loads
#pragma unroll
                for (int k = 0; k < (1 << j); k++) w[k] = LSA_CONST_PTR(double, tw + (1LL << s))[ntt_tw_pos_fp(s, j, G, k)];
            } else {
                ntt_load_tw_fp(tw + (1LL << s), s, j, G, w);
            }
#pragma unroll
            for (int e = 0; e < E; e++) {
                if ((e & half) == 0) {
#if defined(LSA_NTT_DIAG_NO_TWIDDLE_LOADS)   // diagnostic build: constant twiddle (wrong results, same arithmetic)
                    const double T = fp_modmul(v[e + half], (double)(G + j + 3), q, qinv);
#else
                    const double T = fp_modmul(v[e + half], w[e >> (RHO - j)], q, qinv);
#endif
                    const double U = v[e];
                    LSA_EMU_CHECK(__builtin_fabs(U) < 2251799813685248.0 && __builtin_fabs(v[e + half]) < 2251799813685248.0);   // 2^51: the engine's exactness bound
                    v[e] = U + T;          // |.| grows by < 1.1q per stage: <= 10.9q < 2^51 over a 9-stage pass
                    v[e + half] = U - T;
                }
            }
        }
        if (long_pass) {   // longer (single-pass) transforms: back to |.| <= q/2+1 after every sub-pass (< 4q + 4.4q inside one)
#pragma unroll
            for (int e = 0; e < E; e++) v[e] = fp_reduce(v[e], q, qinv);
        }
    } else {
#pragma unroll
        for (int j = RHO - 1; j >= 0; j--) {
            const int half = E >> (j + 1);
            const int s = s_base + j;
            if (j == 0 && scale_here) {   // last stage of the whole transform: N^-1 folded into both outputs
#pragma unroll
                for (int e = 0; e < half; e++) {
                    const double U = v[e], V = v[e + half];
                    v[e] = fp_modmul(U + V, sc0, q, qinv);
                    v[e + half] = fp_modmul(U - V, sc1, q, qinv);
                }
                continue;
            }
            double w[E / 2];
            if (TWU) {   // wave-uniform position: scalar loads
#pragma unroll
                for (int k = 0; k < (1 << j); k++) w[k] = LSA_CONST_PTR(double, tw + (1LL << s))[ntt_tw_pos_fp(s, j, G, k)];
            } else {
                ntt_load_tw_fp(tw + (1LL << s), s, j, G, w);
            }
#pragma unroll
            for (int e = 0; e < E; e++) {
                if ((e & half) == 0) {
                    const double U = v[e], V = v[e + half];
                    v[e] = U + V;   // sums at most double per stage: < 16 * 1.1q inside a sub-pass
                    v[e + half] = fp_modmul(U - V, w[e >> (RHO - j)], q, qinv);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < E; e++) v[e] = fp_reduce(v[e], q, qinv);  // back to |.| <= q/2+1 before the next sub-pass
    }
}

// The same sub-pass on the FP64 engine: LDS holds integer-valued doubles.
template <int RHO, bool LIN, int NT, bool TWL = false>
LSA_HD void ntt_phase_sub_fp(const NttPassArgs& a, const NttBlockCtx& bc, int tid, u64* lds, int sig0) {
    constexpr int E = 1 << RHO;
    const double q = (double)a.mods[bc.mod].q, qinv = 1.0 / q;
    const double* tw = TWL ? reinterpret_cast<const double*>(bc.tw_l) : a.twd + ((long long)bc.mod << a.logn);
    const int beta0 = a.lambda + a.mu - sig0 - RHO;
    const int ngroups = 1 << (a.tau - RHO);
    const bool scale_here = a.inverse && a.apply_scale && a.s_lo + sig0 == 0;
    const double sc0 = scale_here ? a.scaled[2 * bc.mod] : 0.0, sc1 = scale_here ? a.scaled[2 * bc.mod + 1] : 0.0;
    for (int gid = tid; gid < ngroups; gid += NT) {
        int lbase = ((gid >> beta0) << (beta0 + RHO)) | (gid & ((1 << beta0) - 1));
        const int ad0 = lds_addr(lbase), adst = !LIN ? 0 : beta0 ? (1 << beta0) + (1 << (beta0 - 4)) : 1;
        double v[E];
#pragma unroll
        for (int e = 0; e < E; e++) v[e] = d_from_bits(lds[LIN ? ad0 + e * adst : lds_addr(lbase + (e << beta0))]);
        int r = (lbase >> a.lambda) & ((1 << a.mu) - 1);
        int H = ntt_hi_index(a, bc.tile, lbase);
        int G = (H << sig0) + (r >> (a.mu - sig0));
        ntt_group_fp<RHO>(v, a.inverse != 0, tw, a.s_lo + sig0, (unsigned)G, q, qinv, scale_here, sc0, sc1, a.mu > 9);
#pragma unroll
        for (int e = 0; e < E; e++) lds[LIN ? ad0 + e * adst : lds_addr(lbase + (e << beta0))] = d_to_bits(v[e]);
    }
}

// sub-pass plan: mu stages as ceil(mu/MAX_RHO) radix groups in local stage order.  Up to 12 stages: nearly equal groups.
// Longer (single-pass N = 2^13 / 2^14) transforms: full radix-16 groups with the remainder second to last, which keeps
// every group's lowest active bit at 0 or >= 4 (the conflict-free linear LDS addressing of ntt_phase_sub).
LSA_HD int ntt_split(int mu, int* rho /*[4]*/) {
    const int n = (mu + LSA_NTT_MAX_RHO - 1) / LSA_NTT_MAX_RHO;
    if ((mu > 12 || mu == 10) && LSA_NTT_MAX_RHO == 4) {   // 10 = 4 + 2 + 4 (4 + 3 + 3 would put a group's lowest bit at 3)
        for (int i = 0; i < n; i++) rho[i] = 4;
        rho[n - 2] = mu - 4 * (n - 1);
        return n;
    }
    const int base = mu / n, extra = mu % n;
    for (int i = 0; i < n; i++) rho[i] = base + (i < extra ? 1 : 0);
    return n;
}

template <bool LIN, int NT, bool TWL = false>
LSA_HD void ntt_phase_sub_sel(const NttPassArgs& a, const NttBlockCtx& bc, int tid, u64* lds, int sig0, int rho) {
    if (bc.fp) {
        switch (rho) {
            case 1: ntt_phase_sub_fp<1, LIN, NT, TWL>(a, bc, tid, lds, sig0); break;
            case 2: ntt_phase_sub_fp<2, LIN, NT, TWL>(a, bc, tid, lds, sig0); break;
#if LSA_NTT_MAX_RHO >= 4
            case 3: ntt_phase_sub_fp<3, LIN, NT, TWL>(a, bc, tid, lds, sig0); break;
            default: ntt_phase_sub_fp<4, LIN, NT, TWL>(a, bc, tid, lds, sig0); break;
#elif LSA_NTT_MAX_RHO == 3
            default: ntt_phase_sub_fp<3, LIN, NT, TWL>(a, bc, tid, lds, sig0); break;
#else
            default: break;
#endif
        }
        return;
    }
    switch (rho) {
        case 1: ntt_phase_sub<1, LIN, NT, TWL>(a, bc, tid, lds, sig0); break;
        case 2: ntt_phase_sub<2, LIN, NT, TWL>(a, bc, tid, lds, sig0); break;
#if LSA_NTT_MAX_RHO >= 4
        case 3: ntt_phase_sub<3, LIN, NT, TWL>(a, bc, tid, lds, sig0); break;
        default: ntt_phase_sub<4, LIN, NT, TWL>(a, bc, tid, lds, sig0); break;
#elif LSA_NTT_MAX_RHO == 3
        default: ntt_phase_sub<3, LIN, NT, TWL>(a, bc, tid, lds, sig0); break;
#else
        default: break;
#endif
    }
}

template <int NT, bool TWL = false>
LSA_HD void ntt_phase_sub_dyn(const NttPassArgs& a, const NttBlockCtx& bc, int tid, u64* lds, int sig0, int rho) {
    const int beta0 = a.lambda + a.mu - sig0 - rho;
    if (beta0 == 0 || beta0 >= 4)
        ntt_phase_sub_sel<true, NT, TWL>(a, bc, tid, lds, sig0, rho);
    else
        ntt_phase_sub_sel<false, NT, TWL>(a, bc, tid, lds, sig0, rho);
}
