// kernels.hip — hand-written gfx950 kernels for the RNS hot path and their launchers.
// HBM-bound integer work: 16 B/lane coalesced accesses, LDS-staged butterflies, no MFMA (no dense FP contraction here).
#include <atomic>
#include <cstdlib>

#include "build_flags.h"
#include "lsa_internal.h"
#include "ntt_r16.h"

namespace lsa {

const char* kernels_build_flags() { return LSA_BUILD_FLAGS_TEXT; }

static constexpr int TPB = 256;

// ------------------------------------------------------------------------------------------------ K1/K2 NTT pass
#ifndef LSA_NTT_WAVES
#define LSA_NTT_WAVES 4   // min waves/SIMD the register allocator must allow (= co-resident 256-thread workgroups per CU):
                          // 4 = at most 128 VGPRs; the plain and epilogue-only variants fit without spilling
#endif

#if defined(LSA_NTT_DIAG_STAMPS)   // diagnostic build: shader-clock stamp k of this workgroup (first 8192 workgroups)
#define LSA_STAMP(k)                                                                               \
    do {                                                                                           \
        if (a.diag && threadIdx.x == 0 && blockIdx.x < 8192) a.diag[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define LSA_STAMP(k) do { } while (0)
#endif

template <int NT, bool TWL = false>
__device__ __forceinline__ void ntt_butterfly_phases(const NttPassArgs& a, const NttBlockCtx& bc, int tid, u64* lds) {
    int rho[4];
    const int np = ntt_split(a.mu, rho);
    if (!a.inverse) {
        int sig = 0;
        for (int i = 0; i < np; i++) {
            ntt_phase_sub_dyn<NT, TWL>(a, bc, tid, lds, sig, rho[i]);
            LSA_STAMP(3 + 2 * i);
            __syncthreads();
            LSA_STAMP(4 + 2 * i);
            sig += rho[i];
        }
    } else {
        int sig = a.mu;
        for (int i = np - 1; i >= 0; i--) {
            sig -= rho[i];
            ntt_phase_sub_dyn<NT, TWL>(a, bc, tid, lds, sig, rho[i]);
            __syncthreads();
        }
    }
}

// FZ: the launch carries a fused prologue/epilogue (NttPassArgs::fz_*)
#ifndef LSA_NTT_WAVES_FUSED
#define LSA_NTT_WAVES_FUSED 3   // variants with the two-operand prologue (FZ & 1): 3 workgroups per CU, up to 168 VGPRs
#endif
// NT: workgroup size = tile points / 16.  256 threads (4096-point tiles, 3-4 workgroups per CU) is the two-pass shape;
// 512 / 1024 threads hold a whole N = 2^13 / 2^14 limb in LDS (69 / 136 KiB) and transform it in ONE pass: half the HBM
// traffic of the two-pass plan, at one or two workgroups per CU.
// TWL: the pass holds global stage 0 (strided first pass of a two-pass plan): its 2^mu twiddles are the same for every tile
// of the limb and go through LDS -- one fetch per workgroup, issued with the tile loads, instead of one per sub-pass.
// FZ: 0 plain, 1 fused prologue only (first pass of a two-pass fused transform), 2 fused epilogue only (its last pass), 3 both
// (single-pass transforms).  Split so that a pass carries only the tail code and registers it can execute.
template <int FZ, int NT, bool TWL = false>
__global__ __launch_bounds__(NT, NT > 512 ? 1 : NT > 256 ? 2 : (FZ & 1) ? LSA_NTT_WAVES_FUSED : LSA_NTT_WAVES) void k_ntt_pass(NttPassArgs a) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int tid = threadIdx.x;
    NttBlockCtx bc = ntt_decode_block(a, (long long)blockIdx.x);
    if (bc.mod == LSA_ROW_SKIP) return;  // uniform per block, before any barrier
    ulonglong2 twp = {0, 0};
    if (TWL) {
        bc.tw_l = lds + lds_words(a.tau);
        if (tid < (1 << a.mu)) {
            if (bc.fp) twp.x = d_to_bits(a.twd[((long long)bc.mod << a.logn) + tid]);
            else twp = *reinterpret_cast<const ulonglong2*>(a.tw + (((long long)bc.mod << a.logn) << 1) + 2 * tid);
        }
    }
#if defined(LSA_NTT_STAGGER)
    // Workgroups of one launch have identical phase lengths and start together, so the 3-4 co-resident ones run their
    // HBM phase and their butterfly phase in lockstep (measured: kernel time == copy-only time + compute-only time).
    // Delay the workgroups that share a CU (blockIdx 256 apart in the first dispatch wave) by thirds of a tile period once.
    {
        const int phase = blockIdx.x < 768 ? (int)(blockIdx.x >> 8) : 0;   // first dispatch wave only
        for (int i = 0; i < phase * LSA_NTT_STAGGER; i++) __builtin_amdgcn_s_sleep(100);
    }
#endif
#if defined(LSA_NTT_DIAG_COMPUTE_ONLY)   // diagnostic build: butterflies on synthetic LDS contents, no global traffic
    for (int i = tid; i < (1 << a.tau); i += NT)
        lds[lds_addr(i)] = bc.fp ? d_to_bits((double)(i * 7 + 1)) : (u64)(i * 7 + 1);
    __syncthreads();
    ntt_butterfly_phases<NT>(a, bc, tid, lds);
    if (lds[lds_addr(tid)] == 0x123456789abcull) a.dst[bc.base_dst + tid] = 1;   // keeps the work alive
    return;
#endif
    LSA_STAMP(0);
    ntt_phase_load<(FZ & 1) != 0, NT>(a, bc, tid, lds);
    if (TWL && tid < (1 << a.mu)) {
        u64* tw_l = lds + lds_words(a.tau);
        if (bc.fp) tw_l[tid] = twp.x;
        else {
            tw_l[2 * tid] = twp.x;
            tw_l[2 * tid + 1] = twp.y;
        }
    }
    LSA_STAMP(1);
    __syncthreads();
    LSA_STAMP(2);
#if defined(LSA_NTT_DIAG_COPY_ONLY)   // diagnostic build: data movement of the pass structure without butterflies
    ntt_phase_store<(FZ & 2) != 0, NT>(a, bc, tid, lds);
    return;
#endif
    ntt_butterfly_phases<NT, TWL>(a, bc, tid, lds);
    ntt_phase_store<(FZ & 2) != 0, NT>(a, bc, tid, lds);
    LSA_STAMP(7);
}

// ---- the 8-stage passes of two-pass plans: 16 x 16 register butterflies, one LDS exchange (ntt_r16.h)
// PASS 0 (first pass): the exchange crosses wavefronts -> one workgroup barrier.  PASS 1 (second pass): a 256-point transform
// lives in 16 lanes of one wavefront, LDS instructions of a wavefront execute in order: the exchanges need no barrier, only
// that the compiler keeps the LDS accesses in program order (wavefront-scope fences).
template <int PASS>
__device__ __forceinline__ void r16_sync() {
#if defined(LSA_R16_BARRIERS)   // A/B: workgroup barriers in the second pass too
    __syncthreads();
#else
    if (PASS == 0) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#endif
}
#ifndef LSA_R16_WAVES
#define LSA_R16_WAVES 4
#endif
template <int PASS, int FZ, int MU>
__global__ __launch_bounds__(LSA_R16_THREADS, (FZ & 1) ? LSA_NTT_WAVES_FUSED : LSA_R16_WAVES) void k_ntt_r16(NttPassArgs a) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int tid = threadIdx.x;
    const NttBlockCtx bc = ntt_decode_block(a, (long long)blockIdx.x);
    if (bc.mod == LSA_ROW_SKIP) return;  // uniform per block, before any barrier
    u64 v[16];
    r16_phase<PASS, FZ, MU>(a, bc, tid, lds, 0, v);
    if (!(PASS == 0 && a.inverse)) r16_sync<PASS>();   // (first pass, inverse: phase 0 only filled registers)
    r16_phase<PASS, FZ, MU>(a, bc, tid, lds, 1, v);
    if (PASS == 0 && !a.inverse) return;                // (first pass, forward: phase 1 stored the results)
    r16_sync<PASS>();
    r16_phase<PASS, FZ, MU>(a, bc, tid, lds, 2, v);
}
// the nine-stage second pass (N = 2^17 / 2^18): three radix-8 groups per point, two wave-local LDS exchanges (ntt_r16.h)
template <int FZ>
__global__ __launch_bounds__(LSA_R16_THREADS, LSA_R16_WAVES) void k_ntt_r8x3(NttPassArgs a) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int tid = threadIdx.x;
    const NttBlockCtx bc = ntt_decode_block(a, (long long)blockIdx.x);
    if (bc.mod == LSA_ROW_SKIP) return;
    u64 v[16];
    r8x3_phase<FZ>(a, bc, tid, lds, 0, v);
    r16_sync<1>();
    r8x3_phase<FZ>(a, bc, tid, lds, 1, v);
    r16_sync<1>();
    r8x3_phase<FZ>(a, bc, tid, lds, 2, v);
    r16_sync<1>();
    r8x3_phase<FZ>(a, bc, tid, lds, 3, v);
}
static bool ntt_launch_r16(const NttPassArgs& a, int npass, bool fused, long long nblocks, hipStream_t s) {
    static const bool enabled = [] {
        const char* e = getenv("LSA_NTT_R16");
        return !(e && e[0] == '0');
    }();
    if (!enabled || !ntt_r16_shape_ok(a, npass)) return false;
    const bool pro = fused && a.fz_pro && a.s_lo == 0, epi = fused && a.fz_epi && a.final_reduce;
    const size_t lds_bytes = (size_t)LSA_R16_LDS_WORDS * sizeof(u64);
    const dim3 grid((unsigned)nblocks), block(LSA_R16_THREADS);
    // the two-operand prologue: 394 us per headline launch here against 450 on the staged kernel once the lift's block-uniform
    // cases became branches (628 before: both lifts were evaluated per element, profiles/r03/ab_r16_prologue_epilogue_branches.log);
    // LSA_R16_PRO=0 keeps it on the staged kernel (A/B)
    static const bool pro_enabled = [] {
        const char* e = getenv("LSA_R16_PRO");
        return !(e && e[0] == '0');
    }();
    if (a.lambda) {
        if (epi || (pro && !pro_enabled)) return false;   // (a first pass is never the last one of a two-pass plan)
        if (a.mu == 8) {
            if (pro) hipLaunchKernelGGL((k_ntt_r16<0, 1, 8>), grid, block, lds_bytes, s, a);
            else hipLaunchKernelGGL((k_ntt_r16<0, 0, 8>), grid, block, lds_bytes, s, a);
        } else {
            if (pro) return false;
            hipLaunchKernelGGL((k_ntt_r16<0, 0, 7>), grid, block, lds_bytes, s, a);
        }
    } else {
        if (pro) return false;
        if (a.mu == 9) {
            static const bool r8x3_enabled = [] {
                const char* e = getenv("LSA_NTT_R8X3");   // =0: the nine-stage second pass on the staged kernel (A/B)
                return !(e && e[0] == '0');
            }();
            if (!r8x3_enabled) return false;
            if (epi) hipLaunchKernelGGL((k_ntt_r8x3<2>), grid, block, lds_bytes, s, a);
            else hipLaunchKernelGGL((k_ntt_r8x3<0>), grid, block, lds_bytes, s, a);
        } else if (a.mu == 8) {
            if (epi) hipLaunchKernelGGL((k_ntt_r16<1, 2, 8>), grid, block, lds_bytes, s, a);
            else hipLaunchKernelGGL((k_ntt_r16<1, 0, 8>), grid, block, lds_bytes, s, a);
        } else {
            if (epi) hipLaunchKernelGGL((k_ntt_r16<1, 2, 7>), grid, block, lds_bytes, s, a);
            else hipLaunchKernelGGL((k_ntt_r16<1, 0, 7>), grid, block, lds_bytes, s, a);
        }
    }
    LSA_HIP(hipGetLastError());
    return true;
}

// ---- second pass of the extension transform + key MAC in one launch (ntt_r16.h, "second pass + key MAC")
// grid: (tile, tl, b) with b fastest: consecutive workgroups are consecutive ciphertexts at one (limb, tile) -- they read the
// same key tile within microseconds of each other -- and a CU's resident workgroups mix integer- and FP64-engine limbs
#ifndef LSA_KSMAC_WAVES
#define LSA_KSMAC_WAVES 2   // 32 running sums (64 VGPRs) stay live across the transform phases: at 3 workgroups per CU (168 VGPRs) 76 registers spill
#endif
#ifndef LSA_KSMAC_WAVES_FP
#define LSA_KSMAC_WAVES_FP LSA_KSMAC_WAVES
#endif
template <int MU, bool FP>
__global__ __launch_bounds__(LSA_R16_THREADS, FP ? LSA_KSMAC_WAVES_FP : LSA_KSMAC_WAVES) void k_ntt_r16_ksmac(KsFusedArgs g) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int tid = threadIdx.x;
    const int T = g.L + g.np;
    long long bid = blockIdx.x;
    long long b;
    if (g.xcd_deal) {   // consecutive workgroup ids go round-robin over the 8 XCDs: the batch of one key tile stays on one
        const long long slot = bid >> 3;
        b = slot % g.batch;
        bid = (slot / g.batch) * 8 + (bid & 7);
    } else {
        b = bid % g.batch;
        bid /= g.batch;
    }
    const int tl = g.tl_list[bid % g.n_tl], tile = (int)(bid / g.n_tl);
    NttPassArgs a;   // only the scalar fields the pass functions read are set (the row tables are never indexed here)
    a.logn = g.logn;
    a.s_lo = g.logn - MU;
    a.mu = MU;
    a.lambda = 0;
    a.tau = 12;
    a.inverse = 0;
    a.apply_scale = 0;
    a.final_reduce = 1;
    a.fp_raw_in = g.fp_raw_in;
    a.fp_raw_out = 0;
    a.mods = g.mods;
    a.tw = g.tw;
    a.twd = g.twd;
    a.scale = nullptr;
    a.scaled = nullptr;
    a.src = g.ext;
    a.fz_pro = a.fz_epi = 0;
    NttBlockCtx bc;
    bc.tile = tile;
    bc.mod = tl < g.L ? tl : g.nq + (tl - g.L);
    bc.fp = FP;   // (the launcher put this limb on the launch of its engine)
    bc.b = (int)b;
    bc.row = 0;
    bc.tw_l = nullptr;
    bc.base_dst = 0;
    const R16Limb L = r16_limb(a, bc);
    const int kj = tl < g.L ? tl : g.klvl + 1 + (tl - g.L);
    const int own_d = tl < g.L ? tl / g.np : -1;
    int k, i;
    r16_lane<1, MU>(tid, k, i);
    const unsigned G1 = r16_G1<1, MU>(bc, k), G2 = (G1 << 4) + ((unsigned)i << (MU <= 8 ? 8 - MU : 0));   // (G2: 8- and 7-stage passes only)
    u64 acc0[16], acc1[16];
#pragma unroll
    for (int e = 0; e < 16; e++) acc0[e] = acc1[e] = 0;   // (0 is also +0.0)
    for (int d = 0; d < g.beta; d++) {
        const bool own = d == own_d;
        if (!own) {
            bc.base_src = b * g.sext + ((long long)(d * T + tl) << g.logn);
            bc.row = d * T + tl;
            u64 v[16];
            r16_load_direct<1, false, 0, MU>(a, bc, tid, v);
            if constexpr (MU == 9) {   // nine stages: three radix-8 groups per point, two exchanges (ntt_r16.h)
                r8x3_group<0>(v, a, bc, L, G1, i);
                r8x3_put<0>(k, i, lds, v);
                r16_sync<1>();
                r8x3_get<1>(k, i, lds, v);
                r8x3_group<1>(v, a, bc, L, G1, i);
                r8x3_put<1>(k, i, lds, v);
                r16_sync<1>();
                r8x3_get<2>(k, i, lds, v);
                r8x3_group<2>(v, a, bc, L, G1, i);
                r8x3_put<2>(k, i, lds, v);
            } else {
                r16_group<1, 0, MU>(v, a, bc, L, G1);
                r16_lds_put<1, 0, MU>(tid, lds, v);
                r16_sync<1>();
                r16_lds_get<1, 1, MU>(tid, lds, v);
                r16_group<1, 1, MU>(v, a, bc, L, G2);
                r16_lds_put<1, 1, MU>(tid, lds, v);
            }
            r16_sync<1>();
        }
        r16_mac_digit<MU, FP>(g, a, bc, L, tid, d, kj, own, b, tl, lds, acc0, acc1);
        r16_sync<1>();   // the next digit's first exchange overwrites the tile image these lanes just read
    }
    r16_mac_store<MU, FP>(g, a, bc, L, tid, b, tl, acc0, acc1);
}

bool ks_fused_enabled(const Context& c) {
    const char* e = getenv("LSA_KS_FUSED");   // read per call: the parity tests flip it inside one process (a key uploaded either
    const bool on = !(e && e[0] == '0');      // way works with either setting: the fused path needs key.fp and is skipped without it)
    if (!on || c.plan.npass != 2 || !c.fp_raw) return false;
    const NttPassShape& p = c.plan.pass[1];
    return p.tau == 12 && p.lambda == 0 && (p.mu == 7 || p.mu == 8 || p.mu == 9) && p.s_lo == c.logn - p.mu;
}

int ks_fused_engines(const Context& c) {
    // integer-engine target limbs measured slower fused than apart (two lazy REDCs per product and digit, 256 VGPRs and still
    // spilling: 593 us against ~500 for pass + MAC on the headline's 5 integer limbs), FP64-engine limbs faster (643 us for 36
    // transforms + their MAC against ~1050): profiles/r03/ab_ks_fused_kernel_stats.log
    static const int e = [] {
        const char* v = getenv("LSA_KS_FUSED_ENGINES");
        return v ? atoi(v) & 3 : 2;
    }();
    return c.fp64_ntt ? e : 0;
}

bool launch_ntt_ksmac(Context& c, int level, const u64* cx, long long scx, u64* ext, long long sext, const Key& key, u64* acc,
                      long long sacc, int batch, hipStream_t s, int engines) {
    if (batch <= 0) return true;
    if (!ks_fused_enabled(c) || (c.fp64_ntt && !key.fp)) return false;
    LSA_REQUIRE(key.level >= level, "key-switch key exported at a lower level than the ciphertext");
    KsFusedArgs g{};
    g.ext = ext;
    g.cx = cx;
    g.key = key.data;
    g.keyd = key.fp;
    g.acc = acc;
    g.sext = sext;
    g.scx = scx;
    g.sacc = sacc;
    g.mods = c.d_mods;
    g.tw = c.d_psi;
    g.twd = c.d_psi_d;
    g.logn = c.logn;
    g.L = level + 1;
    g.np = c.np;
    g.nq = c.nq;
    g.beta = (g.L + c.np - 1) / c.np;
    g.klvl = key.level;
    g.kcomp = key.level + 1 + c.np;
    g.batch = batch;
    g.allow_fp64 = c.fp64_ntt;
    g.fp_raw_in = c.fp_raw;
    const int T = g.L + c.np, mu = c.plan.pass[1].mu;
    LSA_REQUIRE(T <= 64, "fused key MAC: too many target limbs");
    const size_t lds_bytes = (size_t)LSA_R16_LDS_WORDS * sizeof(u64);
    // one launch per butterfly engine (own register budget each); bytes: one pass of the launch's extension transforms
    // (16 N / 2 per limb) + its share of the gadget inner product as launch_ks_mac counts it
    for (int eng = 0; eng < 2; eng++) {
        g.n_tl = 0;
        int transforms = 0;
        for (int tl = 0; tl < T; tl++) {
            const int mi = tl < g.L ? tl : c.p_mod(tl - g.L);
            const bool fp = c.fp64_ntt && (c.T.mod[mi] >> LSA_FP64_MAX_BITS) == 0;
            if ((int)fp != eng || !((engines >> eng) & 1)) continue;
            g.tl_list[g.n_tl++] = (unsigned char)tl;
            transforms += g.beta - (tl < g.L ? 1 : 0);
        }
        if (!g.n_tl) continue;
        const long long nblocks = (long long)batch * g.n_tl * (1 << (c.logn - 12));
        LSA_REQUIRE(nblocks < (1LL << 31), "ntt: grid too large");
        const char* xcd_env = getenv("LSA_KSMAC_XCD");   // =0: plain batch-fastest order (A/B: +0.45 % headline with the deal, ab_ksmac_xcd_deal.log)
        const bool xcd = !(xcd_env && xcd_env[0] == '0');
        g.xcd_deal = xcd && ((g.n_tl << (c.logn - 12)) % 8 == 0) ? 1 : 0;
        const double ntt_bytes = 16.0 * c.n * transforms * batch / 2;
        ProfScope ps(c, PROF_NTT, ntt_bytes + 8.0 * c.n * g.n_tl * (batch * ((double)g.beta + 2.0) + 2.0 * g.beta), s, ntt_bytes);
        const dim3 grid((unsigned)nblocks), block(LSA_R16_THREADS);
        if (mu == 9 && eng) hipLaunchKernelGGL((k_ntt_r16_ksmac<9, true>), grid, block, lds_bytes, s, g);
        else if (mu == 9) hipLaunchKernelGGL((k_ntt_r16_ksmac<9, false>), grid, block, lds_bytes, s, g);
        else if (mu == 8 && eng) hipLaunchKernelGGL((k_ntt_r16_ksmac<8, true>), grid, block, lds_bytes, s, g);
        else if (mu == 8) hipLaunchKernelGGL((k_ntt_r16_ksmac<8, false>), grid, block, lds_bytes, s, g);
        else if (eng) hipLaunchKernelGGL((k_ntt_r16_ksmac<7, true>), grid, block, lds_bytes, s, g);
        else hipLaunchKernelGGL((k_ntt_r16_ksmac<7, false>), grid, block, lds_bytes, s, g);
    }
    LSA_HIP(hipGetLastError());
    return true;
}

template <int FZ, int NT>
static void ntt_launch_variant(const NttPassArgs& a, long long nblocks, size_t lds_bytes, hipStream_t s) {
    if (lds_bytes > 65536) {   // whole-limb tiles: opt in to more than 64 KiB of dynamic LDS, once per kernel instance AND device
        static std::atomic<unsigned long long> raised{0};   // bit d: done on device d (the attribute is per device)
        int dev = 0;
        LSA_HIP(hipGetDevice(&dev));
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(raised.load(std::memory_order_acquire) & bit)) {
            LSA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ntt_pass<FZ, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            raised.fetch_or(bit, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL((k_ntt_pass<FZ, NT>), dim3((unsigned)nblocks), dim3(NT), lds_bytes, s, a);
    LSA_HIP(hipGetLastError());
}

template <int NT>
static void ntt_launch_pass(const NttPassArgs& a, bool fused, long long nblocks, size_t lds_bytes, hipStream_t s) {
    LSA_REQUIRE((1 << a.tau) <= 2 * LSA_NTT_STAGE_PAIRS * NT, "ntt: tile larger than the staging registers");
    // which fused tail this pass can execute: the prologue lives in the pass that holds stage 0, the epilogue in the pass
    // that reduces and stores the final values
    const bool pro = fused && a.fz_pro && a.s_lo == 0, epi = fused && a.fz_epi && a.final_reduce;
    switch ((pro ? 1 : 0) | (epi ? 2 : 0)) {
        case 0: ntt_launch_variant<0, NT>(a, nblocks, lds_bytes, s); break;
        case 1: ntt_launch_variant<1, NT>(a, nblocks, lds_bytes, s); break;
        case 2: ntt_launch_variant<2, NT>(a, nblocks, lds_bytes, s); break;
        default: ntt_launch_variant<3, NT>(a, nblocks, lds_bytes, s); break;
    }
}

void launch_ntt(Context& c, const u64* src, u64* dst, int batch, long long batch_stride, int rows, const RowMap& rm,
                bool inverse, hipStream_t s) {
    launch_ntt(c, src, dst, batch, batch_stride, batch_stride, rows, rm, inverse, s);
}

void launch_ntt(Context& c, const u64* src, u64* dst, int batch, long long src_stride, long long dst_stride, int rows,
                const RowMap& rm, bool inverse, hipStream_t s, const NttFusion* fz, int passes) {
    if (batch <= 0 || rows <= 0) return;
    LSA_REQUIRE(rm.period >= 1 && rm.period <= LSA_MAX_PERIOD, "ntt: bad row-map period");
    NttPassArgs a{};
    a.batch = batch;
    a.rows = rows;
    a.mods = c.d_mods;
    a.scale = c.d_scale;
    a.diag = c.ntt_diag;
    a.scaled = c.d_scale_d;
    a.allow_fp64 = c.fp64_ntt;
    if (fz) {
        LSA_REQUIRE(!inverse, "fused tails exist for forward transforms only");
        a.fz_epi = fz->epi;
        a.fz_pro = fz->pro;
        a.fz_limbs = fz->limbs;
        a.fz_base_polys = fz->base_polys;
        a.fz_ql_mod = fz->ql_mod;
        a.fz_a = fz->a;
        a.fz_a_stride = fz->a_stride;
        a.fz_a_rpp = fz->a_rpp;
        a.fz_base = fz->base;
        a.fz_base_stride = fz->base_stride;
        a.fz_base_rpp = fz->base_rpp;
        a.fz_k = fz->k;
        a.fz_out = fz->out;
        a.fz_out_stride = fz->out_stride;
        a.fz_out_rpp = fz->out_rpp;
        a.fz_last = fz->last;
        a.fz_last_stride = fz->last_stride;
        a.fz_last_rpp = fz->last_rpp;
        a.fz_k2 = fz->k2;
        a.fz_scatter = fz->scatter;
        LSA_REQUIRE(fz->epi != 2 || fz->k2, "merged tail needs its second factor");
        LSA_REQUIRE(!fz->scatter || fz->epi, "a scattered store needs an epilogue");
    }
    a.period = rm.period;
    a.row0 = rm.row0;
    a.row_step = rm.row_step;
    for (int i = 0; i < rm.period; i++) {
        LSA_REQUIRE(rm.mod_of[i] == LSA_ROW_SKIP || rm.mod_of[i] < c.nmod, "ntt: modulus index out of range");
        a.mod_of[i] = rm.mod_of[i];
    }
    int active_rows = 0;
    for (int r = 0; r < rows; r++) active_rows += rm.mod_of[(rm.row0 + r * rm.row_step) % rm.period] != LSA_ROW_SKIP;
    if (active_rows == 0) return;
    // the grid covers the active rows only
    int launch_rows = rows;
    a.compact = 0;
    if (active_rows <= LSA_MAX_PERIOD && rm.row0 + (rows - 1) * rm.row_step < 65536) {
        a.compact = 1;
        int k = 0;
        for (int r = 0; r < rows; r++) {
            const int row = rm.row0 + r * rm.row_step;
            if (rm.mod_of[row % rm.period] != LSA_ROW_SKIP) a.row_tbl[k++] = (unsigned short)row;
        }
        launch_rows = active_rows;
    }
    a.rows = launch_rows;
    // Two-pass transforms: run both passes on a chunk of the batch that fits the 256 MiB Infinity Cache before moving
    // on, so the second pass reads what the first just wrote from the memory-side cache instead of HBM.
    // whole-limb single pass (N = 2^13 / 2^14) where it measured faster: launches that fill the chip at one (2^14) or two
    // (2^13) workgroups per CU, without fused tails (at 1024 threads their registers spill); at 2^14 only limbs of the
    // FP64 engine -- the integer engine is bound by its multiplies there, not by traffic (tools/probe_wide.py)
    bool wide = c.plan_wide.npass < c.plan.npass && c.wide_mode != 0;
    if (wide && c.wide_mode == 2) {
        bool all_fp = c.fp64_ntt != 0;
        for (int r = 0; r < rows && all_fp; r++) {
            const unsigned char m = rm.mod_of[(rm.row0 + r * rm.row_step) % rm.period];
            if (m != LSA_ROW_SKIP && (c.T.mods[m].q >> LSA_FP64_MAX_BITS) != 0) all_fp = false;
        }
        const long long limbs = (long long)active_rows * batch;
        wide = !fz && (c.logn == 13 ? limbs >= 1024 : (all_fp && limbs >= 512));
    }
    if (passes != 3) wide = false;   // a caller that runs the passes separately means the two-pass plan
    const NttPlan& plan = wide ? c.plan_wide : c.plan;
    a.tw = wide ? (inverse ? c.d_psiinv_w : c.d_psi_w) : (inverse ? c.d_psiinv : c.d_psi);
    a.twd = wide ? (inverse ? c.d_psiinv_d_w : c.d_psi_d_w) : (inverse ? c.d_psiinv_d : c.d_psi_d);
    {   // launches that mix both butterfly engines interleave their limbs (+1.4 % hmult / +2.2 % rotate NTT rate, profiles/r01/ab_row_inner.log)
        bool any_fp = false, any_int = false;
        for (int r = 0; r < rows; r++) {
            const unsigned char m = rm.mod_of[(rm.row0 + r * rm.row_step) % rm.period];
            if (m == LSA_ROW_SKIP) continue;
            const bool fp = c.fp64_ntt && (c.T.mods[m].q >> LSA_FP64_MAX_BITS) == 0;
            any_fp |= fp;
            any_int |= !fp;
        }
        a.row_inner = any_fp && any_int;
    }
    int chunk = batch;
    if (plan.npass > 1 && c.ntt_chunk_mib > 0) {
        const double per_item = 8.0 * c.n * std::max(active_rows, 1);
        chunk = (int)std::max(1.0, std::min((double)batch, c.ntt_chunk_mib * 1048576.0 / per_item));
    }
    for (int b0 = 0; b0 < batch; b0 += chunk) {
        const int nb = std::min(chunk, batch - b0);
        a.batch = nb;
        for (int step = 0; step < plan.npass; step++) {
            if (plan.npass == 2 && !((passes >> step) & 1)) continue;
            const int k = inverse ? plan.npass - 1 - step : step;
            ntt_fill_pass(a, plan, c.logn, k, inverse ? 1 : 0);
            a.fp_raw_out = plan.npass == 2 && step == 0 && c.fp_raw;
            a.fp_raw_in = plan.npass == 2 && step == 1 && c.fp_raw;
            a.src = (step == 0 ? src + (long long)b0 * src_stride : dst + (long long)b0 * dst_stride);
            a.src_stride = step == 0 ? src_stride : dst_stride;
            a.dst = dst + (long long)b0 * dst_stride;
            a.dst_stride = dst_stride;
            a.total_tiles = (long long)nb * launch_rows * (1 << (a.logn - a.tau));
            const long long nblocks = a.total_tiles;
            LSA_REQUIRE(nblocks < (1LL << 31), "ntt: grid too large");
            const size_t lds_bytes = (size_t)lds_words(a.tau) * sizeof(u64);
            // one launch = one pass = 1/npass of the limb transforms it touches (algorithmic 16*N bytes per transform)
            ProfScope ps(c, PROF_NTT, 16.0 * c.n * active_rows * nb / plan.npass, s);
            if (ntt_launch_r16(a, plan.npass, fz != nullptr, nblocks, s)) continue;
            if (a.tau <= 12) ntt_launch_pass<LSA_NTT_THREADS>(a, fz != nullptr, nblocks, lds_bytes, s);
            else if (a.tau == 13) ntt_launch_pass<512>(a, fz != nullptr, nblocks, lds_bytes, s);
            else ntt_launch_pass<1024>(a, fz != nullptr, nblocks, lds_bytes, s);
        }
    }
}

// ------------------------------------------------------------------------------------------------ elementwise (K3/K4)
struct EwArgs {
    const u64* a;
    const u64* b;
    const u64* acc;   // EW_MUL only: optional accumulator operand
    u64* out;
    long long sa, sb, so, sacc;
    const ModDev* mods;
    int rows, logn, op, period;
    unsigned char mod_of[LSA_MAX_PERIOD];
};

__device__ __forceinline__ ulonglong2 ld2(const u64* p) { return *reinterpret_cast<const ulonglong2*>(p); }
__device__ __forceinline__ void st2(u64* p, u64 x, u64 y) {
    ulonglong2 v;
    v.x = x;
    v.y = y;
    *reinterpret_cast<ulonglong2*>(p) = v;
}

// grid: x = rows * (N/2/TPB), y = batch
__global__ __launch_bounds__(TPB) void k_elementwise(EwArgs g) {
    const int chunks = (1 << g.logn) / (2 * TPB);
    const int row = blockIdx.x / chunks;
    const int x = ((blockIdx.x % chunks) * TPB + threadIdx.x) * 2;
    const int mi = g.mod_of[row % g.period];
    if (mi == LSA_ROW_SKIP) return;
    const ModDev m = g.mods[mi];
    const long long off = ((long long)row << g.logn) + x;
    const long long b = blockIdx.y;
    const ulonglong2 va = ld2(g.a + b * g.sa + off);
    u64 r0, r1;
    if (g.op == EW_NEG) {
        r0 = neg_mod(va.x, m.q);
        r1 = neg_mod(va.y, m.q);
    } else {
        const ulonglong2 vb = ld2(g.b + b * g.sb + off);
        if (g.op == EW_ADD) {
            r0 = add_mod(va.x, vb.x, m.q);
            r1 = add_mod(va.y, vb.y, m.q);
        } else if (g.op == EW_SUB) {
            r0 = sub_mod(va.x, vb.x, m.q);
            r1 = sub_mod(va.y, vb.y, m.q);
        } else {
            r0 = mul_mod(va.x, vb.x, m);
            r1 = mul_mod(va.y, vb.y, m);
            if (g.acc) {   // multiply-accumulate: out = acc + a*b
                const ulonglong2 vc = ld2(g.acc + b * g.sacc + off);
                r0 = add_mod(r0, vc.x, m.q);
                r1 = add_mod(r1, vc.y, m.q);
            }
        }
    }
    st2(g.out + b * g.so + off, r0, r1);
}

static void fill_rowmap(unsigned char* dst, int& period, const RowMap& rm, int nmod) {
    LSA_REQUIRE(rm.period >= 1 && rm.period <= LSA_MAX_PERIOD, "bad row-map period");
    period = rm.period;
    for (int i = 0; i < rm.period; i++) {
        LSA_REQUIRE(rm.mod_of[i] == LSA_ROW_SKIP || rm.mod_of[i] < nmod, "modulus index out of range");
        dst[i] = rm.mod_of[i];
    }
}

static dim3 ew_grid(const Context& c, int rows, int batch) {
    LSA_REQUIRE(c.n >= 2 * TPB, "ring degree too small for the elementwise kernels (need N >= 512)");
    return dim3((unsigned)(rows * (c.n / (2 * TPB))), (unsigned)batch);
}

void launch_elementwise(Context& c, EwOp op, const u64* a, const u64* b, u64* out, int batch, long long sa, long long sb,
                        long long so, int rows, const RowMap& rm, hipStream_t s) {
    launch_muladd(c, op, a, b, nullptr, 0, out, batch, sa, sb, so, rows, rm, s);
}

// out = op(a, b) (+ acc for EW_MUL when acc != nullptr)
void launch_muladd(Context& c, EwOp op, const u64* a, const u64* b, const u64* acc, long long sacc, u64* out, int batch,
                   long long sa, long long sb, long long so, int rows, const RowMap& rm, hipStream_t s) {
    if (batch <= 0 || rows <= 0) return;
    EwArgs g{};
    g.acc = acc;
    g.sacc = sacc;
    g.a = a;
    g.b = b;
    g.out = out;
    g.sa = sa;
    g.sb = sb;
    g.so = so;
    g.mods = c.d_mods;
    g.rows = rows;
    g.logn = c.logn;
    g.op = op;
    fill_rowmap(g.mod_of, g.period, rm, c.nmod);
    ProfScope ps(c, PROF_ELEMWISE, 24.0 * c.n * rows * batch, s);
    hipLaunchKernelGGL(k_elementwise, ew_grid(c, rows, batch), dim3(TPB), 0, s, g);
    LSA_HIP(hipGetLastError());
}

// out[b][row] = a[b][row] * mvec[row] + kvec[row]: a real constant added to every slot is the same residue in every NTT
// coefficient of c0 (kvec: plain residues, one per row; mvec: optional Montgomery-form factors, null = 1)
struct AddConstArgs {
    const u64* a;
    u64* out;
    const u64* kvec;
    const u64* mvec;
    long long sa, so;
    const ModDev* mods;
    int logn;
    unsigned char mod_of[LSA_MAX_PERIOD];
};
__global__ __launch_bounds__(TPB) void k_add_const(AddConstArgs g) {
    const int chunks = (1 << g.logn) / (2 * TPB);
    const int row = blockIdx.x / chunks;
    const int x = ((blockIdx.x % chunks) * TPB + threadIdx.x) * 2;
    const ModDev m = g.mods[g.mod_of[row]];
    const u64 k = g.kvec[row];
    const long long b = blockIdx.y, off = ((long long)row << g.logn) + x;
    ulonglong2 v = ld2(g.a + b * g.sa + off);
    if (g.mvec) {
        const u64 f = g.mvec[row];
        v.x = mont_mul(v.x, f, m.q, m.qinv);
        v.y = mont_mul(v.y, f, m.q, m.qinv);
    }
    st2(g.out + b * g.so + off, add_mod(v.x, k, m.q), add_mod(v.y, k, m.q));
}
void launch_add_const(Context& c, const u64* a, long long sa, const u64* kvec, u64* out, long long so, int rows,
                      const RowMap& rm, int batch, hipStream_t s, const u64* mvec) {
    if (batch <= 0 || rows <= 0) return;
    LSA_REQUIRE(rm.period == rows && rows <= LSA_MAX_PERIOD, "add_const: row map must cover the rows");
    AddConstArgs g{};
    g.a = a;
    g.out = out;
    g.kvec = kvec;
    g.mvec = mvec;
    g.sa = sa;
    g.so = so;
    g.mods = c.d_mods;
    g.logn = c.logn;
    int period;
    fill_rowmap(g.mod_of, period, rm, c.nmod);
    ProfScope ps(c, PROF_ELEMWISE, 16.0 * c.n * rows * batch, s);
    hipLaunchKernelGGL(k_add_const, ew_grid(c, rows, batch), dim3(TPB), 0, s, g);
    LSA_HIP(hipGetLastError());
}

// ciphertext x plaintext multiply-accumulate over up to LSA_MAC_MAX_TERMS terms in ONE launch (cmp_sum / cmpac_sum nodes,
// mega_ag_executors_gpu.cu:294-408 does multiply_plain + add_inplace per term):
//   out[p][j] = (partial[p][j]) + sum_i ct_i[p][j] * pt_i[j]
// The products are summed as 128-bit integers and reduced once (REDC, then * R^2), folded every 8 terms.
struct MacPlainArgs {
    const u64* ct[LSA_MAC_MAX_TERMS];
    const u64* pt[LSA_MAC_MAX_TERMS];
    long long sct[LSA_MAC_MAX_TERMS], spt[LSA_MAC_MAX_TERMS];
    const u64* partial;
    u64* out;
    long long spartial, so;
    const ModDev* mods;
    int terms, polys, limbs, logn;
    unsigned char mod_of[LSA_MAX_PERIOD];
};

// grid: x = polys*limbs*(N/2/TPB), y = batch
__global__ __launch_bounds__(TPB) void k_mac_plain(MacPlainArgs g) {
    const int chunks = (1 << g.logn) / (2 * TPB);
    const int row = blockIdx.x / chunks;             // poly * limbs + limb
    const int limb = row % g.limbs;
    const int x = ((blockIdx.x % chunks) * TPB + threadIdx.x) * 2;
    const ModDev m = g.mods[g.mod_of[limb]];
    const long long b = blockIdx.y;
    const long long coff = ((long long)row << g.logn) + x, poff = ((long long)limb << g.logn) + x;
    u64 h0 = 0, l0 = 0, h1 = 0, l1 = 0, r0 = 0, r1 = 0;
    for (int i = 0; i < g.terms; i++) {
        const ulonglong2 c = ld2(g.ct[i] + b * g.sct[i] + coff);
        const ulonglong2 w = ld2(g.pt[i] + b * g.spt[i] + poff);
        mac128(h0, l0, c.x, w.x);
        mac128(h1, l1, c.y, w.y);
        if ((i & 7) == 7) {   // keep the 128-bit sum below q*2^64 (8 products of < q^2, q < 2^61)
            r0 = add_mod(r0, csub(mont_redc_lazy(h0, l0, m.q, m.qinv), m.q), m.q);
            r1 = add_mod(r1, csub(mont_redc_lazy(h1, l1, m.q, m.qinv), m.q), m.q);
            h0 = l0 = h1 = l1 = 0;
        }
    }
    r0 = add_mod(r0, csub(mont_redc_lazy(h0, l0, m.q, m.qinv), m.q), m.q);
    r1 = add_mod(r1, csub(mont_redc_lazy(h1, l1, m.q, m.qinv), m.q), m.q);
    r0 = mont_mul(r0, m.r2, m.q, m.qinv);            // sum * R^-1 -> sum
    r1 = mont_mul(r1, m.r2, m.q, m.qinv);
    if (g.partial) {
        const ulonglong2 v = ld2(g.partial + b * g.spartial + coff);
        r0 = add_mod(r0, v.x, m.q);
        r1 = add_mod(r1, v.y, m.q);
    }
    st2(g.out + b * g.so + coff, r0, r1);
}

void launch_mac_plain(Context& c, int terms, const u64* const* ct, const long long* sct, const u64* const* pt,
                      const long long* spt, const u64* partial, long long spartial, u64* out, long long so, int batch,
                      int polys, int limbs, const RowMap& rm, hipStream_t s) {
    if (batch <= 0 || terms <= 0) return;
    LSA_REQUIRE(terms <= LSA_MAC_MAX_TERMS, "too many terms for one multiply-accumulate launch");
    LSA_REQUIRE(rm.period == limbs && limbs <= LSA_MAX_PERIOD, "mac: row map must cover the limbs");
    MacPlainArgs g{};
    for (int i = 0; i < terms; i++) {
        g.ct[i] = ct[i];
        g.sct[i] = sct[i];
        g.pt[i] = pt[i];
        g.spt[i] = spt[i];
    }
    g.partial = partial;
    g.spartial = spartial;
    g.out = out;
    g.so = so;
    g.mods = c.d_mods;
    g.terms = terms;
    g.polys = polys;
    g.limbs = limbs;
    g.logn = c.logn;
    int period;
    fill_rowmap(g.mod_of, period, rm, c.nmod);
    ProfScope ps(c, PROF_ELEMWISE, 8.0 * c.n * limbs * batch * ((double)terms * (polys + 1) + polys * (partial ? 2 : 1)), s);
    hipLaunchKernelGGL(k_mac_plain, ew_grid(c, polys * limbs, batch), dim3(TPB), 0, s, g);
    LSA_HIP(hipGetLastError());
}

// The inner sums of a baby-step / giant-step linear transform in ONE launch: out[g] = sum_b ct[b] * pt[g][b] for every giant
// step g.  Each baby-step ciphertext is read once for all giant steps (k_mac_plain, one launch per giant step, re-reads all of
// them every time): nb*polys + terms + ng*polys limb streams instead of terms*(polys + 1) + ng*polys.
struct MacPlainMultiArgs {
    const u64* ct[LSA_MACM_MAX];
    long long sct[LSA_MACM_MAX];
    const u64* pt[LSA_MACM_MAX][LSA_MACM_MAX];   // [giant][baby], null = no such diagonal; shared by the batch
    u64* out[LSA_MACM_MAX];
    long long so;
    const ModDev* mods;
    int nb, ng, polys, limbs, logn, batch, xcd_map;
    unsigned char mod_of[LSA_MAX_PERIOD];
};

// One workgroup = one 512-coefficient piece of one limb of one batch item, BOTH polynomials (they meet the same plaintext
// words).  The plaintext pieces are the larger half of a workgroup's reads (nb*ng of them against 2*nb ciphertext pieces) and are
// shared by the whole batch: the batch index varies fastest and, where the piece count allows, the workgroups of one piece
// are dealt to ONE XCD (consecutive workgroup ids go round-robin over the 8 XCDs, each with its own L2), so a piece is
// fetched from HBM once instead of once per batch item (LSA_MACM_NO_XCD=1: batch-fastest order only).
template <int POLYS>
__global__ __launch_bounds__(TPB) void k_mac_plain_multi(MacPlainMultiArgs g) {
    const int chunks = (1 << g.logn) / (2 * TPB);
    int piece, b;
    if (g.xcd_map) {
        const int w = (int)blockIdx.x, slot = w >> 3;
        b = slot % g.batch;
        piece = (slot / g.batch) * 8 + (w & 7);
    } else {
        b = (int)blockIdx.x % g.batch;
        piece = (int)blockIdx.x / g.batch;
    }
    const int limb = piece / chunks;
    const int x = ((piece % chunks) * TPB + threadIdx.x) * 2;
    const ModDev m = g.mods[g.mod_of[limb]];
    const long long poff = ((long long)limb << g.logn) + x, pstep = (long long)g.limbs << g.logn;
    ulonglong2 c[POLYS][LSA_MACM_MAX];
#pragma unroll
    for (int i = 0; i < LSA_MACM_MAX; i++)
        if (i < g.nb) {
#pragma unroll
            for (int p = 0; p < POLYS; p++) c[p][i] = ld2(g.ct[i] + (long long)b * g.sct[i] + p * pstep + poff);
        }
    for (int gi = 0; gi < g.ng; gi++) {
        u64 h[POLYS][2], l[POLYS][2];   // at most 8 products of < q^2, q < 2^61: below q * 2^64
#pragma unroll
        for (int p = 0; p < POLYS; p++) h[p][0] = h[p][1] = l[p][0] = l[p][1] = 0;
#pragma unroll
        for (int i = 0; i < LSA_MACM_MAX; i++) {
            if (i < g.nb && g.pt[gi][i]) {
                const ulonglong2 w = ld2(g.pt[gi][i] + poff);
#pragma unroll
                for (int p = 0; p < POLYS; p++) {
                    mac128(h[p][0], l[p][0], c[p][i].x, w.x);
                    mac128(h[p][1], l[p][1], c[p][i].y, w.y);
                }
            }
        }
#pragma unroll
        for (int p = 0; p < POLYS; p++) {
            u64 r0 = csub(mont_redc_lazy(h[p][0], l[p][0], m.q, m.qinv), m.q), r1 = csub(mont_redc_lazy(h[p][1], l[p][1], m.q, m.qinv), m.q);
            r0 = mont_mul(r0, m.r2, m.q, m.qinv);            // sum * R^-1 -> sum
            r1 = mont_mul(r1, m.r2, m.q, m.qinv);
            st2(g.out[gi] + (long long)b * g.so + p * pstep + poff, r0, r1);
        }
    }
}

void launch_mac_plain_multi(Context& c, int nb, const u64* const* ct, const long long* sct, int ng, const u64* const* pt /*[ng*nb]*/,
                            u64* const* out, long long so, int batch, int polys, int limbs, const RowMap& rm, hipStream_t s) {
    if (batch <= 0 || nb <= 0 || ng <= 0) return;
    LSA_REQUIRE(nb <= LSA_MACM_MAX && ng <= LSA_MACM_MAX, "too many baby or giant steps for one multiply-accumulate launch");
    LSA_REQUIRE(rm.period == limbs && limbs <= LSA_MAX_PERIOD, "mac: row map must cover the limbs");
    MacPlainMultiArgs g{};
    int terms = 0;
    for (int i = 0; i < nb; i++) {
        g.ct[i] = ct[i];
        g.sct[i] = sct[i];
    }
    for (int gi = 0; gi < ng; gi++) {
        g.out[gi] = out[gi];
        for (int i = 0; i < nb; i++) {
            g.pt[gi][i] = pt[gi * nb + i];
            terms += pt[gi * nb + i] != nullptr;
        }
    }
    g.so = so;
    g.mods = c.d_mods;
    g.nb = nb;
    g.ng = ng;
    g.polys = polys;
    g.limbs = limbs;
    g.logn = c.logn;
    int period;
    fill_rowmap(g.mod_of, period, rm, c.nmod);
    // algorithmic bytes: every ciphertext piece read once, every plaintext ONCE per launch (shared by the batch), the sums written
    ProfScope ps(c, PROF_ELEMWISE, 8.0 * c.n * limbs * ((double)batch * polys * (nb + ng) + (double)terms), s);
    LSA_REQUIRE(polys == 1 || polys == 2, "mac: one or two polynomials per ciphertext");
    const int pieces = limbs * (c.n / (2 * TPB));
    g.batch = batch;
    g.xcd_map = (pieces % 8 == 0 && !std::getenv("LSA_MACM_NO_XCD")) ? 1 : 0;
    LSA_REQUIRE(c.n >= 2 * TPB, "ring degree too small for the elementwise kernels (need N >= 512)");
    const dim3 grid((unsigned)(pieces * batch));
    if (polys == 2) hipLaunchKernelGGL(k_mac_plain_multi<2>, grid, dim3(TPB), 0, s, g);
    else hipLaunchKernelGGL(k_mac_plain_multi<1>, grid, dim3(TPB), 0, s, g);
    LSA_HIP(hipGetLastError());
}

// tensor product of two degree-1 ciphertexts: d0=a0*b0, d1=a0*b1+a1*b0, d2=a1*b1 (mega_ag_executors_gpu.cu:185,223)
struct TensorArgs {
    const u64* a;
    const u64* b;
    u64* d;
    long long sa, sb, sd;
    long long pa, pb;   // elements between the two polynomials of a / b (operands kept at a higher level: more rows per polynomial)
    const ModDev* mods;
    int limbs, logn;
    unsigned char mod_of[LSA_MAX_PERIOD];
};

__global__ __launch_bounds__(TPB) void k_tensor(TensorArgs g) {
    const int chunks = (1 << g.logn) / (2 * TPB);
    const int limb = blockIdx.x / chunks;
    const int x = ((blockIdx.x % chunks) * TPB + threadIdx.x) * 2;
    const ModDev m = g.mods[g.mod_of[limb]];
    const long long b = blockIdx.y;
    const long long poly = (long long)g.limbs << g.logn;
    const long long off = ((long long)limb << g.logn) + x;
    const u64* pa = g.a + b * g.sa + off;
    const u64* pb = g.b + b * g.sb + off;
    u64* pd = g.d + b * g.sd + off;
    const ulonglong2 a0 = ld2(pa), a1 = ld2(pa + g.pa), b0 = ld2(pb), b1 = ld2(pb + g.pb);
    u64 r[3][2];
    const u64 a0v[2] = {a0.x, a0.y}, a1v[2] = {a1.x, a1.y}, b0v[2] = {b0.x, b0.y}, b1v[2] = {b1.x, b1.y};
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const u64 b0m = mont_mul_lazy(b0v[e], m.r2, m.q, m.qinv);  // to Montgomery form, [0,2q)
        const u64 b1m = mont_mul_lazy(b1v[e], m.r2, m.q, m.qinv);
        r[0][e] = mont_mul(a0v[e], b0m, m.q, m.qinv);
        r[2][e] = mont_mul(a1v[e], b1m, m.q, m.qinv);
        r[1][e] = add_mod(mont_mul(a0v[e], b1m, m.q, m.qinv), mont_mul(a1v[e], b0m, m.q, m.qinv), m.q);
    }
    st2(pd, r[0][0], r[0][1]);
    st2(pd + poly, r[1][0], r[1][1]);
    st2(pd + 2 * poly, r[2][0], r[2][1]);
}

void launch_tensor(Context& c, const u64* a, const u64* b, u64* d, int batch, long long sa, long long sb, long long sd,
                   int limbs, const RowMap& rm, hipStream_t s, int a_rpp, int b_rpp) {
    if (batch <= 0) return;
    TensorArgs g{};
    LSA_REQUIRE((a_rpp == 0 || a_rpp >= limbs) && (b_rpp == 0 || b_rpp >= limbs), "tensor: rows per polynomial below the limb count");
    g.pa = (long long)(a_rpp ? a_rpp : limbs) << c.logn;
    g.pb = (long long)(b_rpp ? b_rpp : limbs) << c.logn;
    g.a = a;
    g.b = b;
    g.d = d;
    g.sa = sa;
    g.sb = sb;
    g.sd = sd;
    g.mods = c.d_mods;
    g.limbs = limbs;
    g.logn = c.logn;
    LSA_REQUIRE(rm.period == limbs && limbs <= LSA_MAX_PERIOD, "tensor: row map must cover the limbs");
    int period;
    fill_rowmap(g.mod_of, period, rm, c.nmod);
    ProfScope ps(c, PROF_TENSOR, 7.0 * 8 * c.n * limbs * batch, s);
    hipLaunchKernelGGL(k_tensor, ew_grid(c, limbs, batch), dim3(TPB), 0, s, g);
    LSA_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ exact base conversion
//  y_i = x_i*(S/q_i)^-1 mod q_i ; v = floor(sum double(y_i)/double(q_i)) ;
// out_j = sum_i y_i*(S/q_i) - v*S mod p_j   [centered: x+floor(S/2) in, -floor(S/2) out]
struct BaseConvArgs {
    const BaseConvConsts* k;
    const ModDev* mods;
    const u64* src;
    u64* dst;
    long long ssrc, sdst;
    int logn;
    BaseConvRows rows;
    const u64* sub;   // optional: the converted value is src - sub (limb i of sub at row sub_row[i], batch stride ssub)
    long long ssub;
    int sub_row[LSA_BC_MAX_SRC];
};

// NSMAX = compile-time bound of the source-limb loops (registers for y[] scale with it; dispatched from ns); EXACT: ns ==
// NSMAX, so the source loads carry no guard and are all in flight before the first is consumed.  The target loop has a
// fixed trip count (the tail group repeats its last target: same value stored twice by the same thread), so the per-target
// constants are fetched up front instead of one scalar-load latency chain per target.
// SPLIT (every modulus of the conversion below 2^58, BaseConvConsts::split29): the products y_i * shat_ij are accumulated on
// 29-bit halves -- y = y1 * 2^29 + y0, shat = w1 * 2^29 + w0, three 64-bit column sums y0 w0, y0 w1 + y1 w0, y1 w1 that
// cannot overflow for up to 16 terms -- so every partial product is ONE v_mad_u64_u32 into its own accumulator pair: no
// carries, no 64-bit addend built from register moves (the 128-bit form spends 14 instructions per term, 4 of them moves).
// The columns are recombined into the 128-bit sum once per target; same sum, same REDC, same residues.
// The conversion plan and the modulus table are read through the CONSTANT address space: they are never written while a
// kernel runs, and only then may the compiler keep their (wave-uniform) loads on the scalar unit after the kernel's first
// store -- as plain global loads 86 of a thread's 103 vector-memory instructions were broadcast reads of these constants,
// queued in front of its 13 stores.
#define LSA_CONST_AS __attribute__((address_space(4)))
__device__ __forceinline__ ModDev ld_mod(const LSA_CONST_AS ModDev* t, int i) {
    ModDev m;
    m.q = t[i].q;
    m.qinv = t[i].qinv;
    m.r2 = t[i].r2;
    m.r1 = t[i].r1;
    return m;
}
template <int NSMAX, bool EXACT, int TGT, bool SPLIT = false>
__global__ __launch_bounds__(TPB) void k_baseconv(BaseConvArgs g) {
    const LSA_CONST_AS BaseConvConsts& K = *(const LSA_CONST_AS BaseConvConsts*)g.k;
    const LSA_CONST_AS ModDev* const mods_c = (const LSA_CONST_AS ModDev*)g.mods;
    const int x = (blockIdx.x * TPB + threadIdx.x) * 2;
    const long long b = blockIdx.y;
    const u64* src = g.src + b * g.ssrc + x;
    u64* dst = g.dst + b * g.sdst + x;
    const int ns = EXACT ? NSMAX : K.ns, nd = K.nd;
    ulonglong2 xin[NSMAX];
#pragma unroll
    for (int i = 0; i < NSMAX; i++)
        if (EXACT || i < ns) xin[i] = ld2(src + ((long long)g.rows.src_row[i] << g.logn));
    if (g.sub) {   // block-uniform: a branch around the loop, only the launches that fold a subtraction take it
        const u64* sub = g.sub + b * g.ssub + x;
#pragma unroll
        for (int i = 0; i < NSMAX; i++)
            if (EXACT || i < ns) {
                const u64 q = mods_c[K.src_mod[i]].q;
                const ulonglong2 w = ld2(sub + ((long long)g.sub_row[i] << g.logn));
                xin[i].x = sub_mod(xin[i].x, w.x, q);
                xin[i].y = sub_mod(xin[i].y, w.y, q);
            }
    }
    u64 y[NSMAX][2];
    double vf0 = 0.0, vf1 = 0.0;
#pragma unroll
    for (int i = 0; i < NSMAX; i++) {
        if (EXACT || i < ns) {
            const ModDev m = ld_mod(mods_c, K.src_mod[i]);
            ulonglong2 v = xin[i];
            if (K.centered) {
                v.x = add_mod(v.x, K.half_src[i], m.q);
                v.y = add_mod(v.y, K.half_src[i], m.q);
            }
            y[i][0] = mont_mul(v.x, K.shat_inv_m[i], m.q, m.qinv);
            y[i][1] = mont_mul(v.y, K.shat_inv_m[i], m.q, m.qinv);
            // y/q correctly rounded == the oracle's IEEE division, in 3 operations (Markstein: with r = RN(1/q) and
            // q0 = RN(y*r), RN(q0 + (y - q0*q)*r) is the correctly rounded quotient; q's significand is not all ones),
            // then sequential adds: same float sequence as the oracle
            const double qf = K.qf[i], rf = K.rf[i];
            const double a0 = (double)y[i][0], a1 = (double)y[i][1];
            const double e0 = a0 * rf, e1 = a1 * rf;
            vf0 += __builtin_fma(__builtin_fma(-e0, qf, a0), rf, e0);
            vf1 += __builtin_fma(__builtin_fma(-e1, qf, a1), rf, e1);
        }
    }
    const int v0 = (int)(u64)vf0, v1 = (int)(u64)vf1;
    // targets are split over blockIdx.z (each block recomputes y_i/v and converts TGT targets)
    const int j0 = blockIdx.z * TGT;
    u32 ylo[SPLIT ? NSMAX : 1][2], yhi[SPLIT ? NSMAX : 1][2], ysum[SPLIT ? NSMAX : 1][2];
    if (SPLIT) {
#pragma unroll
        for (int i = 0; i < NSMAX; i++)
#pragma unroll
            for (int e = 0; e < 2; e++) {
                ylo[i][e] = (u32)y[i][e] & ((1u << 29) - 1);
                yhi[i][e] = (u32)(y[i][e] >> 29);
                ysum[i][e] = ylo[i][e] + yhi[i][e];
            }
    }
#pragma unroll
    for (int jj = 0; jj < TGT; jj++) {
        const int j = min(j0 + jj, nd - 1);
        const ModDev m = ld_mod(mods_c, K.dst_mod[j]);
        u64 h0 = 0, l0 = 0, h1 = 0, l1 = 0, r0 = 0, r1 = 0;
#if defined(LSA_BC_DIAG_NO_MATH)       // diagnostic build: the loads and the stores, no conversion arithmetic
        if (SPLIT) {
            st2(dst + ((long long)g.rows.dst_row[j] << g.logn), xin[jj % NSMAX].x + v0, xin[jj % NSMAX].y + v1);
            continue;
        }
#endif
        if (SPLIT) {
            // columns y0 w0, (y0 + y1)(w0 + w1), y1 w1: three multiply-accumulates per term (the middle column is recovered as
            // cm - c0 - c2); 30-bit factors: up to 8 terms stay below 2^63, more take the four-product form
            constexpr bool KARA = NSMAX <= 8;
            u64 c0[2] = {0, 0}, c1[2] = {0, 0}, c2[2] = {0, 0};
#pragma unroll
            for (int i = 0; i < NSMAX; i++) {
                if (EXACT || i < ns) {
                    const u32 w0 = K.shat_lo[j][i], w1 = K.shat_hi[j][i], ws = K.shat_sum[j][i];
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        c0[e] += (u64)ylo[i][e] * w0;
                        c2[e] += (u64)yhi[i][e] * w1;
                        if (KARA) {
                            c1[e] += (u64)ysum[i][e] * ws;
                        } else {
                            c1[e] += (u64)ylo[i][e] * w1;
                            c1[e] += (u64)yhi[i][e] * w0;
                        }
                    }
                }
            }
            // sum = c0 + c1 * 2^29 + c2 * 2^58 (+ the output corrections, Montgomery form) as (hi, lo) below p_j * 2^64
            u64 hs[2], ls[2];
            const u64 cr[2] = {(u64)(u32)v0 * K.corr_a[j] + K.corr_b[j], (u64)(u32)v1 * K.corr_a[j] + K.corr_b[j]};
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const u64 mid = KARA ? c1[e] - c0[e] - c2[e] : c1[e];
#if defined(LSA_BC_CARRY_COMPARES)   // A/B: carries recovered by 64-bit compares (18 cycles each on gfx950, tools/probe_issue.hip)
                const u64 a = c0[e] + (mid << 29);
                const u64 b = a + (c2[e] << 58);
                const u64 d = b + cr[e];
                ls[e] = d;
                hs[e] = (mid >> 35) + (c2[e] >> 6) + (a < c0[e] ? 1 : 0) + (b < a ? 1 : 0) + (d < b ? 1 : 0);
#else                                // one 128-bit sum: the carries ride the v_addc chain
                const unsigned __int128 sum = (unsigned __int128)c0[e] + ((unsigned __int128)mid << 29) + ((unsigned __int128)c2[e] << 58) + cr[e];
                ls[e] = (u64)sum;
                hs[e] = (u64)(sum >> 64);
#endif
            }
            r0 = csub(mont_redc_lazy(hs[0], ls[0], m.q, m.qinv), m.q);
            r1 = csub(mont_redc_lazy(hs[1], ls[1], m.q, m.qinv), m.q);
#if defined(LSA_BC_DIAG_NO_STORE)      // diagnostic build: all the arithmetic, (almost) no store traffic
            if (r0 == 0x123456789abcdefull) st2(dst + ((long long)g.rows.dst_row[j] << g.logn), r0, r1);
#else
            st2(dst + ((long long)g.rows.dst_row[j] << g.logn), r0, r1);
#endif
            continue;
        } else {
#pragma unroll
        for (int i = 0; i < NSMAX; i++) {
            if (EXACT || i < ns) {
                const u64 w = K.shat_m[j][i];
                mac128(h0, l0, y[i][0], w);
                mac128(h1, l1, y[i][1], w);
                if ((i & 7) == 7) {  // keep the 128-bit sum below p_j*2^64 (8 products of < 2^61 * p_j)
                    r0 = add_mod(r0, csub(mont_redc_lazy(h0, l0, m.q, m.qinv), m.q), m.q);
                    r1 = add_mod(r1, csub(mont_redc_lazy(h1, l1, m.q, m.qinv), m.q), m.q);
                    h0 = l0 = h1 = l1 = 0;
                }
            }
        }
        r0 = add_mod(r0, csub(mont_redc_lazy(h0, l0, m.q, m.qinv), m.q), m.q);
        r1 = add_mod(r1, csub(mont_redc_lazy(h1, l1, m.q, m.qinv), m.q), m.q);
        }
        r0 = sub_mod(r0, K.vs[j][v0], m.q);
        r1 = sub_mod(r1, K.vs[j][v1], m.q);
        if (K.centered) {
            r0 = sub_mod(r0, K.half_dst[j], m.q);
            r1 = sub_mod(r1, K.half_dst[j], m.q);
        }
        st2(dst + ((long long)g.rows.dst_row[j] << g.logn), r0, r1);
    }
}

template <int NSMAX, int TGT>
static void launch_baseconv_nt(int ns, int nd, bool split, dim3 grid, hipStream_t s, const BaseConvArgs& g) {
    grid.z = (unsigned)((nd + TGT - 1) / TGT);
    if (split) {
        if (ns == NSMAX) hipLaunchKernelGGL((k_baseconv<NSMAX, true, TGT, true>), grid, dim3(TPB), 0, s, g);
        else hipLaunchKernelGGL((k_baseconv<NSMAX, false, TGT, true>), grid, dim3(TPB), 0, s, g);
        return;
    }
    if (ns == NSMAX) hipLaunchKernelGGL((k_baseconv<NSMAX, true, TGT>), grid, dim3(TPB), 0, s, g);
    else hipLaunchKernelGGL((k_baseconv<NSMAX, false, TGT>), grid, dim3(TPB), 0, s, g);
}
// targets per block: the candidate with the least total work ceil(nd/T) * (Y + T*C), Y = y/v phase ~ 2.5 target conversions
template <int NSMAX>
static void launch_baseconv_ns(int ns, int nd, bool split, dim3 grid, hipStream_t s, const BaseConvArgs& g) {
    const int cand[3] = {4, 7, 13};
    int best = 4;
    double best_cost = 1e30;
    for (int T : cand) {
        const double cost = (double)((nd + T - 1) / T) * (2.5 + T);
        if (cost < best_cost) {
            best_cost = cost;
            best = T;
        }
    }
    if (best == 4) launch_baseconv_nt<NSMAX, 4>(ns, nd, split, grid, s, g);
    else if (best == 7) launch_baseconv_nt<NSMAX, 7>(ns, nd, split, grid, s, g);
    else launch_baseconv_nt<NSMAX, 13>(ns, nd, split, grid, s, g);
}

void launch_baseconv(Context& c, const BaseConvPlan* k, const BaseConvRows& rows, const u64* src, u64* dst, int batch,
                     long long ssrc, long long sdst, hipStream_t s, const u64* sub, long long ssub, const int* sub_row) {
    if (batch <= 0) return;
    BaseConvArgs g{};
    g.sub = sub;
    g.ssub = ssub;
    LSA_REQUIRE(!sub || sub_row, "base conversion: subtrahend rows missing");
    if (sub)
        for (int i = 0; i < k->ns; i++) g.sub_row[i] = sub_row[i];
    g.k = k->dev;
    g.mods = c.d_mods;
    g.src = src;
    g.dst = dst;
    g.ssrc = ssrc;
    g.sdst = sdst;
    g.logn = c.logn;
    g.rows = rows;
    ProfScope ps(c, PROF_BASECONV, 8.0 * c.n * batch * (double)(k->ns * (sub ? 2 : 1) + k->nd), s);
    const dim3 grid((unsigned)(c.n / (2 * TPB)), (unsigned)batch, 1);
    const int ns = k->ns, nd = k->nd;
    if (ns <= 1) launch_baseconv_ns<1>(ns, nd, k->split29, grid, s, g);
    else if (ns <= 2) launch_baseconv_ns<2>(ns, nd, k->split29, grid, s, g);
    else if (ns <= 3) launch_baseconv_ns<3>(ns, nd, k->split29, grid, s, g);
    else if (ns <= 4) launch_baseconv_ns<4>(ns, nd, k->split29, grid, s, g);
    else if (ns <= 5) launch_baseconv_ns<5>(ns, nd, k->split29, grid, s, g);
    else if (ns <= 8) launch_baseconv_ns<8>(ns, nd, k->split29, grid, s, g);
    else if (ns <= 12) launch_baseconv_ns<12>(ns, nd, k->split29, grid, s, g);
    else launch_baseconv_ns<LSA_BC_MAX_SRC>(ns, nd, k->split29, grid, s, g);
    LSA_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ key-switch MAC (K7)
struct KsMacArgs {
    const u64* cx;
    const u64* ext;
    const u64* key;
    u64* acc;
    long long scx, sext, sacc;
    const ModDev* mods;
    int logn, L, np, nq, beta, kcomp, klvl, batch, bpt;
    int n_tl;                      // 0: every target limb; else the launch covers tl_list[0..n_tl)
    unsigned char tl_list[64];
    // EXT (k_ks_mac<KB, true>): the product leaves as a ROTATED EXTENDED ciphertext instead of the plain accumulator --
    // acc[b][h][tl][scatter[x]] = sum(x) + (h == 0 and tl < L ? P * base[b][tl][x] : 0): the gadget product, the c0 * P term and
    // the automorphism of a baby-step rotation in one pass (otherwise k_permute_ext re-reads and re-writes 2(L+k) limbs)
    const unsigned* scatter;
    const u64* base;
    const u64* pm;                 // [L] P mod q_j, Montgomery form
    long long sbase;
};

// grid: x = T * (N/2/TPB), y = groups of `bpt` batch items.  The key is in Montgomery form, so sum_d ext_d*key_d needs
// ONE REDC per output.  A thread keeps its 2*beta key words in registers and walks `bpt` ciphertexts with them: the key
// (68 MiB at the headline shape) is then streamed once per group instead of once per ciphertext.
// KB = number of digits whose key words are register-resident (0: stream the key per ciphertext)
template <int KB, bool EXT = false>
__global__ __launch_bounds__(TPB) void k_ks_mac(KsMacArgs g) {
    const int chunks = (1 << g.logn) / (2 * TPB);
    const int tl = g.n_tl ? g.tl_list[blockIdx.x / chunks] : blockIdx.x / chunks;
    const int x = ((blockIdx.x % chunks) * TPB + threadIdx.x) * 2;
    const int T = g.L + g.np;
    const int mi = tl < g.L ? tl : g.nq + (tl - g.L);
    const int kj = tl < g.L ? tl : g.klvl + 1 + (tl - g.L);
    const ModDev m = g.mods[mi];
    const long long N = 1LL << g.logn;
    const int own_d = tl < g.L ? tl / g.np : -1;   // the digit that contains this limb reads cx directly
    ulonglong2 k0[KB > 0 ? KB : 1], k1[KB > 0 ? KB : 1];
    constexpr bool in_regs = KB > 0;
    if constexpr (in_regs) {
#pragma unroll
        for (int d = 0; d < KB; d++) {
            if (d < g.beta) {
                const u64* pk = g.key + ((long long)(d * 2) * g.kcomp + kj) * N + x;
                k0[d] = ld2(pk);
                k1[d] = ld2(pk + (long long)g.kcomp * N);
            }
        }
    }
    const int b_begin = blockIdx.y * g.bpt;
    const int b_end = min(g.batch, b_begin + g.bpt);
    // register-resident key: the NEXT ciphertext's digit values are fetched before the current one is multiplied, so a wave
    // always has one ciphertext's loads in flight behind its arithmetic and its stores (the kernel is traffic-bound: with the
    // loads issued and consumed in the same iteration it ran at 4.5 TB/s, no faster without its arithmetic)
    ulonglong2 en[KB > 0 ? KB : 1];
    auto fetch = [&](long long b) {
        if constexpr (in_regs) {
#pragma unroll
            for (int d = 0; d < KB; d++)
                if (d < g.beta)
                    en[d] = ld2(d == own_d ? g.cx + b * g.scx + tl * N + x : g.ext + b * g.sext + ((long long)d * T + tl) * N + x);
        }
    };
    if (b_begin < b_end) fetch(b_begin);
    for (long long b = b_begin; b < b_end; b++) {
        u64 h00 = 0, l00 = 0, h01 = 0, l01 = 0, h10 = 0, l10 = 0, h11 = 0, l11 = 0;
        u64 r00 = 0, r01 = 0, r10 = 0, r11 = 0;
        if constexpr (in_regs) {
            ulonglong2 ec[KB > 0 ? KB : 1];
#pragma unroll
            for (int d = 0; d < KB; d++) ec[d] = en[d];
            if (b + 1 < b_end) fetch(b + 1);
#pragma unroll
            for (int d = 0; d < KB; d++) {
                if (d < g.beta) {
                    const ulonglong2 e = ec[d];
#if defined(LSA_KS_DIAG_NO_MATH)
                    l00 += e.x ^ k0[d].x; l01 += e.y ^ k0[d].y; l10 += e.x ^ k1[d].x; l11 += e.y ^ k1[d].y;
#else
                    mac128(h00, l00, e.x, k0[d].x);
                    mac128(h01, l01, e.y, k0[d].y);
                    mac128(h10, l10, e.x, k1[d].x);
                    mac128(h11, l11, e.y, k1[d].y);
#endif
                }
            }
        } else {
            for (int d = 0; d < g.beta; d++) {
                const u64* pe = d == own_d ? g.cx + b * g.scx + tl * N + x : g.ext + b * g.sext + ((long long)d * T + tl) * N + x;
                const ulonglong2 e = ld2(pe);
                const u64* pk = g.key + ((long long)(d * 2) * g.kcomp + kj) * N + x;
                const ulonglong2 kk0 = ld2(pk), kk1 = ld2(pk + (long long)g.kcomp * N);
                mac128(h00, l00, e.x, kk0.x);
                mac128(h01, l01, e.y, kk0.y);
                mac128(h10, l10, e.x, kk1.x);
                mac128(h11, l11, e.y, kk1.y);
                if ((d & 7) == 7) {  // fold so the 128-bit sum stays below q*2^64 (8 products of < q^2, q < 2^61)
                    r00 = add_mod(r00, csub(mont_redc_lazy(h00, l00, m.q, m.qinv), m.q), m.q);
                    r01 = add_mod(r01, csub(mont_redc_lazy(h01, l01, m.q, m.qinv), m.q), m.q);
                    r10 = add_mod(r10, csub(mont_redc_lazy(h10, l10, m.q, m.qinv), m.q), m.q);
                    r11 = add_mod(r11, csub(mont_redc_lazy(h11, l11, m.q, m.qinv), m.q), m.q);
                    h00 = l00 = h01 = l01 = h10 = l10 = h11 = l11 = 0;
                }
            }
        }
#if defined(LSA_KS_DIAG_NO_MATH)
        r00 = l00; r01 = l01; r10 = l10; r11 = l11;
#else
        r00 = add_mod(r00, csub(mont_redc_lazy(h00, l00, m.q, m.qinv), m.q), m.q);
        r01 = add_mod(r01, csub(mont_redc_lazy(h01, l01, m.q, m.qinv), m.q), m.q);
        r10 = add_mod(r10, csub(mont_redc_lazy(h10, l10, m.q, m.qinv), m.q), m.q);
        r11 = add_mod(r11, csub(mont_redc_lazy(h11, l11, m.q, m.qinv), m.q), m.q);
#endif
        if constexpr (EXT) {
            if (tl < g.L) {
                const ulonglong2 c0 = ld2(g.base + b * g.sbase + tl * N + x);
                const u64 k = g.pm[tl];
                r00 = add_mod(r00, mont_mul(c0.x, k, m.q, m.qinv), m.q);
                r01 = add_mod(r01, mont_mul(c0.y, k, m.q, m.qinv), m.q);
            }
            const uint2 sx = *reinterpret_cast<const uint2*>(g.scatter + x);
            u64* po = g.acc + b * g.sacc + tl * N;
            po[sx.x] = r00;
            po[sx.y] = r01;
            po[(long long)T * N + sx.x] = r10;
            po[(long long)T * N + sx.y] = r11;
            continue;
        }
        u64* pa = g.acc + b * g.sacc + tl * N + x;
#if defined(LSA_KS_DIAG_NO_STORE)
        if (r00 == 0x123456789abcdefull) st2(pa, r00, r01);
        if (r10 == 0x123456789abcdefull) st2(pa + (long long)T * N, r10, r11);
#else
        st2(pa, r00, r01);
        st2(pa + (long long)T * N, r10, r11);
#endif
    }
}

template <bool EXT>
static void launch_ks_mac_kb(int beta, dim3 grid, hipStream_t s, const KsMacArgs& g) {
    // the register-resident key costs 8 VGPRs per digit slot: 5 and 6 digits (the 25Q+5P chains) get their own instantiations
    // instead of the 8-slot one (182 VGPRs, 2 waves per SIMD)
    if (beta <= 2) hipLaunchKernelGGL((k_ks_mac<2, EXT>), grid, dim3(TPB), 0, s, g);   // low levels: 86 VGPRs, 5 waves per SIMD
    else if (beta <= 4) hipLaunchKernelGGL((k_ks_mac<4, EXT>), grid, dim3(TPB), 0, s, g);
    else if (beta <= 5) hipLaunchKernelGGL((k_ks_mac<5, EXT>), grid, dim3(TPB), 0, s, g);
    else if (beta <= 6) hipLaunchKernelGGL((k_ks_mac<6, EXT>), grid, dim3(TPB), 0, s, g);
    else if (beta <= 8) hipLaunchKernelGGL((k_ks_mac<8, EXT>), grid, dim3(TPB), 0, s, g);
    else hipLaunchKernelGGL((k_ks_mac<0, EXT>), grid, dim3(TPB), 0, s, g);
}

// scatter (with engine < 0): the result is written as the rotated extended ciphertext perm(acc + P * c0) -- scatter = the index
// map of the rotation's inverse element, base = the ciphertext whose c0 enters (see KsMacArgs)
void launch_ks_mac(Context& c, int level, const u64* cx, long long scx, const u64* ext, long long sext, const Key& key,
                   u64* acc, long long sacc, int batch, hipStream_t s, int engine, const u32* scatter, const u64* base, long long sbase) {
    if (batch <= 0) return;
    KsMacArgs g{};
    LSA_REQUIRE(!scatter || (engine < 0 && base), "key MAC: the extended output covers every target limb and needs the ciphertext");
    if (scatter) {
        g.pm = c.pmodq_vec(level);
        g.scatter = scatter;
        g.base = base;
        g.sbase = sbase;
    }
    g.cx = cx;
    g.ext = ext;
    g.key = key.data;
    g.acc = acc;
    g.scx = scx;
    g.sext = sext;
    g.sacc = sacc;
    g.mods = c.d_mods;
    g.logn = c.logn;
    g.L = level + 1;
    g.np = c.np;
    g.nq = c.nq;
    g.beta = (g.L + c.np - 1) / c.np;
    g.klvl = key.level;
    g.kcomp = key.level + 1 + c.np;
    LSA_REQUIRE(key.level >= level, "key-switch key exported at a lower level than the ciphertext");
    int targets = g.L + c.np;
    if (engine >= 0) {   // only the target limbs of one butterfly engine (the others went through the fused kernel)
        LSA_REQUIRE(targets <= 64, "key MAC: too many target limbs for a subset launch");
        for (int tl = 0; tl < targets; tl++) {
            const int mi = tl < g.L ? tl : c.p_mod(tl - g.L);
            const bool fp = c.fp64_ntt && (c.T.mod[mi] >> LSA_FP64_MAX_BITS) == 0;
            if ((int)fp == engine) g.tl_list[g.n_tl++] = (unsigned char)tl;
        }
        if (!g.n_tl) return;
        targets = g.n_tl;
    }
    const double T = targets;
    ProfScope ps(c, PROF_KSMAC, 8.0 * c.n * (batch * (g.beta * T + 2 * T + (scatter ? g.L : 0)) + 2.0 * g.beta * T), s);
    // enough workgroups to fill the chip, as few key re-reads as possible
    const dim3 grid1 = ew_grid(c, targets, 1);
    const int groups = std::max(1, std::min(batch, (int)((2048 + grid1.x - 1) / grid1.x)));
    g.batch = batch;
    g.bpt = (batch + groups - 1) / groups;
    const dim3 grid(grid1.x, (unsigned)((batch + g.bpt - 1) / g.bpt));
    if (scatter) launch_ks_mac_kb<true>(g.beta, grid, s, g);
    else launch_ks_mac_kb<false>(g.beta, grid, s, g);
    LSA_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ (a-b)*k, a*k, ModDown tail
struct SubMulArgs {
    const u64* a;
    const u64* b;     // may be null: out = a*k
    const u64* base;  // may be null: else out += base
    const u64* kvec;  // per row-class constant, Montgomery form, indexed like mod_of
    u64* out;
    long long sa, sb, sbase, so;
    int a_rpp, b_rpp, base_rpp, out_rpp;  // rows per polynomial of each operand (row = poly*rpp + limb)
    int base_polys;                       // base is added to polynomials [0, base_polys)
    int limbs, logn;
    const ModDev* mods;
    unsigned char mod_of[LSA_MAX_PERIOD];  // per limb
};

// grid: x = polys*limbs*(N/2/TPB), y = batch
__global__ __launch_bounds__(TPB) void k_sub_mul(SubMulArgs g) {
    const int chunks = (1 << g.logn) / (2 * TPB);
    const int pl = blockIdx.x / chunks;
    const int poly = pl / g.limbs, limb = pl % g.limbs;
    const int x = ((blockIdx.x % chunks) * TPB + threadIdx.x) * 2;
    const ModDev m = g.mods[g.mod_of[limb]];
    const u64 k = g.kvec[limb];
    const long long b = blockIdx.y;
    ulonglong2 va = ld2(g.a + b * g.sa + (((long long)poly * g.a_rpp + limb) << g.logn) + x);
    if (g.b) {
        const ulonglong2 vb = ld2(g.b + b * g.sb + (((long long)poly * g.b_rpp + limb) << g.logn) + x);
        va.x = sub_mod(va.x, vb.x, m.q);
        va.y = sub_mod(va.y, vb.y, m.q);
    }
    u64 r0 = mont_mul(va.x, k, m.q, m.qinv), r1 = mont_mul(va.y, k, m.q, m.qinv);
    if (g.base && poly < g.base_polys) {
        const ulonglong2 vc = ld2(g.base + b * g.sbase + (((long long)poly * g.base_rpp + limb) << g.logn) + x);
        r0 = add_mod(r0, vc.x, m.q);
        r1 = add_mod(r1, vc.y, m.q);
    }
    st2(g.out + b * g.so + (((long long)poly * g.out_rpp + limb) << g.logn) + x, r0, r1);
}

void launch_sub_mul_general(Context& c, int polys, int limbs, const unsigned char* limb_mod, const u64* kvec,
                                   const u64* a, long long sa, int a_rpp, const u64* b, long long sb, int b_rpp,
                                   const u64* base, long long sbase, int base_rpp, int base_polys, u64* out,
                                   long long so, int out_rpp, int batch, hipStream_t s) {
    if (batch <= 0) return;
    SubMulArgs g{};
    g.a = a;
    g.b = b;
    g.base = base;
    g.kvec = kvec;
    g.out = out;
    g.sa = sa;
    g.sb = sb;
    g.sbase = sbase;
    g.so = so;
    g.a_rpp = a_rpp;
    g.b_rpp = b_rpp;
    g.base_rpp = base_rpp;
    g.base_polys = base_polys;
    g.out_rpp = out_rpp;
    g.limbs = limbs;
    g.logn = c.logn;
    g.mods = c.d_mods;
    LSA_REQUIRE(limbs <= LSA_MAX_PERIOD, "too many limbs");
    for (int i = 0; i < limbs; i++) g.mod_of[i] = limb_mod[i];
    ProfScope ps(c, PROF_ELEMWISE, 8.0 * c.n * polys * limbs * batch * (2 + (b != nullptr) + (base != nullptr)), s);
    hipLaunchKernelGGL(k_sub_mul, ew_grid(c, polys * limbs, batch), dim3(TPB), 0, s, g);
    LSA_HIP(hipGetLastError());
}

void launch_moddown_final(Context& c, int level, const u64* acc, long long sacc, int acc_rpp, const u64* conv,
                          long long sconv, const u64* base, long long sbase, int base_rpp, int base_polys, u64* out,
                          long long sout, int batch, hipStream_t s) {
    const int L = level + 1;
    std::vector<int> mods(L);
    std::vector<u64> pinv(L);
    unsigned char lm[LSA_MAX_PERIOD];
    for (int i = 0; i < L; i++) {
        mods[i] = i;
        lm[i] = (unsigned char)i;
        u64 q = c.T.mod[i], pr = 1;
        for (int l = 0; l < c.np; l++) pr = mul_mod_host(pr, c.T.mod[c.p_mod(l)] % q, q);
        pinv[i] = inv_mod(pr, q);
    }
    const u64* kv = c.const_vec("pinv" + std::to_string(L), mods, pinv);
    launch_sub_mul_general(c, 2, L, lm, kv, acc, sacc, acc_rpp, conv, sconv, L, base, sbase, base_rpp, base_polys, out,
                           sout, L, batch, s);
}

void launch_sub_mul_const(Context& c, const u64* a, long long sa, const u64* b, long long sb, const u64* kvec, u64* out,
                          long long so, int rows, const RowMap& rm, int batch, hipStream_t s) {
    launch_sub_mul_general(c, 1, rows, rm.mod_of, kvec, a, sa, rows, b, sb, rows, nullptr, 0, 0, 0, out, so, rows, batch,
                           s);
}

void launch_mul_const(Context& c, const u64* a, long long sa, const u64* kvec, u64* out, long long so, int rows,
                      const RowMap& rm, int batch, hipStream_t s) {
    launch_sub_mul_general(c, 1, rows, rm.mod_of, kvec, a, sa, rows, nullptr, 0, 0, nullptr, 0, 0, 0, out, so, rows,
                           batch, s);
}

// ------------------------------------------------------------------------------------------------ rescale (K9)
struct RescaleArgs {
    const u64* last;  // [batch][polys][N] coefficient-domain last limb
    u64* tmp;         // [batch][polys][level][N]
    long long slast, stmp;
    const ModDev* mods;
    int level, polys, logn;
};

// tmp[p][i] = ((last + h) mod q_l) mod q_i - (h mod q_i),  h = (q_l-1)/2     (divide-and-round, centred remainder)
__global__ __launch_bounds__(TPB) void k_rescale_prep(RescaleArgs g) {
    const int chunks = (1 << g.logn) / (2 * TPB);
    const int pl = blockIdx.x / chunks;
    const int poly = pl / g.level, limb = pl % g.level;
    const int x = ((blockIdx.x % chunks) * TPB + threadIdx.x) * 2;
    const long long b = blockIdx.y;
    const ModDev ml = g.mods[g.level], mi = g.mods[limb];
    const u64 h = (ml.q - 1) >> 1;
    const u64 hq = reduce_u64(h, mi);
    const ulonglong2 v = ld2(g.last + b * g.slast + ((long long)poly << g.logn) + x);
    const u64 r0 = sub_mod(reduce_u64(add_mod(v.x, h, ml.q), mi), hq, mi.q);
    const u64 r1 = sub_mod(reduce_u64(add_mod(v.y, h, ml.q), mi), hq, mi.q);
    st2(g.tmp + b * g.stmp + (((long long)poly * g.level + limb) << g.logn) + x, r0, r1);
}

void launch_rescale_prep(Context& c, int level, int polys, const u64* last, long long slast, u64* tmp, long long stmp,
                         int batch, hipStream_t s) {
    if (batch <= 0) return;
    RescaleArgs g{};
    g.last = last;
    g.tmp = tmp;
    g.slast = slast;
    g.stmp = stmp;
    g.mods = c.d_mods;
    g.level = level;
    g.polys = polys;
    g.logn = c.logn;
    ProfScope ps(c, PROF_ELEMWISE, 8.0 * c.n * polys * (level + 1) * batch, s);
    hipLaunchKernelGGL(k_rescale_prep, ew_grid(c, polys * level, batch), dim3(TPB), 0, s, g);
    LSA_HIP(hipGetLastError());
}

void launch_rescale_final(Context& c, int level, int polys, const u64* in, long long sin, const u64* tmp, long long stmp,
                          u64* out, long long sout, int batch, hipStream_t s) {
    std::vector<int> mods(level);
    std::vector<u64> qlinv(level);
    unsigned char lm[LSA_MAX_PERIOD];
    const u64 ql = c.T.mod[level];
    for (int i = 0; i < level; i++) {
        mods[i] = i;
        lm[i] = (unsigned char)i;
        qlinv[i] = inv_mod(ql % c.T.mod[i], c.T.mod[i]);
    }
    const u64* kv = c.const_vec("qlinv" + std::to_string(level), mods, qlinv);
    launch_sub_mul_general(c, polys, level, lm, kv, in, sin, level + 1, tmp, stmp, level, nullptr, 0, 0, 0, out, sout,
                           level, batch, s);
}

// ------------------------------------------------------------------------------------------------ automorphisms (K10)
struct PermArgs {
    const u32* perm;
    const u64* in;
    u64* out;
    long long sin, sout;
    const ModDev* mods;
    int rows, logn, with_sign, period;
    unsigned char mod_of[LSA_MAX_PERIOD];
};

// out[row][i] = (+/-) in[row][perm[i] & 0x7fffffff]; sign bit 31 negates (coefficient-domain automorphism)
__global__ __launch_bounds__(TPB) void k_permute(PermArgs g) {
    const int chunks = (1 << g.logn) / (2 * TPB);
    const int row = blockIdx.x / chunks;
    const int x = ((blockIdx.x % chunks) * TPB + threadIdx.x) * 2;
    const long long b = blockIdx.y;
    const u64* in = g.in + b * g.sin + ((long long)row << g.logn);
    const uint2 p = *reinterpret_cast<const uint2*>(g.perm + x);
    u64 r0 = in[p.x & 0x7fffffffu], r1 = in[p.y & 0x7fffffffu];
    if (g.with_sign) {
        const u64 q = g.mods[g.mod_of[row % g.period]].q;
        if (p.x >> 31) r0 = neg_mod(r0, q);
        if (p.y >> 31) r1 = neg_mod(r1, q);
    }
    st2(g.out + b * g.sout + ((long long)row << g.logn) + x, r0, r1);
}

void launch_permute_ntt(Context& c, const u32* perm, const u64* in, long long sin, u64* out, long long sout, int rows,
                        int batch, hipStream_t s) {
    if (batch <= 0) return;
    PermArgs g{};
    g.perm = perm;
    g.in = in;
    g.out = out;
    g.sin = sin;
    g.sout = sout;
    g.mods = c.d_mods;
    g.rows = rows;
    g.logn = c.logn;
    g.with_sign = 0;
    g.period = 1;
    ProfScope ps(c, PROF_ELEMWISE, 16.0 * c.n * rows * batch, s);
    hipLaunchKernelGGL(k_permute, ew_grid(c, rows, batch), dim3(TPB), 0, s, g);
    LSA_HIP(hipGetLastError());
}

// Extended (Q_level u P) ciphertexts, the operands of double-hoisted linear transforms: a rotation WITHOUT its division by P.
//   out[h][tl][i] (+)= acc[h][tl][perm[i]] + (tl < L and h < base_polys ? P * base[h][tl][perm[i]] mod q_tl : 0)
// acc = the gadget product [2][T][N] (null: zero), base = the ciphertext [2][L][N] whose c0 (base_polys = 1) or both
// polynomials (2: the plain lift of a ciphertext) enter times P; perm null = identity.
struct PermExtArgs {
    const u32* perm;
    const u64* acc;
    const u64* base;
    const u64* pm;   // [L] P mod q_j, Montgomery form
    u64* out;
    long long sacc, sbase, sout;
    const ModDev* mods;
    int L, T, logn, accumulate, base_polys;
    unsigned char mod_of[LSA_MAX_PERIOD];   // [T]
};

__global__ __launch_bounds__(TPB) void k_permute_ext(PermExtArgs g) {
    const int chunks = (1 << g.logn) / (2 * TPB);
    const int row = blockIdx.x / chunks;
    const int h = row / g.T, tl = row - h * g.T;
    const int x = ((blockIdx.x % chunks) * TPB + threadIdx.x) * 2;
    const long long b = blockIdx.y;
    const ModDev m = g.mods[g.mod_of[tl]];
    unsigned i0 = (unsigned)x, i1 = (unsigned)x + 1;
    if (g.perm) {
        const uint2 p = *reinterpret_cast<const uint2*>(g.perm + x);
        i0 = p.x;
        i1 = p.y;
    }
    u64 r0 = 0, r1 = 0;
    if (g.acc) {
        const u64* in = g.acc + b * g.sacc + ((long long)row << g.logn);
        r0 = in[i0];
        r1 = in[i1];
    }
    if (tl < g.L && h < g.base_polys) {
        const u64* bs = g.base + b * g.sbase + ((long long)(h * g.L + tl) << g.logn);
        const u64 k = g.pm[tl];
        r0 = add_mod(r0, mont_mul(bs[i0], k, m.q, m.qinv), m.q);
        r1 = add_mod(r1, mont_mul(bs[i1], k, m.q, m.qinv), m.q);
    }
    u64* o = g.out + b * g.sout + ((long long)row << g.logn) + x;
    if (g.accumulate) {
        const ulonglong2 v = ld2(o);
        r0 = add_mod(r0, v.x, m.q);
        r1 = add_mod(r1, v.y, m.q);
    }
    st2(o, r0, r1);
}

void launch_permute_ext(Context& c, int level, const u32* perm, const u64* acc, long long sacc, const u64* base, long long sbase,
                        int base_polys, u64* out, long long sout, bool accumulate, int batch, hipStream_t s) {
    if (batch <= 0) return;
    const int L = level + 1, T = L + c.np;
    LSA_REQUIRE(T <= LSA_MAX_PERIOD, "extended ciphertext: too many limbs");
    LSA_REQUIRE(base_polys == 0 || base, "extended ciphertext: base polynomial missing");
    PermExtArgs g{};
    g.perm = perm;
    g.acc = acc;
    g.base = base;
    g.out = out;
    g.sacc = sacc;
    g.sbase = sbase;
    g.sout = sout;
    g.mods = c.d_mods;
    g.L = L;
    g.T = T;
    g.logn = c.logn;
    g.accumulate = accumulate ? 1 : 0;
    g.base_polys = base_polys;
    g.pm = c.pmodq_vec(level);
    for (int tl = 0; tl < T; tl++) g.mod_of[tl] = (unsigned char)(tl < L ? tl : c.p_mod(tl - L));
    const double streams = (acc ? 1.0 : 0.0) * 2 * T + (double)base_polys * L + (accumulate ? 2.0 : 1.0) * 2 * T;
    ProfScope ps(c, PROF_ELEMWISE, 8.0 * c.n * streams * batch, s);
    hipLaunchKernelGGL(k_permute_ext, ew_grid(c, 2 * T, batch), dim3(TPB), 0, s, g);
    LSA_HIP(hipGetLastError());
}

void launch_permute_coeff(Context& c, const u32* perm, const u64* in, long long sin, u64* out, long long sout, int rows,
                          const RowMap& rm, int batch, hipStream_t s) {
    if (batch <= 0) return;
    PermArgs g{};
    g.perm = perm;
    g.in = in;
    g.out = out;
    g.sin = sin;
    g.sout = sout;
    g.mods = c.d_mods;
    g.rows = rows;
    g.logn = c.logn;
    g.with_sign = 1;
    fill_rowmap(g.mod_of, g.period, rm, c.nmod);
    ProfScope ps(c, PROF_ELEMWISE, 16.0 * c.n * rows * batch, s);
    hipLaunchKernelGGL(k_permute, ew_grid(c, rows, batch), dim3(TPB), 0, s, g);
    LSA_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ row copy (K12) / to-Montgomery
struct CopyRowsArgs {
    const u64* in;
    u64* out;
    long long sin, sout;
    int rows, logn;
    short src_row[LSA_MAX_PERIOD];
};

__global__ __launch_bounds__(TPB) void k_copy_rows(CopyRowsArgs g) {
    const int chunks = (1 << g.logn) / (2 * TPB);
    const int row = blockIdx.x / chunks;
    const int x = ((blockIdx.x % chunks) * TPB + threadIdx.x) * 2;
    const long long b = blockIdx.y;
    const ulonglong2 v = ld2(g.in + b * g.sin + ((long long)g.src_row[row] << g.logn) + x);
    st2(g.out + b * g.sout + ((long long)row << g.logn) + x, v.x, v.y);
}

void launch_copy_rows(Context& c, const u64* in, long long sin, u64* out, long long sout, int rows, const int* src_row,
                      int batch, hipStream_t s) {
    if (batch <= 0 || rows <= 0) return;
    LSA_REQUIRE(rows <= LSA_MAX_PERIOD, "copy_rows: too many rows");
    CopyRowsArgs g{};
    g.in = in;
    g.out = out;
    g.sin = sin;
    g.sout = sout;
    g.rows = rows;
    g.logn = c.logn;
    for (int i = 0; i < rows; i++) g.src_row[i] = (short)src_row[i];
    ProfScope ps(c, PROF_ELEMWISE, 16.0 * c.n * rows * batch, s);
    hipLaunchKernelGGL(k_copy_rows, ew_grid(c, rows, batch), dim3(TPB), 0, s, g);
    LSA_HIP(hipGetLastError());
}

struct ToMontArgs {
    u64* data;
    const ModDev* mods;
    int logn, period;
    unsigned char mod_of[LSA_MAX_PERIOD];
};

__global__ __launch_bounds__(TPB) void k_to_mont(ToMontArgs g) {
    const int chunks = (1 << g.logn) / (2 * TPB);
    const int row = blockIdx.x / chunks;
    const int x = ((blockIdx.x % chunks) * TPB + threadIdx.x) * 2;
    const ModDev m = g.mods[g.mod_of[row % g.period]];
    u64* p = g.data + ((long long)row << g.logn) + x;
    const ulonglong2 v = ld2(p);
    st2(p, mont_mul(v.x, m.r2, m.q, m.qinv), mont_mul(v.y, m.r2, m.q, m.qinv));
}

void launch_to_mont(Context& c, u64* data, int rows, const RowMap& rm, hipStream_t s) {
    if (rows <= 0) return;
    ToMontArgs g{};
    g.data = data;
    g.mods = c.d_mods;
    g.logn = c.logn;
    fill_rowmap(g.mod_of, g.period, rm, c.nmod);
    hipLaunchKernelGGL(k_to_mont, ew_grid(c, rows, 1), dim3(TPB), 0, s, g);
    LSA_HIP(hipGetLastError());
}

// a freshly loaded key: plain residues -> Montgomery form in place, and (fp != null) the same values as doubles for the limbs
// the FP64 engine serves
struct KeyPrepArgs {
    u64* data;
    double* fp;
    const ModDev* mods;
    int logn, period, allow_fp64;
    unsigned char mod_of[LSA_MAX_PERIOD];
};
__global__ __launch_bounds__(TPB) void k_key_prepare(KeyPrepArgs g) {
    const int chunks = (1 << g.logn) / (2 * TPB);
    const int row = blockIdx.x / chunks;
    const int x = ((blockIdx.x % chunks) * TPB + threadIdx.x) * 2;
    const ModDev m = g.mods[g.mod_of[row % g.period]];
    const long long off = ((long long)row << g.logn) + x;
    const ulonglong2 v = ld2(g.data + off);
    st2(g.data + off, mont_mul(v.x, m.r2, m.q, m.qinv), mont_mul(v.y, m.r2, m.q, m.qinv));
    if (g.fp) {
        const bool fp = g.allow_fp64 && (m.q >> LSA_FP64_MAX_BITS) == 0;
        double2 d;
        d.x = fp ? (double)v.x : 0.0;
        d.y = fp ? (double)v.y : 0.0;
        *reinterpret_cast<double2*>(g.fp + off) = d;
    }
}
void launch_key_prepare(Context& c, u64* data, double* fp, int key_level, hipStream_t s) {
    const int comp = key_level + 1 + c.np, beta = (key_level + 1 + c.np - 1) / c.np;
    LSA_REQUIRE(comp <= LSA_MAX_PERIOD, "key has too many limbs");
    KeyPrepArgs g{};
    g.data = data;
    g.fp = fp;
    g.mods = c.d_mods;
    g.logn = c.logn;
    g.period = comp;
    g.allow_fp64 = c.fp64_ntt;
    for (int j = 0; j < comp; j++) g.mod_of[j] = (unsigned char)(j <= key_level ? j : c.p_mod(j - key_level - 1));
    hipLaunchKernelGGL(k_key_prepare, ew_grid(c, beta * 2 * comp, 1), dim3(TPB), 0, s, g);
    LSA_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ ring-t plaintext lifts (K11)
// A ring-t plaintext is one coefficient-domain limb (plug-in/lattigo/acc/c_struct_import_export.go:179-184).
//   mode 0 (CKKS): residues mod q_src as CENTRED integers -> every q_i
//   mode 1 (BFV multiply): message mod t used directly as residues mod q_i
//   mode 2 (BFV add/sub): scale up by Q/t with rounding: u = (m*[Q]_t + t/2) mod t, out_i = (u - t/2) * (-t^-1) mod q_i
struct LiftArgs {
    const u64* pt;
    u64* out;
    long long spt, sout;
    const ModDev* mods;
    const u64* neg_tinv;   // mode 2: -(t^-1) mod q_i, Montgomery form
    u64 t, qmodt;
    int mode, src_mod, logn, limbs;
};

__global__ __launch_bounds__(TPB) void k_lift_ringt(LiftArgs g) {
    const int chunks = (1 << g.logn) / (2 * TPB);
    const int limb = blockIdx.x / chunks;
    const int x = ((blockIdx.x % chunks) * TPB + threadIdx.x) * 2;
    const long long b = blockIdx.y;
    const ModDev mi = g.mods[limb];
    const ulonglong2 v = ld2(g.pt + b * g.spt + x);
    u64 r[2] = {v.x, v.y};
#pragma unroll
    for (int e = 0; e < 2; e++) {
        if (g.mode == 0) {
            const u64 q0 = g.mods[g.src_mod].q;
            r[e] = r[e] > (q0 >> 1) ? neg_mod(reduce_u64(q0 - r[e], mi), mi.q) : reduce_u64(r[e], mi);
        } else if (g.mode == 1) {
            r[e] = reduce_u64(r[e], mi);
        } else {
            const u64 thalf = g.t >> 1;
            const u64 u = ((r[e] % g.t) * g.qmodt + thalf) % g.t;   // t < 2^32: the product fits 64 bits
            r[e] = mont_mul(sub_mod(reduce_u64(u, mi), reduce_u64(thalf, mi), mi.q), g.neg_tinv[limb], mi.q, mi.qinv);
        }
    }
    st2(g.out + b * g.sout + ((long long)limb << g.logn) + x, r[0], r[1]);
}

void launch_lift_ringt(Context& c, int mode, int level, const u64* pt, long long spt, u64* out, long long sout, int batch,
                       hipStream_t s) {
    if (batch <= 0) return;
    const int L = level + 1;
    LiftArgs g{};
    g.pt = pt;
    g.out = out;
    g.spt = spt;
    g.sout = sout;
    g.mods = c.d_mods;
    g.mode = mode;
    g.src_mod = 0;
    g.logn = c.logn;
    g.limbs = L;
    if (mode == 2) {
        LSA_REQUIRE(c.t > 1 && c.t < (1ull << 32), "ring-t scale-up needs a plaintext modulus below 2^32");
        g.t = c.t;
        u64 qm = 1 % c.t;
        for (int i = 0; i < L; i++) qm = mul_mod_host(qm, c.T.mod[i] % c.t, c.t);
        g.qmodt = qm;
        std::vector<int> mods(L);
        std::vector<u64> v(L);
        for (int i = 0; i < L; i++) {
            mods[i] = i;
            v[i] = c.T.mod[i] - inv_mod(c.t % c.T.mod[i], c.T.mod[i]);
        }
        g.neg_tinv = c.const_vec("neg_tinv" + std::to_string(L), mods, v);
    }
    ProfScope ps(c, PROF_ELEMWISE, 8.0 * c.n * (1 + L) * batch, s);
    hipLaunchKernelGGL(k_lift_ringt, ew_grid(c, L, batch), dim3(TPB), 0, s, g);
    LSA_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ probes (ceilings for DESIGN.md)
__global__ __launch_bounds__(TPB) void k_probe_copy(u64* dst, const u64* src, size_t n2) {
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n2; i += (size_t)gridDim.x * TPB) {
        const ulonglong2 v = ld2(src + 2 * i);
        st2(dst + 2 * i, v.x, v.y);
    }
}
void launch_probe_copy(u64* dst, const u64* src, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_probe_copy, dim3(256 * 8), dim3(TPB), 0, s, dst, src, n / 2);
    LSA_HIP(hipGetLastError());
}

// dependent chains of Montgomery multiplies, 4 independent chains per lane: measures the 64-bit modmul issue ceiling
__global__ __launch_bounds__(TPB) void k_probe_mulhi(u64* buf, size_t n, int iters) {
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i * 4 + 3 >= n) return;
    const u64 q = 0x1FFFFFFFFFE00001ull, qinv = 1;  // constants irrelevant for timing
    u64 a = buf[4 * i], b = buf[4 * i + 1], c = buf[4 * i + 2], d = buf[4 * i + 3];
    const u64 w = a | 1;
    for (int it = 0; it < iters; it++) {
        a = mont_mul_lazy(a, w, q, qinv + it);
        b = mont_mul_lazy(b, w, q, qinv + it);
        c = mont_mul_lazy(c, w, q, qinv + it);
        d = mont_mul_lazy(d, w, q, qinv + it);
    }
    buf[4 * i] = a;
    buf[4 * i + 1] = b;
    buf[4 * i + 2] = c;
    buf[4 * i + 3] = d;
}
// the same chains with Shoup/Harvey multiplication by a constant with precomputed quotient w' = floor(w*2^64/q):
// r = w*v - mulhi(w', v)*q  (mod 2^64), in [0, 2q)
__device__ __forceinline__ u64 shoup_mul_lazy(u64 v, u64 w, u64 wq, u64 q) {
    const u64 t = mulhi64(wq, v);
    return mul_lo64(w, v) - mul_lo64(t, q);
}
__global__ __launch_bounds__(TPB) void k_probe_shoup(u64* buf, size_t n, int iters) {
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i * 4 + 3 >= n) return;
    const u64 q = 0x1FFFFFFFFFE00001ull;
    u64 a = buf[4 * i], b = buf[4 * i + 1], c = buf[4 * i + 2], d = buf[4 * i + 3];
    const u64 w = a | 1, wq = b | 1;
    for (int it = 0; it < iters; it++) {
        a = shoup_mul_lazy(a, w, wq + it, q);
        b = shoup_mul_lazy(b, w, wq + it, q);
        c = shoup_mul_lazy(c, w, wq + it, q);
        d = shoup_mul_lazy(d, w, wq + it, q);
    }
    buf[4 * i] = a;
    buf[4 * i + 1] = b;
    buf[4 * i + 2] = c;
    buf[4 * i + 3] = d;
}
// the same chains on the FP64 engine's exact modular product (6 double operations: mul, fma, mul, rndne, fma, add)
__global__ __launch_bounds__(TPB) void k_probe_fp(u64* buf, size_t n, int iters) {
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i * 4 + 3 >= n) return;
    const double q = 70368744161281.0, qinv = 1.0 / q;   // a 46-bit modulus; values irrelevant for timing
    double a = (double)(buf[4 * i] & 0xFFFFFFFFFFull), b = (double)(buf[4 * i + 1] & 0xFFFFFFFFFFull);
    double c = (double)(buf[4 * i + 2] & 0xFFFFFFFFFFull), d = (double)(buf[4 * i + 3] & 0xFFFFFFFFFFull);
    const double w = a + 1.0;
    for (int it = 0; it < iters; it++) {
        a = fp_modmul(a, w + it, q, qinv);
        b = fp_modmul(b, w + it, q, qinv);
        c = fp_modmul(c, w + it, q, qinv);
        d = fp_modmul(d, w + it, q, qinv);
    }
    buf[4 * i] = (u64)(long long)a;
    buf[4 * i + 1] = (u64)(long long)b;
    buf[4 * i + 2] = (u64)(long long)c;
    buf[4 * i + 3] = (u64)(long long)d;
}
void launch_probe_mulhi(u64* buf, size_t n, int iters, hipStream_t s) {
    if (iters >= (1 << 20)) {   // offset by 2^20: the FP64 variant
        const size_t threads = n / 4;
        hipLaunchKernelGGL(k_probe_fp, dim3((unsigned)((threads + TPB - 1) / TPB)), dim3(TPB), 0, s, buf, n, iters - (1 << 20));
        LSA_HIP(hipGetLastError());
        return;
    }
    if (iters < 0) {   // negative: the Shoup variant
        const size_t threads = n / 4;
        hipLaunchKernelGGL(k_probe_shoup, dim3((unsigned)((threads + TPB - 1) / TPB)), dim3(TPB), 0, s, buf, n, -iters);
        LSA_HIP(hipGetLastError());
        return;
    }
    const size_t threads = n / 4;
    hipLaunchKernelGGL(k_probe_mulhi, dim3((unsigned)((threads + TPB - 1) / TPB)), dim3(TPB), 0, s, buf, n, iters);
    LSA_HIP(hipGetLastError());
}

}  // namespace lsa
