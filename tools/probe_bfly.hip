// probe_bfly.hip — issue cost of the integer-engine NTT butterfly in isolation, by formulation.
// Each thread keeps 16 points in registers and runs radix-16 sub-passes (4 stages x 8 butterflies, 15 twiddle pairs per
// sub-pass loaded from an L1-resident table, as k_ntt_pass does between two LDS crossings); no LDS / HBM traffic inside the
// timed loop.  Variants:
//   0  the kernel's current butterfly: Harvey [0,4q) range, exact Shoup quotient (mulhi64), compiler-lowered 64-bit ops
//   1  approximate quotient (two mul_hi + one mad; t' in [t-2,t], product in [0,4q)), w*v + t'*(2^64-q) as one chain of
//      v_mad_u64_u32 accumulations, sign-test conditional subtraction, [0,8q) range
//   2  variant 1 without the per-butterfly conditional subtraction (moduli below 2^57: 65q < 2^64 over 16 stages)
// `check` runs every variant on random operands against a Montgomery reference (bit-exact after full reduction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../lattisense_amd/csrc/modarith.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct QC {
    u64 q, q2, q4, nq, nq4;
};

__device__ __forceinline__ u64 mad64(u32 a, u32 b, u64 c) {   // a*b + c mod 2^64: exactly one v_mad_u64_u32
    u64 d, cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ u64 mul64(u32 a, u32 b) {
    u64 d, cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(cy) : "v"(a), "v"(b));
    return d;
}
// w*v mod q in [0,4q) for any 64-bit v: quotient estimate from the three high partial products only
__device__ __forceinline__ u64 shoup_mul_approx(u64 v, u64 w, u64 ws, const QC& c) {
    const u32 vl = (u32)v, vh = (u32)(v >> 32), wl = (u32)w, wh = (u32)(w >> 32), sl = (u32)ws, sh = (u32)(ws >> 32);
    const u32 x = __umulhi(vh, sl), y = __umulhi(vl, sh);
    const u64 t = mad64(vh, sh, (u64)x) + y;
    const u32 tl = (u32)t, th = (u32)(t >> 32), nl = (u32)c.nq, nh = (u32)(c.nq >> 32);
    u64 p = mul64(vl, wh);      // only the low word of this chain is used
    p = mad64(vh, wl, p);
    p = mad64(tl, nh, p);
    p = mad64(th, nl, p);
    u64 r = mul64(vl, wl);
    r = mad64(tl, nl, r);
    return r + ((u64)(u32)p << 32);
}
__device__ __forceinline__ u64 csub_sign(u64 a, u64 nm) {   // a in [0, 2m), m < 2^63: a - m if a >= m (nm = 2^64 - m)
    const u64 d = a + nm;
    return (int)(d >> 32) < 0 ? a : d;
}

// ---- variant 3: no zero-extended register pairs and no flag-carried 64-bit ops inside the product
__device__ __forceinline__ u64 madd1(u32 x, u64 c) {   // c + zext(x) as a multiply-add by the constant 1
    u64 d, cy;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(d), "=s"(cy) : "v"(x), "v"(c));
    return d;
}
__device__ __forceinline__ u32 add32(u32 a, u32 b) {
    u32 d;
    asm("v_add_u32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ u64 pack64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
// base + w*v mod q, the product part in [0,4q): the addend rides in the multiply-add chain
__device__ __forceinline__ u64 shoup_mac3(u64 base, u64 v, u64 w, u64 ws, const QC& c) {
    const u32 vl = (u32)v, vh = (u32)(v >> 32), wl = (u32)w, wh = (u32)(w >> 32), sl = (u32)ws, sh = (u32)(ws >> 32);
    const u32 x = __umulhi(vh, sl), y = __umulhi(vl, sh);
    u64 t = mul64(vh, sh);
    t = madd1(x, t);
    t = madd1(y, t);
    const u32 tl = (u32)t, th = (u32)(t >> 32), nl = (u32)c.nq, nh = (u32)(c.nq >> 32);
    u64 p = mul64(vl, wh);
    p = mad64(vh, wl, p);
    p = mad64(tl, nh, p);
    p = mad64(th, nl, p);
    u64 r = mad64(vl, wl, base);
    r = mad64(tl, nl, r);
    return pack64((u32)r, add32((u32)(r >> 32), (u32)p));
}
// a < 8q + 2^32 -> a or a - 4q, below 4q + 2^32: decided on the high word alone (khi = high word of 4q)
template <int FLAGS> __device__ __forceinline__ u64 csub_hi(u64 a, const QC& c) {
    const u32 khi = (u32)(c.q4 >> 32);
    if (FLAGS) {
        const bool ge = (u32)(a >> 32) > khi;
        return a + pack64(ge ? (u32)c.nq4 : 0u, ge ? (u32)(c.nq4 >> 32) : 0u);
    }
    const u32 m = (u32)((int)(khi - (u32)(a >> 32)) >> 31);
    return a + pack64(m & (u32)c.nq4, m & (u32)(c.nq4 >> 32));
}
template <int FLAGS> __device__ __forceinline__ void bf_fwd3(u64& a, u64& b, u64 w, u64 ws, const QC& c) {
    const u64 U = csub_hi<FLAGS>(a, c);
    const u64 o0 = shoup_mac3(U, b, w, ws, c);
    a = o0;
    b = ((U << 1) + c.q4) - o0;
}
template <int FLAGS> __device__ __forceinline__ void bf_inv3(u64& a, u64& b, u64 w, u64 ws, const QC& c) {
    const u64 U = a, V = b;
    a = csub_hi<FLAGS>(U + V, c);
    b = shoup_mac3(0, (U + c.q4) - V, w, ws, c);
}

template <int VAR> __device__ __forceinline__ void bf_fwd(u64& a, u64& b, u64 w, u64 ws, const QC& c) {
    if (VAR == 3 || VAR == 4) {
        bf_fwd3<VAR == 4>(a, b, w, ws, c);
    } else if (VAR == 0) {
        const u64 U = csub(a, c.q2), T = shoup_mul_lazy(b, w, ws, c.q);
        a = U + T;
        b = sub64(U + c.q2, T);
    } else {
        const u64 U = VAR == 1 ? csub_sign(a, c.nq4) : a, T = shoup_mul_approx(b, w, ws, c);
        a = U + T;
        b = sub64(U + c.q4, T);
    }
}
template <int VAR> __device__ __forceinline__ void bf_inv(u64& a, u64& b, u64 w, u64 ws, const QC& c) {
    if (VAR == 3 || VAR == 4) {
        bf_inv3<VAR == 4>(a, b, w, ws, c);
    } else if (VAR == 0) {
        const u64 U = a, V = b;
        a = csub(U + V, c.q2);
        b = shoup_mul_lazy(sub64(U + c.q2, V), w, ws, c.q);
    } else {
        const u64 U = a, V = b;
        a = csub_sign(U + V, c.nq4);
        b = shoup_mul_approx(sub64(U + c.q4, V), w, ws, c);
    }
}

template <int VAR, bool INV> __global__ __launch_bounds__(256) void k_bfly(u64* data, const u64* tw, QC c, int iters) {
    u64 v[16];
    const long long base = ((long long)blockIdx.x * 256 + threadIdx.x) * 16;
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = data[base + e];
    for (int it = 0; it < iters; it++) {
        const u64* t0 = tw + 2 * ((((threadIdx.x >> 10) + it) & 63) * 16);   // one line per wave: the table walk costs no address cycles
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int j = INV ? 3 - jj : jj;
            const int half = 16 >> (j + 1);
            const u64* twj = t0 + 2 * ((1 << j) - 1);
#pragma unroll
            for (int e = 0; e < 16; e++)
                if ((e & half) == 0) {
                    const int ti = 2 * (e >> (4 - j));
                    if (INV) bf_inv<VAR>(v[e], v[e + half], twj[ti], twj[ti + 1], c);
                    else bf_fwd<VAR>(v[e], v[e + half], twj[ti], twj[ti + 1], c);
                }
        }
    }
#pragma unroll
    for (int e = 0; e < 16; e++) data[base + e] = v[e];
}

__device__ __forceinline__ u64 full_reduce(u64 x, u64 q) {
    for (int k = 3; k >= 0; k--) x = csub(x, q << k);   // x < 16q
    return x;
}
// out[0..] mismatches: one butterfly per thread per variant against (a + w b) mod q, (a - w b) mod q
__global__ void k_check(const u64* in, ModDev md, QC c, unsigned* bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const u64 w = in[4 * i + 2], ws = in[4 * i + 3];   // w < q and its Shoup quotient, from the host
    const u64 a4 = in[4 * i] % c.q4, b4 = in[4 * i + 1] % c.q4;          // Harvey range
    const u64 a8 = in[4 * i] % (2 * c.q4), b8 = in[4 * i + 1] % (2 * c.q4);  // [0,8q)
    auto ref = [&](u64 a, u64 b, u64& s, u64& d) {
        const u64 ar = a % c.q, br = b % c.q, p = mul_mod(w, br, md);
        s = add_mod(ar, p, c.q);
        d = sub_mod(ar, p, c.q);
    };
    auto refi = [&](u64 a, u64 b, u64& s, u64& d) {
        const u64 ar = a % c.q, br = b % c.q;
        s = add_mod(ar, br, c.q);
        d = mul_mod(w, sub_mod(ar, br, c.q), md);
    };
    u64 s, d, x, y;
    ref(a4, b4, s, d); x = a4; y = b4; bf_fwd<0>(x, y, w, ws, c);
    if (full_reduce(x, c.q) != s || full_reduce(y, c.q) != d || x >= c.q4 || y >= c.q4) atomicAdd(bad + 0, 1);
    ref(a8, b8, s, d); x = a8; y = b8; bf_fwd<1>(x, y, w, ws, c);
    if (full_reduce(x, c.q) != s || full_reduce(y, c.q) != d || x >= 2 * c.q4 || y >= 2 * c.q4) atomicAdd(bad + 1, 1);
    ref(a8, b8, s, d); x = a8; y = b8; bf_fwd<2>(x, y, w, ws, c);
    if (full_reduce(x, c.q) != s || full_reduce(y, c.q) != d) atomicAdd(bad + 2, 1);
    const u64 lim = 2 * c.q4 + (1ull << 32), lim4 = c.q4 + (1ull << 32);
    const u64 a9 = lim > c.q4 ? in[4 * i] % lim : a8, b9 = lim > c.q4 ? in[4 * i + 1] % lim : b8;
    if (lim > c.q4) {   // variant 3/4 ranges hold only while 8q + 2^33 < 2^64
        ref(a9, b9, s, d); x = a9; y = b9; bf_fwd<3>(x, y, w, ws, c);
        if (full_reduce(x, c.q) != s || full_reduce(y, c.q) != d || x >= lim || y >= lim) atomicAdd(bad + 5, 1);
        ref(a9, b9, s, d); x = a9; y = b9; bf_fwd<4>(x, y, w, ws, c);
        if (full_reduce(x, c.q) != s || full_reduce(y, c.q) != d || x >= lim || y >= lim) atomicAdd(bad + 5, 1);
        const u64 a5 = in[4 * i] % lim4, b5 = in[4 * i + 1] % lim4;
        refi(a5, b5, s, d); x = a5; y = b5; bf_inv<3>(x, y, w, ws, c);
        if (full_reduce(x, c.q) != s || full_reduce(y, c.q) != d || x >= lim4 || y >= lim4) atomicAdd(bad + 6, 1);
        refi(a5, b5, s, d); x = a5; y = b5; bf_inv<4>(x, y, w, ws, c);
        if (full_reduce(x, c.q) != s || full_reduce(y, c.q) != d || x >= lim4 || y >= lim4) atomicAdd(bad + 6, 1);
    }
    const u64 a2 = in[4 * i] % c.q2, b2 = in[4 * i + 1] % c.q2;
    refi(a2, b2, s, d); x = a2; y = b2; bf_inv<0>(x, y, w, ws, c);
    if (full_reduce(x, c.q) != s || full_reduce(y, c.q) != d || x >= c.q2 || y >= c.q2) atomicAdd(bad + 3, 1);
    refi(a4, b4, s, d); x = a4; y = b4; bf_inv<1>(x, y, w, ws, c);
    if (full_reduce(x, c.q) != s || full_reduce(y, c.q) != d || x >= c.q4 || y >= c.q4) atomicAdd(bad + 4, 1);
}

static u64 splitmix(u64& s) {
    u64 z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int VAR, bool INV> static void run(const char* name, u64* d, const u64* tw, QC c, int blocks, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    k_bfly<VAR, INV><<<blocks, 256>>>(d, tw, c, 4);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_bfly<VAR, INV><<<blocks, 256>>>(d, tw, c, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bf = (double)blocks * 256 * iters * 32 / (ms * 1e-3);
    // one N = 2^16 limb transform is 2^19 butterflies and 2^20 algorithmic bytes
    printf("%-34s %.3e butterflies/s = %.2f TB/s-equivalent of N=2^16 transforms (%.2f ms)\n", name, bf, bf * 2.0 / 1e12, ms);
}

int main() {
    const u64 primes[3] = {0x1fffffffffe00001ull, 0xffffffffffc0001ull, 0x7fffffffe90001ull};   // 61, 60, 55 bits
    for (int pi = 0; pi < 3; pi++) {
        const u64 q = primes[pi];
        ModDev md;
        md.q = q;
        u64 inv = 1;
        for (int i = 0; i < 6; i++) inv *= 2 - q * inv;
        md.qinv = inv;
        md.r1 = (u64)((((unsigned __int128)1) << 64) % q);
        md.r2 = (u64)(((unsigned __int128)md.r1 * md.r1) % q);
        QC c{q, 2 * q, 4 * q, 0 - q, 0 - 4 * q};
        const int n = 1 << 20;
        std::vector<u64> h(4 * n);
        u64 s = 1234 + pi;
        for (auto& x : h) x = splitmix(s);
        for (int i = 0; i < n; i++) h[4 * i + 2] %= q;
        // edge operands
        h[0] = 0; h[1] = 0; h[2] = 0;
        h[4] = ~0ull; h[5] = ~0ull; h[6] = q - 1;
        h[8] = 4 * q - 1; h[9] = 4 * q - 1; h[10] = q - 1;
        h[12] = 8 * q - 1; h[13] = 8 * q - 1; h[14] = 1;
        for (int i = 0; i < n; i++) h[4 * i + 3] = (u64)((((unsigned __int128)h[4 * i + 2]) << 64) / q);
        u64* din;
        unsigned* bad;
        CK(hipMalloc(&din, h.size() * 8));
        CK(hipMalloc(&bad, 8 * 4));
        CK(hipMemset(bad, 0, 8 * 4));
        CK(hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        k_check<<<n / 256, 256>>>(din, md, c, bad);
        unsigned hb[8];
        CK(hipMemcpy(hb, bad, 8 * 4, hipMemcpyDeviceToHost));
        printf("q = %llu (%d bits): mismatches fwd0 %u fwd1 %u fwd2 %u inv0 %u inv1 %u fwd3/4 %u inv3/4 %u of %d\n", (unsigned long long)q,
               64 - __builtin_clzll(q), hb[0], hb[1], hb[2], hb[3], hb[4], hb[5], hb[6], n);
        CK(hipFree(din));
        CK(hipFree(bad));
    }
    const u64 q = primes[2];
    QC c{q, 2 * q, 4 * q, 0 - q, 0 - 4 * q};
    const int blocks = 256 * 16, iters = 2000;
    u64 *d, *tw;
    CK(hipMalloc(&d, (size_t)blocks * 256 * 16 * 8));
    CK(hipMalloc(&tw, 64 * 16 * 2 * 8));
    std::vector<u64> ht(64 * 16 * 2), hd((size_t)blocks * 256 * 16);
    u64 s = 99;
    for (size_t i = 0; i < ht.size(); i += 2) {
        ht[i] = splitmix(s) % q;
        ht[i + 1] = (u64)((((unsigned __int128)ht[i]) << 64) / q);
    }
    for (auto& x : hd) x = splitmix(s) % q;
    CK(hipMemcpy(tw, ht.data(), ht.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d, hd.data(), hd.size() * 8, hipMemcpyHostToDevice));
    run<0, false>("forward, current", d, tw, c, blocks, iters);
    run<1, false>("forward, mad chain + approx quot.", d, tw, c, blocks, iters);
    run<2, false>("forward, same without cond. sub", d, tw, c, blocks, iters);
    run<3, false>("forward, v3 (mask cond. sub)", d, tw, c, blocks, iters);
    run<4, false>("forward, v3 (flag cond. sub)", d, tw, c, blocks, iters);
    run<0, true>("inverse, current", d, tw, c, blocks, iters);
    run<1, true>("inverse, mad chain + approx quot.", d, tw, c, blocks, iters);
    run<3, true>("inverse, v3 (mask cond. sub)", d, tw, c, blocks, iters);
    run<4, true>("inverse, v3 (flag cond. sub)", d, tw, c, blocks, iters);
    return 0;
}
