// modarith.h — 64-bit modular arithmetic for gfx950 (no native 64x64->128: built from v_mad_u64_u32 / v_mul_hi_u32
// sequences the compiler emits for __umul64hi).  Montgomery radix 2^64, primes up to 61 bits so that 4q < 2^63 and
// Harvey-style lazy butterflies stay inside one 64-bit word.
//
// The same functions compile for the host (unsigned __int128) so the kernel phase functions can be replayed
// thread-by-thread on the CPU (LSA_EMULATE, tests/test_emulate_ntt.py) — that is a debugging aid for the kernel's
// own indexing, not a product fallback: the library refuses to run without a GPU.
#pragma once
#include <stdint.h>

typedef uint64_t u64;
typedef uint32_t u32;

#if defined(LSA_EMULATE)
#define LSA_HD inline
#define LSA_D inline
#else
#include <hip/hip_runtime.h>
#define LSA_HD __host__ __device__ __forceinline__
#define LSA_D __device__ __forceinline__
#endif

struct ModDev {
    u64 q;      // modulus
    u64 qinv;   // q^-1 mod 2^64   (Lattigo MRedParams convention: r = hi - mulhi(lo*qinv, q))
    u64 r2;     // 2^128 mod q     (to-Montgomery constant)
    u64 r1;     // 2^64 mod q      (Montgomery one)
};

// ---- 64x64 multiplies built explicitly from 32-bit halves.  Each line below is one v_mad_u64_u32 (32x32+64 -> 64) or
// v_mul_hi/lo_u32; hipcc's own lowering of `__umul64hi(a,b)` next to `a*b` recomputes the partial products
// (measured: ~20 multiplier ops per Montgomery multiply instead of the 11 written here; profiles/ntt_r1a).
struct U128 {
    u64 hi, lo;
};
LSA_HD U128 mul_wide(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 al = (u32)a, ah = (u32)(a >> 32), bl = (u32)b, bh = (u32)(b >> 32);
    const u64 p0 = (u64)al * bl;
    const u64 p1 = (u64)ah * bl + (p0 >> 32);
    const u64 p2 = (u64)al * bh + (u32)p1;
    const u64 p3 = (u64)ah * bh + ((p1 >> 32) + (p2 >> 32));
    U128 r;
    r.hi = p3;
    r.lo = (p2 << 32) | (u32)p0;
    return r;
#else
    const unsigned __int128 p = (unsigned __int128)a * b;
    U128 r;
    r.hi = (u64)(p >> 64);
    r.lo = (u64)p;
    return r;
#endif
}
// low 64 bits of a*b: one mad + two 32-bit low multiplies
LSA_HD u64 mul_lo64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 al = (u32)a, ah = (u32)(a >> 32), bl = (u32)b, bh = (u32)(b >> 32);
    const u64 p0 = (u64)al * bl;
    const u32 hi = (u32)(p0 >> 32) + al * bh + ah * bl;
    return ((u64)hi << 32) | (u32)p0;
#else
    return a * b;
#endif
}
// high 64 bits of a*b
LSA_HD u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 al = (u32)a, ah = (u32)(a >> 32), bl = (u32)b, bh = (u32)(b >> 32);
    const u32 c0 = __umulhi(al, bl);
    const u64 p1 = (u64)ah * bl + c0;
    const u64 p2 = (u64)al * bh + (u32)p1;
    return (u64)ah * bh + ((p1 >> 32) + (p2 >> 32));
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

// REDC of the 128-bit value (hi,lo) < q*2^64  ->  [0, 2q)   ("lazy": caller reduces when it must)
LSA_HD u64 mont_redc_lazy(u64 hi, u64 lo, u64 q, u64 qinv) {
    const u64 m = mul_lo64(lo, qinv);
    const u64 t = mulhi64(m, q);
    return hi - t + q;
}
// a*b*2^-64 mod q in [0,2q); requires a*b < q*2^64 (e.g. a < 2^63, b < q)
LSA_HD u64 mont_mul_lazy(u64 a, u64 b, u64 q, u64 qinv) {
    const U128 p = mul_wide(a, b);
    return mont_redc_lazy(p.hi, p.lo, q, qinv);
}
// Shoup / Harvey multiplication by a constant w < q with its precomputed quotient ws = floor(w * 2^64 / q):
// w*v - mulhi(ws, v)*q (mod 2^64) is w*v mod q in [0, 2q) for any 64-bit v.  Measured on gfx950 (tools/probe_mul.py):
// 2.65e12 against 1.66e12 Montgomery multiplies per second -- the low halves of two products cost less than REDC's.
LSA_HD u64 shoup_mul_lazy(u64 v, u64 w, u64 ws, u64 q);
// (measured and dropped: flag-free formulations of the 64-bit compare/select and subtract -- sign masks, a + ~b + 1 through
// v_lshl_add_u64 -- remove the VCC hazards hipcc's lowering has, but the extra register moves cost more: integer engine
// 1.63 -> 1.27 TB/s; adding t*(2^64-q) instead of subtracting t*q drops a third of the hazard s_nops but 9 more VGPRs cost
// the plain kernel its fourth workgroup per CU: 1.63 -> 1.53, and 1.62 when forced back to 128 VGPRs)
LSA_HD u64 sub64(u64 a, u64 b) { return a - b; }
LSA_HD u64 csub(u64 a, u64 q) { return a >= q ? a - q : a; }
LSA_HD u64 shoup_mul_lazy(u64 v, u64 w, u64 ws, u64 q) { return sub64(mul_lo64(w, v), mul_lo64(mulhi64(ws, v), q)); }

// ---- the NTT butterfly's multiply, written for what the instructions cost on gfx950 (tools/probe_issue.hip, cycles per
// wave64 instruction and SIMD): v_mad_u64_u32 5.2, v_mul_lo/hi_u32 4.5-4.9, v_lshl_add_u64 4.6, 32-bit add/logic 2.6, but
// a flag-carried 64-bit add or subtract (v_add_co/v_addc pair) 7 and a 64-bit compare-select-subtract 18.4: the adds,
// compares and register moves around the ten multiplies were 60 % of the exact form's 128 cycles per butterfly.
//   * quotient from the three high partial products only (two v_mul_hi + one multiply-add): t' in [t-2, t], so the product
//     lands in [0, 4q) instead of [0, 2q) -- the range the extra bits of a 64-bit word pay for (q < 2^61: 8q < 2^64);
//   * w*v + t'*(2^64 - q) as ONE chain of multiply-adds (the low-word cross terms accumulate in a chain of their own whose
//     upper word is never read), no separate subtract;
//   * the conditional subtraction decided by the sign of a - 4q (v_lshl_add_u64 + one compare + two selects).
// One v_mad_u64_u32 each; the carry-out operand is unused.
LSA_HD u64 mad64(u32 a, u32 b, u64 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 d, cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "v"(b), "v"(c));
    return d;
#else
    return (u64)a * b + c;
#endif
}
LSA_HD u64 mul64(u32 a, u32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 d, cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(cy) : "v"(a), "v"(b));
    return d;
#else
    return (u64)a * b;
#endif
}
LSA_HD u32 mulhi32(u32 a, u32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (u32)(((u64)a * b) >> 32);
#endif
}
// w*v mod q in [0, 4q) for ANY 64-bit v; w < q with its quotient ws = floor(w * 2^64 / q), nq = 2^64 - q
// Device form: the nine multiply-adds as TWO asm blocks.  hipcc puts an `s_nop 0` after every single-instruction asm
// statement that is followed by another one (it cannot see inside and plays safe; its own back-to-back dependent
// v_mad_u64_u32 carry none -- the hardware interlocks them): written one per statement the product had 6 of them and two
// register moves for the final recombination, a fifth of the butterfly's issue slots.
LSA_HD u64 shoup_mul_approx(u64 v, u64 w, u64 ws, u64 nq) {
    const u32 vl = (u32)v, vh = (u32)(v >> 32), wl = (u32)w, wh = (u32)(w >> 32), sl = (u32)ws, sh = (u32)(ws >> 32);
    const u32 nl = (u32)nq, nh = (u32)(nq >> 32);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LSA_SHOUP_SINGLE_ASM)
    const u64 t0 = (u64)mulhi32(vh, sl);
    const u32 t1 = mulhi32(vl, sh);
    u64 t, p, r, cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4\n\t"
        "v_mad_u64_u32 %0, %1, %5, 1, %0"
        : "=&v"(t), "=&s"(cy)
        : "v"(vh), "v"(sh), "v"(t0), "v"(t1));
    const u32 tl = (u32)t, th = (u32)(t >> 32);
    asm("v_mad_u64_u32 %0, %2, %3, %6, 0\n\t"
        "v_mad_u64_u32 %0, %2, %4, %5, %0\n\t"
        "v_mad_u64_u32 %0, %2, %7, %10, %0\n\t"
        "v_mad_u64_u32 %0, %2, %8, %9, %0\n\t"
        "v_mad_u64_u32 %1, %2, %3, %5, 0\n\t"
        "v_mad_u64_u32 %1, %2, %7, %9, %1\n\t"
        "v_lshlrev_b64 %0, 32, %0\n\t"          // r + (p << 32) on whole register pairs: composing the pair from halves in
        "v_lshl_add_u64 %1, %0, 0, %1"            // C cost up to two register moves per product
        : "=&v"(p), "=&v"(r), "=&s"(cy)
        : "v"(vl), "v"(vh), "v"(wl), "v"(wh), "v"(tl), "v"(th), "v"(nl), "v"(nh));
    return r;
#else
    const u64 t = mad64(vh, sh, (u64)mulhi32(vh, sl)) + mulhi32(vl, sh);
    const u32 tl = (u32)t, th = (u32)(t >> 32);
    u64 p = mul64(vl, wh);   // low-word cross terms; the upper word of this chain is never read
    p = mad64(vh, wl, p);
    p = mad64(tl, nh, p);
    p = mad64(th, nl, p);
    u64 r = mul64(vl, wl);
    r = mad64(tl, nl, r);
    return r + ((u64)(u32)p << 32);
#endif
}
// a in [0, 2m), m < 2^63: a - m if a >= m (nm = 2^64 - m)
LSA_HD u64 csub_sign(u64 a, u64 nm) {
    const u64 d = a + nm;
    return (int)(u32)(d >> 32) < 0 ? a : d;
}
// a*b*2^-64 mod q in [0,q)
LSA_HD u64 mont_mul(u64 a, u64 b, u64 q, u64 qinv) { return csub(mont_mul_lazy(a, b, q, qinv), q); }
// plain a*b mod q for a,b in [0,q): two REDCs (a*b*R^-1, then *R^2*R^-1)
LSA_HD u64 mul_mod(u64 a, u64 b, const ModDev& m) {
    return mont_mul(mont_mul_lazy(a, b, m.q, m.qinv), m.r2, m.q, m.qinv);
}
LSA_HD u64 add_mod(u64 a, u64 b, u64 q) { return csub(a + b, q); }
LSA_HD u64 sub_mod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
LSA_HD u64 neg_mod(u64 a, u64 q) { return a ? q - a : 0; }

// 128-bit accumulate helper: (hi,lo) += a*b
LSA_HD void mac128(u64& hi, u64& lo, u64 a, u64 b) {
    const U128 p = mul_wide(a, b);
#if defined(LSA_MAC128_COMPARE)   // A/B: carry recovered by a 64-bit compare (18 cycles on gfx950 against 7 for the add-with-carry pair)
    lo += p.lo;
    hi += p.hi + (lo < p.lo ? 1 : 0);
#else
    const unsigned __int128 acc = (((unsigned __int128)hi << 64) | lo) + (((unsigned __int128)p.hi << 64) | p.lo);
    lo = (u64)acc;
    hi = (u64)(acc >> 64);
#endif
}
// x mod q for arbitrary 64-bit x and q > 2^32-ish chain primes of any size: via Montgomery (x*R^-1 then *R^2)
LSA_HD u64 reduce_u64(u64 x, const ModDev& m) {
    return mont_mul(mont_redc_lazy(0, x, m.q, m.qinv), m.r2, m.q, m.qinv);
}
