// Host-side check of the integer primitives the NTT butterfly and the 128-bit accumulates are built from
// (lattisense_amd/csrc/modarith.h; the device build runs the same formulas on v_mad_u64_u32 chains): against unsigned
// __int128 arithmetic, on random and edge operands, for moduli of 40 to 61 bits.  Compiled and run by tests/test_modarith_host.py.
#include <cstdio>
#include <cstdlib>
#include "../../lattisense_amd/csrc/modarith.h"

typedef unsigned __int128 u128;
static u64 rng_state = 0x243F6A8885A308D3ull;
static u64 rnd() {
    u64 z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s (line %d) q=%llu\n", #c, __LINE__, (unsigned long long)q); return 1; } } while (0)

int main() {
    const u64 primes[] = {0x1fffffffffe00001ull, 0xffffffffffc0001ull, 0x7fffffffe90001ull, 1099511922689ull /* 40 bit */, 35184372121601ull /* 45 bit */};
    for (u64 q : primes) {
        const u64 nq = 0 - q;
        for (int it = 0; it < 2000000; it++) {
            u64 v = rnd(), w = rnd() % q;
            if (it < 16) v = it & 1 ? ~0ull : 0;            // edges
            if (it >= 16 && it < 32) w = it & 1 ? q - 1 : 0;
            if (it >= 32 && it < 64) v = (u64)(it - 31) * q - (it & 1);   // multiples of q and their neighbours
            const u64 ws = (u64)(((u128)w << 64) / q);
            const u64 exact = (u64)(((u128)w * v) % q);
            const u64 r = shoup_mul_approx(v, w, ws, nq);
            CHECK(r < 4 * q || 4 * q < q);   // (4q wraps only for q >= 2^62, which build() refuses)
            CHECK(r % q == exact);
            const u64 r2 = shoup_mul_lazy(v, w, ws, q);
            CHECK(r2 < 2 * q && r2 % q == exact);
            // conditional subtraction by the sign of a - m, for a in [0, 2m), m = q, 2q, 4q
            for (int k = 0; k < 3; k++) {
                const u64 m = q << k, a = m >= (1ull << 63) ? 0 : rnd() % (2 * m);
                if (m >> 63) continue;
                CHECK(csub_sign(a, 0 - m) == (a >= m ? a - m : a));
                CHECK(csub_sign(m, 0 - m) == 0 && csub_sign(m - 1, 0 - m) == m - 1 && csub_sign(2 * m - 1, 0 - m) == m - 1);
            }
            // 128-bit multiply-accumulate
            u64 hi = rnd() >> 2, lo = rnd();
            const u128 want = (((u128)hi << 64) | lo) + (u128)v * w;
            mac128(hi, lo, v, w);
            CHECK(hi == (u64)(want >> 64) && lo == (u64)want);
            CHECK(mulhi64(v, w) == (u64)(((u128)v * w) >> 64) && mul_lo64(v, w) == v * w);
        }
    }
    std::printf("OK modarith\n");
    return 0;
}
