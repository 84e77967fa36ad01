// tables.cpp — number theory + per-modulus table generation (host only).
#include "tables.h"
#include <stdexcept>
#include <string>

namespace lsa {

typedef unsigned __int128 u128;

u64 mul_mod_host(u64 a, u64 b, u64 q) { return (u64)((u128)a * b % q); }

u64 pow_mod(u64 a, u64 e, u64 q) {
    u64 r = 1 % q;
    a %= q;
    for (; e; e >>= 1) {
        if (e & 1) r = mul_mod_host(r, a, q);
        a = mul_mod_host(a, a, q);
    }
    return r;
}

u64 inv_mod(u64 a, u64 q) { return pow_mod(a % q, q - 2, q); }

bool is_prime64(u64 n) {
    if (n < 2) return false;
    const u64 bases[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (u64 b : bases)
        if (n % b == 0) return n == b;
    u64 d = n - 1;
    int s = 0;
    while (!(d & 1)) {
        d >>= 1;
        s++;
    }
    for (u64 b : bases) {
        u64 x = pow_mod(b, d, n);
        if (x == 1 || x == n - 1) continue;
        bool witness = true;
        for (int r = 1; r < s && witness; r++) {
            x = mul_mod_host(x, x, n);
            if (x == n - 1) witness = false;
        }
        if (witness) return false;
    }
    return true;
}

static u64 gcd_u64(u64 a, u64 b) {
    while (b) {
        u64 t = a % b;
        a = b;
        b = t;
    }
    return a;
}

static u64 rho_factor(u64 n) {
    if (!(n & 1)) return 2;
    for (u64 c = 1;; c++) {
        u64 x = 2, y = 2, d = 1;
        auto f = [&](u64 v) { return (mul_mod_host(v, v, n) + c) % n; };
        while (d == 1) {
            x = f(x);
            y = f(f(y));
            d = gcd_u64(x > y ? x - y : y - x, n);
        }
        if (d != n) return d;
    }
}

static void distinct_prime_factors(u64 n, std::vector<u64>& out) {
    if (n == 1) return;
    if (is_prime64(n)) {
        for (u64 f : out)
            if (f == n) return;
        out.push_back(n);
        return;
    }
    u64 d = rho_factor(n);
    distinct_prime_factors(d, out);
    distinct_prime_factors(n / d, out);
}

// ABI canonical root choice: psi = g^((q-1)/2N) with g the smallest generator of Z_q^*.
u64 smallest_primitive_root(u64 q) {
    std::vector<u64> fac;
    u64 m = q - 1;
    for (u64 p = 2; p < 1000 && m > 1; p++) {
        if (m % p == 0) {
            fac.push_back(p);
            while (m % p == 0) m /= p;
        }
    }
    distinct_prime_factors(m, fac);
    for (u64 g = 2;; g++) {
        bool gen = true;
        for (u64 f : fac)
            if (pow_mod(g, (q - 1) / f, q) == 1) {
                gen = false;
                break;
            }
        if (gen) return g;
    }
}

int product_bitlen(const u64* q, int k) {
    std::vector<u64> w(1, 1);
    for (int i = 0; i < k; i++) {
        u64 carry = 0;
        for (auto& wj : w) {
            u128 t = (u128)wj * q[i] + carry;
            wj = (u64)t;
            carry = (u64)(t >> 64);
        }
        if (carry) w.push_back(carry);
    }
    int bl = (int)(w.size() - 1) * 64;
    for (u64 top = w.back(); top; top >>= 1) bl++;
    return bl;
}

int bfv_aux_count(const u64* q, int k, int logn) { return (product_bitlen(q, k) + logn + 60) / 61; }

std::vector<u64> gen_aux_primes(int n, int count, const std::vector<u64>& avoid) {
    std::vector<u64> out;
    const u64 step = 2 * (u64)n;
    u64 x = ((u64)1 << 61) + 1;
    while ((int)out.size() < count) {
        if (x <= step) throw std::runtime_error("gen_aux_primes: exhausted");
        x -= step;
        if (!is_prime64(x)) continue;
        bool clash = false;
        for (u64 a : avoid) clash |= (a == x);
        if (!clash) out.push_back(x);
    }
    return out;
}

static unsigned bit_reverse(unsigned x, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; i++) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
}

void HostTables::build(int n_, const std::vector<u64>& moduli) {
    n = n_;
    logn = 0;
    while ((1 << logn) < n) logn++;
    if ((1 << logn) != n) throw std::runtime_error("ring degree must be a power of two");
    mod = moduli;
    const size_t nm = mod.size();
    mods.resize(nm);
    psi.assign(nm * (size_t)n * 2, 0);
    psiinv.assign(nm * (size_t)n * 2, 0);
    scale.assign(nm * 4, 0);
    psi_d.assign(nm * (size_t)n, 0.0);
    psiinv_d.assign(nm * (size_t)n, 0.0);
    scale_d.assign(nm * 2, 0.0);
    for (size_t i = 0; i < nm; i++) {
        const u64 q = mod[i];
        if (q >> 61) throw std::runtime_error("modulus exceeds 61 bits: " + std::to_string(q));
        if (!is_prime64(q) || (q - 1) % (2 * (u64)n) != 0)
            throw std::runtime_error("modulus is not an NTT prime for this ring degree: " + std::to_string(q));
        ModDev m;
        m.q = q;
        u64 inv = 1;  // Newton iteration for q^-1 mod 2^64
        for (int it = 0; it < 6; it++) inv *= 2 - q * inv;
        m.qinv = inv;
        m.r1 = to_mont_host(1, q);
        m.r2 = to_mont_host(m.r1, q);
        mods[i] = m;
        const u64 g = smallest_primitive_root(q);
        const u64 ps = pow_mod(g, (q - 1) / (2 * (u64)n), q);
        const u64 psi_inv = inv_mod(ps, q);
        u64 pw = 1, pwi = 1;
        u64* t = &psi[i * (size_t)n * 2];
        u64* ti = &psiinv[i * (size_t)n * 2];
        for (int j = 0; j < n; j++) {
            unsigned x = bit_reverse((unsigned)j, logn);
            t[2 * x] = pw;
            t[2 * x + 1] = shoup_quotient_host(pw, q);
            ti[2 * x] = pwi;
            ti[2 * x + 1] = shoup_quotient_host(pwi, q);
            if ((q >> 53) == 0) {  // exactly representable; only consumed when q < 2^47
                psi_d[i * (size_t)n + x] = (double)pw;
                psiinv_d[i * (size_t)n + x] = (double)pwi;
            }
            pw = mul_mod_host(pw, ps, q);
            pwi = mul_mod_host(pwi, psi_inv, q);
        }
        const u64 ninv = inv_mod((u64)n % q, q);
        scale[4 * i] = ninv;
        scale[4 * i + 1] = shoup_quotient_host(ninv, q);
        // psiinv[1] = psi^{-brv(1)} = psi^{-n/2}
        u64 w1 = pow_mod(psi_inv, (u64)n / 2, q);
        scale[4 * i + 2] = mul_mod_host(w1, ninv, q);
        scale[4 * i + 3] = shoup_quotient_host(scale[4 * i + 2], q);
        if ((q >> 53) == 0) {
            scale_d[2 * i] = (double)ninv;
            scale_d[2 * i + 1] = (double)mul_mod_host(w1, ninv, q);
        }
    }
}

}  // namespace lsa
