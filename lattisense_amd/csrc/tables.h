// tables.h — host-side precomputation for a parameter set (the part of HEonGPU's HEContext::generate that
// mega_ag_runners/gpu/gpu_wrapper.cu:53-138 rebuilds on every run; here it is built once per context and cached).
#pragma once
#include <cstdint>
#include <vector>
#include "modarith.h"

namespace lsa {

bool is_prime64(u64 n);
u64 pow_mod(u64 a, u64 e, u64 q);
u64 mul_mod_host(u64 a, u64 b, u64 q);
u64 inv_mod(u64 a, u64 q);
u64 smallest_primitive_root(u64 q);
int product_bitlen(const u64* q, int k);
// BFV auxiliary basis: count = ceil((bitlen(prod q) + logn)/61); primes = 61-bit, == 1 mod 2n, descending from 2^61
int bfv_aux_count(const u64* q, int k, int logn);
std::vector<u64> gen_aux_primes(int n, int count, const std::vector<u64>& avoid);

struct HostTables {
    int n = 0, logn = 0;
    std::vector<u64> mod;       // all moduli: Q chain, then P, then BFV aux
    std::vector<ModDev> mods;   // Montgomery constants per modulus
    // integer engine: every twiddle w with its Shoup quotient floor(w * 2^64 / q), interleaved {w, ws}
    std::vector<u64> psi;       // [nmod][n][2]  w = psi^{brv(x)}
    std::vector<u64> psiinv;    // [nmod][n][2]  w = psi^{-brv(x)}
    std::vector<u64> scale;     // [nmod][2][2]  w = n^-1 and psiinv[1] * n^-1 (the inverse transform's last stage)
    // the same tables as plain integer-valued doubles for the FP64 butterfly engine (used for q < 2^47 only)
    std::vector<double> psi_d, psiinv_d, scale_d;
    void build(int n_, const std::vector<u64>& moduli);
};

inline u64 to_mont_host(u64 a, u64 q) { return (u64)(((unsigned __int128)a << 64) % q); }
inline u64 shoup_quotient_host(u64 a, u64 q) { return (u64)(((unsigned __int128)a << 64) / q); }

}  // namespace lsa
