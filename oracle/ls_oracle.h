/*
 * ls_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE ONLY; never linked into the product path).
 *
 * Plain-C restatement of the RNS polynomial arithmetic behind LattiSense's GPU executor
 * (reference call sites: mega_ag_runners/gpu/mega_ag_executors_gpu.cu:71-426, data conventions
 * abi/c_types.h:26-60 and plug-in/lattigo/acc/c_struct_import_export.go:23-135).
 *
 * PARITY STATUS: "ciphertext-bit parity UNPINNED".  The arithmetic of the reference lives in two
 * third-party modules that are absent from /root/reference (empty submodules):
 *   - github.com/cipherflow-fhe/lattigo (fork of tuneinsight/lattigo v4, pinned v4.0.0 in
 *     plug-in/lattigo/go.mod:7)  — the CPU path;
 *   - cipherflow-fhe/HEonGPU fork of HEonGPU 1.1 (README.md:135)   — the GPU path.
 * The reference holds no golden vectors / KATs for this path (its tests decrypt and compare
 * messages: unittests/test_gpu_bfv.cpp:332-335, unittests/test_gpu_ckks.cpp:37-43).  This file
 * restates the published Lattigo-v4 algorithms (ring/ntt.go, ring/basis_extension.go,
 * rlwe/keyswitch.go, ckks/evaluator.go, bfv/evaluator.go) from their public description; what pins
 * it is (i) big-integer / schoolbook identities checked in tests/test_oracle_math.py and (ii) the
 * reference's own message-level assertion (decrypt(op(enc)) == plain op) in tests/test_oracle_scheme.py.
 *
 * Conventions fixed here:
 *  - modulus index space of a context: [0,nq) = Q chain, [nq,nq+np) = special primes P,
 *    [nq+np, nq+np+nmul) = BFV auxiliary "QMul" basis.
 *  - forward NTT: in-place Cooley-Tukey, natural-order input -> bit-reversed output,
 *    twiddle psi^{brv(m+i)}, psi = g^((q-1)/2N), g = smallest primitive root of q (Lattigo ring.go).
 *  - inverse NTT: Gentleman-Sande, bit-reversed -> natural, scaled by N^-1.
 *  - all residues canonical in [0,q).
 */
#ifndef LS_ORACLE_H
#define LS_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORA_MAX_MOD 96

typedef struct ora_ctx {
    int n, logn;
    int nq, np, nmul, nmod;
    uint64_t t; /* BFV plaintext modulus, 0 for CKKS */
    uint64_t mod[ORA_MAX_MOD];
    uint64_t psi_root[ORA_MAX_MOD]; /* psi (2N-th primitive root) per modulus */
    uint64_t* psi[ORA_MAX_MOD];     /* psi[i][x]    = psi^{brv(x)}  mod q_i, x in [0,N) */
    uint64_t* psiinv[ORA_MAX_MOD];  /* psiinv[i][x] = psi^{-brv(x)} mod q_i */
    uint64_t ninv[ORA_MAX_MOD];
} ora_ctx;

/* number theory helpers (exported for tests) */
int ora_is_prime(uint64_t q);
uint64_t ora_primitive_root(uint64_t q);         /* smallest generator of Z_q^* */
uint64_t ora_psi(uint64_t q, int n);             /* g^((q-1)/2n) */
uint64_t ora_mulmod(uint64_t a, uint64_t b, uint64_t q);
uint64_t ora_powmod(uint64_t a, uint64_t e, uint64_t q);
/* 61-bit NTT primes descending from 2^61, == 1 mod 2n, skipping those in avoid[] */
int ora_gen_aux_primes(int n, int count, const uint64_t* avoid, int navoid, uint64_t* out);
int ora_bfv_aux_count(const uint64_t* q, int nlimbs, int logn);

ora_ctx* ora_ctx_new(int n, const uint64_t* q, int nq, const uint64_t* p, int np, uint64_t t);
void ora_ctx_free(ora_ctx* c);

/* single-limb transforms, in place, a[N] */
void ora_ntt(const ora_ctx* c, int mi, uint64_t* a);
void ora_intt(const ora_ctx* c, int mi, uint64_t* a);

/* limb-wise helpers: r may alias a or b */
void ora_vec_add(const ora_ctx* c, int mi, const uint64_t* a, const uint64_t* b, uint64_t* r);
void ora_vec_sub(const ora_ctx* c, int mi, const uint64_t* a, const uint64_t* b, uint64_t* r);
void ora_vec_neg(const ora_ctx* c, int mi, const uint64_t* a, uint64_t* r);
void ora_vec_mul(const ora_ctx* c, int mi, const uint64_t* a, const uint64_t* b, uint64_t* r);

/* Exact (float-corrected) RNS base conversion, coefficient domain.
 * src[i] = residues mod mod[sidx[i]], dst[j] receives residues mod mod[didx[j]].
 * centered=0: value reconstructed in [0,S); centered=1: in [-S/2, S/2) (Lattigo ModUpQtoP/PtoQ). */
void ora_baseconv(const ora_ctx* c, const int* sidx, int ns, const int* didx, int nd,
                  const uint64_t* const* src, uint64_t* const* dst, int centered);

/* Galois automorphism X -> X^g on one limb. */
void ora_automorph_ntt(const ora_ctx* c, uint64_t g, const uint64_t* a, uint64_t* r);          /* NTT domain: pure permutation */
void ora_automorph_coeff(const ora_ctx* c, int mi, uint64_t g, const uint64_t* a, uint64_t* r); /* coefficient domain: permutation + sign */

/* Hybrid key-switch (Lattigo GadgetProduct + ModDownQPtoQNTT) at level `lvl` (L=lvl+1 Q-limbs).
 * cx: [L][N] NTT domain.  key: compact ABI order [beta_k][2][klvl+1+np][N] (Q[0..klvl] then P), NTT, non-Montgomery.
 * out0,out1: [L][N] NTT domain. */
void ora_keyswitch(const ora_ctx* c, int lvl, const uint64_t* cx, const uint64_t* key, int klvl,
                   uint64_t* out0, uint64_t* out1);
/* its two halves (double-hoisted linear transforms keep sums over Q u P between them, Lattigo ckks/linear_transform.go
 * MultiplyByDiagMatrixBSGS): the gadget product alone, acc0 / acc1 = [L+np][N] over Q_lvl u P, NTT domain, not divided by P;
 * and the ModDown of one such polynomial (acc's P rows are left in the coefficient domain), out = [L][N]. */
void ora_gadget_product(const ora_ctx* c, int lvl, const uint64_t* cx, const uint64_t* key, int klvl,
                        uint64_t* acc0, uint64_t* acc1);
void ora_moddown(const ora_ctx* c, int lvl, uint64_t* acc, uint64_t* out);

/* CKKS (NTT domain ciphertexts, layout [poly][L][N]) */
void ora_ckks_mult(const ora_ctx* c, int lvl, const uint64_t* a, const uint64_t* b, uint64_t* d3);
void ora_ckks_relin(const ora_ctx* c, int lvl, const uint64_t* d3, const uint64_t* rlk, int klvl, uint64_t* out2);
void ora_ckks_rescale(const ora_ctx* c, int lvl, const uint64_t* in2, int npoly, uint64_t* out); /* out: [npoly][lvl][N] */
void ora_ckks_rotate(const ora_ctx* c, int lvl, const uint64_t* in2, uint64_t g, const uint64_t* glk, int klvl, uint64_t* out2);
void ora_ckks_mult_relin_rescale(const ora_ctx* c, int lvl, const uint64_t* a, const uint64_t* b,
                                 const uint64_t* rlk, int klvl, uint64_t* out /* [2][lvl][N] */);

/* BFV (coefficient domain ciphertexts) */
void ora_bfv_mult(const ora_ctx* c, int lvl, const uint64_t* a, const uint64_t* b, uint64_t* d3);
void ora_bfv_relin(const ora_ctx* c, int lvl, const uint64_t* d3, const uint64_t* rlk, int klvl, uint64_t* out2);
void ora_bfv_rotate(const ora_ctx* c, int lvl, const uint64_t* in2, uint64_t g, const uint64_t* glk, int klvl, uint64_t* out2);
void ora_bfv_rescale(const ora_ctx* c, int lvl, const uint64_t* in2, int npoly, uint64_t* out);
void ora_bfv_mult_relin(const ora_ctx* c, int lvl, const uint64_t* a, const uint64_t* b,
                        const uint64_t* rlk, int klvl, uint64_t* out2);

/* ring-t plaintext operands (one coefficient-domain limb): op 0 add, 1 sub, 2 mul */
void ora_lift_centered(const ora_ctx* c, int src_mi, int lvl, const uint64_t* pt, uint64_t* out);
void ora_bfv_scale_up(const ora_ctx* c, int lvl, const uint64_t* pt, uint64_t* out);
void ora_ckks_plain_ringt(const ora_ctx* c, int op, int lvl, int polys, const uint64_t* ct, const uint64_t* pt, uint64_t* out);
void ora_bfv_plain_ringt(const ora_ctx* c, int op, int lvl, int polys, const uint64_t* ct, const uint64_t* pt, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif
