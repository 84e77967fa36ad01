/*
 * ls_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY).  See ls_oracle.h for status and conventions.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include "ls_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ number theory */
u64 ora_mulmod(u64 a, u64 b, u64 q) { return (u64)((u128)a * b % q); }
u64 ora_powmod(u64 a, u64 e, u64 q) {
    u64 r = 1 % q;
    a %= q;
    while (e) {
        if (e & 1) r = ora_mulmod(r, a, q);
        a = ora_mulmod(a, a, q);
        e >>= 1;
    }
    return r;
}
static u64 invmod(u64 a, u64 q) { return ora_powmod(a, q - 2, q); } /* q prime */
static inline u64 addmod(u64 a, u64 b, u64 q) { u64 s = a + b; return s >= q ? s - q : s; }
static inline u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }

int ora_is_prime(u64 n) {
    if (n < 2) return 0;
    static const u64 sp[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (int i = 0; i < 12; i++) {
        if (n % sp[i] == 0) return n == sp[i];
    }
    u64 d = n - 1;
    int s = 0;
    while ((d & 1) == 0) { d >>= 1; s++; }
    for (int i = 0; i < 12; i++) { /* deterministic for n < 3.3e24 */
        u64 x = ora_powmod(sp[i], d, n);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int r = 1; r < s; r++) {
            x = ora_mulmod(x, x, n);
            if (x == n - 1) { comp = 0; break; }
        }
        if (comp) return 0;
    }
    return 1;
}

static u64 gcd64(u64 a, u64 b) { while (b) { u64 t = a % b; a = b; b = t; } return a; }

static u64 pollard_rho(u64 n) {
    if ((n & 1) == 0) return 2;
    for (u64 cst = 1;; cst++) {
        u64 x = 2, y = 2, d = 1;
        while (d == 1) {
            x = (ora_mulmod(x, x, n) + cst) % n;
            y = (ora_mulmod(y, y, n) + cst) % n;
            y = (ora_mulmod(y, y, n) + cst) % n;
            d = gcd64(x > y ? x - y : y - x, n);
        }
        if (d != n) return d;
    }
}

static void factor_rec(u64 n, u64* f, int* nf) {
    if (n == 1) return;
    if (ora_is_prime(n)) {
        for (int i = 0; i < *nf; i++) if (f[i] == n) return;
        f[(*nf)++] = n;
        return;
    }
    u64 d = pollard_rho(n);
    factor_rec(d, f, nf);
    factor_rec(n / d, f, nf);
}

/* Smallest primitive root (Lattigo ring.PrimitiveRoot: tries g = 2,3,... against the prime factors of q-1). */
u64 ora_primitive_root(u64 q) {
    u64 f[64];
    int nf = 0;
    u64 m = q - 1;
    for (u64 p = 2; p < 1000 && m > 1; p++) {
        if (m % p == 0) {
            f[nf++] = p;
            while (m % p == 0) m /= p;
        }
    }
    factor_rec(m, f, &nf);
    for (u64 g = 2;; g++) {
        int ok = 1;
        for (int i = 0; i < nf && ok; i++)
            if (ora_powmod(g, (q - 1) / f[i], q) == 1) ok = 0;
        if (ok) return g;
    }
}

u64 ora_psi(u64 q, int n) {
    u64 g = ora_primitive_root(q);
    return ora_powmod(g, (q - 1) / (2 * (u64)n), q);
}

/* bit length of prod q[0..k) via schoolbook multi-precision */
static int product_bitlen(const u64* q, int k) {
    u64 w[ORA_MAX_MOD + 2];
    int nw = 1;
    w[0] = 1;
    for (int i = 0; i < k; i++) {
        u64 carry = 0;
        for (int j = 0; j < nw; j++) {
            u128 t = (u128)w[j] * q[i] + carry;
            w[j] = (u64)t;
            carry = (u64)(t >> 64);
        }
        if (carry) w[nw++] = carry;
    }
    int bl = (nw - 1) * 64;
    u64 top = w[nw - 1];
    while (top) { bl++; top >>= 1; }
    return bl;
}

/* Lattigo bfv: nbQiMul = ceil((bitlen(Q) + logN) / 61) */
int ora_bfv_aux_count(const u64* q, int nlimbs, int logn) {
    int bl = product_bitlen(q, nlimbs) + logn;
    return (bl + 60) / 61;
}

/* Lattigo ring.GenerateNTTPrimesP(61, 2N, count): primes == 1 mod 2N, descending from 2^61 */
int ora_gen_aux_primes(int n, int count, const u64* avoid, int navoid, u64* out) {
    u64 step = 2 * (u64)n;
    u64 x = ((u64)1 << 61) + 1;
    int got = 0;
    while (got < count) {
        x -= step;
        if (x < step) return got;
        if (!ora_is_prime(x)) continue;
        int bad = 0;
        for (int i = 0; i < navoid; i++) if (avoid[i] == x) bad = 1;
        if (bad) continue;
        out[got++] = x;
    }
    return got;
}

/* ------------------------------------------------------------------ context */
static unsigned brv(unsigned x, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

ora_ctx* ora_ctx_new(int n, const u64* q, int nq, const u64* p, int np, u64 t) {
    ora_ctx* c = (ora_ctx*)calloc(1, sizeof(ora_ctx));
    c->n = n;
    c->logn = 0;
    while ((1 << c->logn) < n) c->logn++;
    c->nq = nq;
    c->np = np;
    c->t = t;
    for (int i = 0; i < nq; i++) c->mod[i] = q[i];
    for (int i = 0; i < np; i++) c->mod[nq + i] = p[i];
    c->nmul = 0;
    if (t) {
        c->nmul = ora_bfv_aux_count(q, nq, c->logn);
        int got = ora_gen_aux_primes(n, c->nmul, c->mod, nq + np, c->mod + nq + np);
        if (got != c->nmul) { free(c); return NULL; }
    }
    c->nmod = nq + np + c->nmul;
    for (int i = 0; i < c->nmod; i++) {
        u64 qi = c->mod[i];
        u64 psi = ora_psi(qi, n);
        u64 psii = invmod(psi, qi);
        c->psi_root[i] = psi;
        c->psi[i] = (u64*)malloc(sizeof(u64) * n);
        c->psiinv[i] = (u64*)malloc(sizeof(u64) * n);
        u64 pw = 1, pwi = 1;
        for (int j = 0; j < n; j++) {
            unsigned x = brv((unsigned)j, c->logn);
            c->psi[i][x] = pw;
            c->psiinv[i][x] = pwi;
            pw = ora_mulmod(pw, psi, qi);
            pwi = ora_mulmod(pwi, psii, qi);
        }
        c->ninv[i] = invmod((u64)n % qi, qi);
    }
    return c;
}

void ora_ctx_free(ora_ctx* c) {
    if (!c) return;
    for (int i = 0; i < c->nmod; i++) { free(c->psi[i]); free(c->psiinv[i]); }
    free(c);
}

/* ------------------------------------------------------------------ NTT (Lattigo ring/ntt.go NTT / InvNTT) */
void ora_ntt(const ora_ctx* c, int mi, u64* a) {
    const u64 q = c->mod[mi];
    const u64* psi = c->psi[mi];
    int n = c->n;
    int t = n;
    for (int m = 1; m < n; m <<= 1) {
        t >>= 1;
        for (int i = 0; i < m; i++) {
            int j1 = 2 * i * t;
            u64 F = psi[m + i];
            for (int j = j1; j < j1 + t; j++) {
                u64 U = a[j];
                u64 V = ora_mulmod(a[j + t], F, q);
                a[j] = addmod(U, V, q);
                a[j + t] = submod(U, V, q);
            }
        }
    }
}

void ora_intt(const ora_ctx* c, int mi, u64* a) {
    const u64 q = c->mod[mi];
    const u64* psii = c->psiinv[mi];
    int n = c->n;
    int t = 1;
    for (int m = n; m > 1; m >>= 1) {
        int h = m >> 1;
        int j1 = 0;
        for (int i = 0; i < h; i++) {
            u64 F = psii[h + i];
            for (int j = j1; j < j1 + t; j++) {
                u64 U = a[j], V = a[j + t];
                a[j] = addmod(U, V, q);
                a[j + t] = ora_mulmod(submod(U, V, q), F, q);
            }
            j1 += 2 * t;
        }
        t <<= 1;
    }
    u64 ni = c->ninv[mi];
    for (int j = 0; j < n; j++) a[j] = ora_mulmod(a[j], ni, q);
}

/* ------------------------------------------------------------------ limb-wise */
void ora_vec_add(const ora_ctx* c, int mi, const u64* a, const u64* b, u64* r) {
    u64 q = c->mod[mi];
    for (int j = 0; j < c->n; j++) r[j] = addmod(a[j], b[j], q);
}
void ora_vec_sub(const ora_ctx* c, int mi, const u64* a, const u64* b, u64* r) {
    u64 q = c->mod[mi];
    for (int j = 0; j < c->n; j++) r[j] = submod(a[j], b[j], q);
}
void ora_vec_neg(const ora_ctx* c, int mi, const u64* a, u64* r) {
    u64 q = c->mod[mi];
    for (int j = 0; j < c->n; j++) r[j] = a[j] ? q - a[j] : 0;
}
void ora_vec_mul(const ora_ctx* c, int mi, const u64* a, const u64* b, u64* r) {
    u64 q = c->mod[mi];
    for (int j = 0; j < c->n; j++) r[j] = ora_mulmod(a[j], b[j], q);
}

/* ------------------------------------------------------------------ base conversion
 * Lattigo ring/basis_extension.go modUpExact / reconstructRNS / multSum:
 *   y_i = x_i * (S/q_i)^-1 mod q_i ;  v = floor( sum_i double(y_i)/double(q_i) )  (sequential float64 adds)
 *   out_j = ( sum_i y_i * (S/q_i) - v*S ) mod p_j
 * centered (ModUpQtoP / ModUpPtoQ): add floor(S/2) before, subtract after. */
void ora_baseconv(const ora_ctx* c, const int* sidx, int ns, const int* didx, int nd,
                  const u64* const* src, u64* const* dst, int centered) {
    int n = c->n;
    u64 qs[ORA_MAX_MOD], shat_inv[ORA_MAX_MOD], half_s[ORA_MAX_MOD];
    for (int i = 0; i < ns; i++) qs[i] = c->mod[sidx[i]];
    for (int i = 0; i < ns; i++) {
        u64 pr = 1 % qs[i];
        for (int l = 0; l < ns; l++) if (l != i) pr = ora_mulmod(pr, qs[l] % qs[i], qs[i]);
        shat_inv[i] = invmod(pr, qs[i]);
        /* floor(S/2) mod q_i: S == 0 mod q_i and S odd => (S-1)/2 == (q_i-1)/2 * ... use (0-1)*inv2 */
        u64 inv2 = (qs[i] + 1) >> 1;
        half_s[i] = ora_mulmod(qs[i] - 1, inv2, qs[i]);
    }
    /* per target constants */
    u64* shat = (u64*)malloc(sizeof(u64) * ns * nd);
    u64* smod = (u64*)malloc(sizeof(u64) * nd);
    u64* half_d = (u64*)malloc(sizeof(u64) * nd);
    for (int j = 0; j < nd; j++) {
        u64 pj = c->mod[didx[j]];
        u64 all = 1 % pj;
        for (int l = 0; l < ns; l++) all = ora_mulmod(all, qs[l] % pj, pj);
        smod[j] = all;
        for (int i = 0; i < ns; i++) {
            u64 pr = 1 % pj;
            for (int l = 0; l < ns; l++) if (l != i) pr = ora_mulmod(pr, qs[l] % pj, pj);
            shat[j * ns + i] = pr;
        }
        u64 inv2 = (pj + 1) >> 1;
        half_d[j] = ora_mulmod(submod(all, 1 % pj, pj), inv2, pj);
    }
    u64 y[ORA_MAX_MOD];
    for (int x = 0; x < n; x++) {
        double vf = 0.0;
        for (int i = 0; i < ns; i++) {
            u64 xi = src[i][x];
            if (centered) xi = addmod(xi, half_s[i], qs[i]);
            y[i] = ora_mulmod(xi, shat_inv[i], qs[i]);
            vf += (double)y[i] / (double)qs[i];
        }
        u64 v = (u64)vf;
        for (int j = 0; j < nd; j++) {
            u64 pj = c->mod[didx[j]];
            u128 acc = 0;
            for (int i = 0; i < ns; i++) acc += (u128)(y[i] % pj) * shat[j * ns + i] % pj;
            u64 r = (u64)(acc % pj);
            r = submod(r, ora_mulmod(v % pj, smod[j], pj), pj);
            if (centered) r = submod(r, half_d[j], pj);
            dst[j][x] = r;
        }
    }
    free(shat);
    free(smod);
    free(half_d);
}

/* ------------------------------------------------------------------ automorphisms
 * NTT domain (Lattigo ring.PermuteNTTIndex): out[i] = in[ brv( ((g*(2*brv(i)+1) mod 2N) - 1)/2 ) ] */
void ora_automorph_ntt(const ora_ctx* c, u64 g, const u64* a, u64* r) {
    int n = c->n;
    u64 mask = 2 * (u64)n - 1;
    for (int i = 0; i < n; i++) {
        u64 e = 2 * (u64)brv((unsigned)i, c->logn) + 1;
        u64 e2 = (g * e) & mask;
        unsigned src = brv((unsigned)((e2 - 1) >> 1), c->logn);
        r[i] = a[src];
    }
}
/* coefficient domain (Lattigo ring.Permute): out[(i*g mod 2N) mod N] = +-in[i] */
void ora_automorph_coeff(const ora_ctx* c, int mi, u64 g, const u64* a, u64* r) {
    int n = c->n;
    u64 q = c->mod[mi];
    u64 mask = 2 * (u64)n - 1;
    for (int i = 0; i < n; i++) {
        u64 idx = ((u64)i * g) & mask;
        u64 v = a[i];
        if (idx >= (u64)n) { idx -= n; v = v ? q - v : 0; }
        r[idx] = v;
    }
}

/* ------------------------------------------------------------------ hybrid key-switch
 * Lattigo rlwe GadgetProduct (DecomposeSingleNTT per digit, MAC with the gadget ciphertexts)
 * followed by BasisExtender.ModDownQPtoQNTT.  digit d = Q-limbs [d*np, min((d+1)*np, L)). */
/* the gadget product alone: acc0, acc1 = [L + np][n] over Q_lvl u P, NTT domain, NOT divided by P (Lattigo
 * rlwe GadgetProductNoModDown; the operand of the double-hoisted linear transform, ckks MultiplyByDiagMatrixBSGS) */
void ora_gadget_product(const ora_ctx* c, int lvl, const u64* cx, const u64* key, int klvl, u64* acc0, u64* acc1) {
    int n = c->n, L = lvl + 1, np = c->np, nq = c->nq;
    int beta = (L + np - 1) / np;
    int kcomp = klvl + 1 + np; /* limbs per key polynomial */
    int T = L + np;            /* target limbs */
    u64* cxi = (u64*)malloc(sizeof(u64) * L * n);
    memcpy(cxi, cx, sizeof(u64) * L * n);
    for (int i = 0; i < L; i++) ora_intt(c, i, cxi + (size_t)i * n);
    memset(acc0, 0, sizeof(u64) * (size_t)T * n);
    memset(acc1, 0, sizeof(u64) * (size_t)T * n);
    u64* ext = (u64*)malloc(sizeof(u64) * n);
    for (int d = 0; d < beta; d++) {
        int d0 = d * np, d1 = d0 + np < L ? d0 + np : L;
        int sidx[ORA_MAX_MOD];
        const u64* srcp[ORA_MAX_MOD];
        for (int i = d0; i < d1; i++) { sidx[i - d0] = i; srcp[i - d0] = cxi + (size_t)i * n; }
        for (int tl = 0; tl < T; tl++) {
            int mi = tl < L ? tl : nq + (tl - L);
            if (tl >= d0 && tl < d1) {
                memcpy(ext, cx + (size_t)tl * n, sizeof(u64) * n);
            } else {
                u64* dstp[1] = {ext};
                int didx[1] = {mi};
                ora_baseconv(c, sidx, d1 - d0, didx, 1, srcp, dstp, 0);
                ora_ntt(c, mi, ext);
            }
            int kj = tl < L ? tl : klvl + 1 + (tl - L);
            const u64* k0 = key + ((size_t)(d * 2 + 0) * kcomp + kj) * n;
            const u64* k1 = key + ((size_t)(d * 2 + 1) * kcomp + kj) * n;
            u64 q = c->mod[mi];
            u64* a0 = acc0 + (size_t)tl * n;
            u64* a1 = acc1 + (size_t)tl * n;
            for (int x = 0; x < n; x++) {
                a0[x] = addmod(a0[x], ora_mulmod(ext[x], k0[x], q), q);
                a1[x] = addmod(a1[x], ora_mulmod(ext[x], k1[x], q), q);
            }
        }
    }
    free(ext); free(cxi);
}

/* ModDown of ONE polynomial over Q_lvl u P (NTT domain, [L + np][n]; its P rows are left in the coefficient domain):
 * out = (accQ - NTT(ModUpPtoQ_centered(INTT(accP)))) * P^-1   (Lattigo BasisExtender.ModDownQPtoQNTT) */
void ora_moddown(const ora_ctx* c, int lvl, u64* acc, u64* out) {
    int n = c->n, L = lvl + 1, np = c->np, nq = c->nq;
    int pidx[ORA_MAX_MOD], qidx[ORA_MAX_MOD];
    for (int i = 0; i < np; i++) pidx[i] = nq + i;
    for (int i = 0; i < L; i++) qidx[i] = i;
    u64* conv = (u64*)malloc(sizeof(u64) * L * n);
    const u64* srcp[ORA_MAX_MOD];
    u64* dstp[ORA_MAX_MOD];
    for (int i = 0; i < np; i++) {
        ora_intt(c, nq + i, acc + (size_t)(L + i) * n);
        srcp[i] = acc + (size_t)(L + i) * n;
    }
    for (int i = 0; i < L; i++) dstp[i] = conv + (size_t)i * n;
    ora_baseconv(c, pidx, np, qidx, L, srcp, dstp, 1);
    for (int i = 0; i < L; i++) {
        u64 q = c->mod[i];
        ora_ntt(c, i, conv + (size_t)i * n);
        u64 pinv = 1;
        for (int l = 0; l < np; l++) pinv = ora_mulmod(pinv, c->mod[nq + l] % q, q);
        pinv = invmod(pinv, q);
        for (int x = 0; x < n; x++)
            out[(size_t)i * n + x] = ora_mulmod(submod(acc[(size_t)i * n + x], conv[(size_t)i * n + x], q), pinv, q);
    }
    free(conv);
}

void ora_keyswitch(const ora_ctx* c, int lvl, const u64* cx, const u64* key, int klvl,
                   u64* out0, u64* out1) {
    int T = lvl + 1 + c->np;
    u64* acc0 = (u64*)malloc(sizeof(u64) * (size_t)T * c->n);
    u64* acc1 = (u64*)malloc(sizeof(u64) * (size_t)T * c->n);
    ora_gadget_product(c, lvl, cx, key, klvl, acc0, acc1);
    ora_moddown(c, lvl, acc0, out0);
    ora_moddown(c, lvl, acc1, out1);
    free(acc0); free(acc1);
}

/* ------------------------------------------------------------------ CKKS */
void ora_ckks_mult(const ora_ctx* c, int lvl, const u64* a, const u64* b, u64* d3) {
    int n = c->n, L = lvl + 1;
    for (int i = 0; i < L; i++) {
        u64 q = c->mod[i];
        const u64 *a0 = a + (size_t)i * n, *a1 = a + (size_t)(L + i) * n;
        const u64 *b0 = b + (size_t)i * n, *b1 = b + (size_t)(L + i) * n;
        u64 *d0 = d3 + (size_t)i * n, *d1 = d3 + (size_t)(L + i) * n, *d2 = d3 + (size_t)(2 * L + i) * n;
        for (int x = 0; x < n; x++) {
            u64 t0 = ora_mulmod(a0[x], b0[x], q);
            u64 t1 = addmod(ora_mulmod(a0[x], b1[x], q), ora_mulmod(a1[x], b0[x], q), q);
            u64 t2 = ora_mulmod(a1[x], b1[x], q);
            d0[x] = t0; d1[x] = t1; d2[x] = t2;
        }
    }
}

void ora_ckks_relin(const ora_ctx* c, int lvl, const u64* d3, const u64* rlk, int klvl, u64* out2) {
    int n = c->n, L = lvl + 1;
    size_t P = (size_t)L * n;
    u64* p0 = (u64*)malloc(sizeof(u64) * P);
    u64* p1 = (u64*)malloc(sizeof(u64) * P);
    ora_keyswitch(c, lvl, d3 + 2 * P, rlk, klvl, p0, p1);
    for (int i = 0; i < L; i++) {
        ora_vec_add(c, i, d3 + (size_t)i * n, p0 + (size_t)i * n, out2 + (size_t)i * n);
        ora_vec_add(c, i, d3 + P + (size_t)i * n, p1 + (size_t)i * n, out2 + P + (size_t)i * n);
    }
    free(p0); free(p1);
}

/* Lattigo ring.DivRoundByLastModulusNTTLvl: out_i = (x_i - NTT_i(([x_l + h]_{q_l} - h) mod q_i)) * q_l^-1, h=(q_l-1)/2 */
static void div_round_last(const ora_ctx* c, int lvl, const u64* in, u64* out, int ntt_domain) {
    int n = c->n;
    u64 ql = c->mod[lvl];
    u64 h = (ql - 1) >> 1;
    u64* last = (u64*)malloc(sizeof(u64) * n);
    u64* tmp = (u64*)malloc(sizeof(u64) * n);
    memcpy(last, in + (size_t)lvl * n, sizeof(u64) * n);
    if (ntt_domain) ora_intt(c, lvl, last);
    for (int x = 0; x < n; x++) last[x] = addmod(last[x], h, ql);
    for (int i = 0; i < lvl; i++) {
        u64 q = c->mod[i];
        u64 hq = h % q;
        u64 qlinv = invmod(ql % q, q);
        for (int x = 0; x < n; x++) tmp[x] = submod(last[x] % q, hq, q);
        if (ntt_domain) ora_ntt(c, i, tmp);
        for (int x = 0; x < n; x++)
            out[(size_t)i * n + x] = ora_mulmod(submod(in[(size_t)i * n + x], tmp[x], q), qlinv, q);
    }
    free(last); free(tmp);
}

void ora_ckks_rescale(const ora_ctx* c, int lvl, const u64* in2, int npoly, u64* out) {
    int n = c->n, L = lvl + 1;
    for (int p = 0; p < npoly; p++)
        div_round_last(c, lvl, in2 + (size_t)p * L * n, out + (size_t)p * lvl * n, 1);
}

/* Lattigo rlwe Evaluator.Automorphism: GadgetProduct(c1, key_g) ; +c0 ; then permute both by g */
void ora_ckks_rotate(const ora_ctx* c, int lvl, const u64* in2, u64 g, const u64* glk, int klvl, u64* out2) {
    int n = c->n, L = lvl + 1;
    size_t P = (size_t)L * n;
    u64* p0 = (u64*)malloc(sizeof(u64) * P);
    u64* p1 = (u64*)malloc(sizeof(u64) * P);
    ora_keyswitch(c, lvl, in2 + P, glk, klvl, p0, p1);
    for (int i = 0; i < L; i++) {
        ora_vec_add(c, i, p0 + (size_t)i * n, in2 + (size_t)i * n, p0 + (size_t)i * n);
        ora_automorph_ntt(c, g, p0 + (size_t)i * n, out2 + (size_t)i * n);
        ora_automorph_ntt(c, g, p1 + (size_t)i * n, out2 + P + (size_t)i * n);
    }
    free(p0); free(p1);
}

void ora_ckks_mult_relin_rescale(const ora_ctx* c, int lvl, const u64* a, const u64* b,
                                 const u64* rlk, int klvl, u64* out) {
    int n = c->n, L = lvl + 1;
    u64* d3 = (u64*)malloc(sizeof(u64) * 3 * L * n);
    u64* r2 = (u64*)malloc(sizeof(u64) * 2 * L * n);
    ora_ckks_mult(c, lvl, a, b, d3);
    ora_ckks_relin(c, lvl, d3, rlk, klvl, r2);
    ora_ckks_rescale(c, lvl, r2, 2, out);
    free(d3); free(r2);
}

/* ------------------------------------------------------------------ BFV (coefficient-domain ciphertexts)
 * Lattigo bfv evaluator tensorAndRescale: modUpAndNTT (Q -> QMul, centered), tensor in Q u QMul,
 * quantize: round(d/Q) via ModDownQPtoP, centered ModUpPtoQ back to Q, times t. */
void ora_bfv_mult(const ora_ctx* c, int lvl, const u64* a, const u64* b, u64* d3) {
    int n = c->n, L = lvl + 1;
    int M = ora_bfv_aux_count(c->mod, L, c->logn);
    int T = L + M;
    int qidx[ORA_MAX_MOD], midx[ORA_MAX_MOD], tidx[ORA_MAX_MOD];
    for (int i = 0; i < L; i++) { qidx[i] = i; tidx[i] = i; }
    for (int i = 0; i < M; i++) { midx[i] = c->nq + c->np + i; tidx[L + i] = midx[i]; }
    /* extended operands: [4][T][N] */
    u64* e = (u64*)malloc(sizeof(u64) * 4 * (size_t)T * n);
    for (int k = 0; k < 4; k++) {
        const u64* srcpoly = (k < 2 ? a : b) + (size_t)(k & 1) * L * n;
        u64* ep = e + (size_t)k * T * n;
        memcpy(ep, srcpoly, sizeof(u64) * L * n);
        const u64* srcp[ORA_MAX_MOD];
        u64* dstp[ORA_MAX_MOD];
        for (int i = 0; i < L; i++) srcp[i] = srcpoly + (size_t)i * n;
        for (int i = 0; i < M; i++) dstp[i] = ep + (size_t)(L + i) * n;
        ora_baseconv(c, qidx, L, midx, M, srcp, dstp, 1);
        for (int i = 0; i < T; i++) ora_ntt(c, tidx[i], ep + (size_t)i * n);
    }
    u64* d = (u64*)malloc(sizeof(u64) * 3 * (size_t)T * n);
    for (int i = 0; i < T; i++) {
        u64 q = c->mod[tidx[i]];
        const u64 *a0 = e + (size_t)i * n, *a1 = e + ((size_t)T + i) * n;
        const u64 *b0 = e + ((size_t)2 * T + i) * n, *b1 = e + ((size_t)3 * T + i) * n;
        u64 *d0 = d + (size_t)i * n, *d1 = d + ((size_t)T + i) * n, *d2 = d + ((size_t)2 * T + i) * n;
        for (int x = 0; x < n; x++) {
            d0[x] = ora_mulmod(a0[x], b0[x], q);
            d1[x] = addmod(ora_mulmod(a0[x], b1[x], q), ora_mulmod(a1[x], b0[x], q), q);
            d2[x] = ora_mulmod(a1[x], b1[x], q);
        }
        ora_intt(c, tidx[i], d0); ora_intt(c, tidx[i], d1); ora_intt(c, tidx[i], d2);
    }
    u64* ext = (u64*)malloc(sizeof(u64) * (size_t)M * n);
    for (int k = 0; k < 3; k++) {
        u64* dk = d + (size_t)k * T * n;
        const u64* srcp[ORA_MAX_MOD];
        u64* dstp[ORA_MAX_MOD];
        /* rounded division by Q into basis QMul */
        for (int i = 0; i < L; i++) srcp[i] = dk + (size_t)i * n;
        for (int i = 0; i < M; i++) dstp[i] = ext + (size_t)i * n;
        ora_baseconv(c, qidx, L, midx, M, srcp, dstp, 1);
        for (int i = 0; i < M; i++) {
            u64 p = c->mod[midx[i]];
            u64 qinv = 1;
            for (int l = 0; l < L; l++) qinv = ora_mulmod(qinv, c->mod[l] % p, p);
            qinv = invmod(qinv, p);
            u64* dm = dk + (size_t)(L + i) * n;
            for (int x = 0; x < n; x++) dm[x] = ora_mulmod(submod(dm[x], ext[(size_t)i * n + x], p), qinv, p);
        }
        /* centered extension QMul -> Q, times t */
        for (int i = 0; i < M; i++) srcp[i] = dk + (size_t)(L + i) * n;
        for (int i = 0; i < L; i++) dstp[i] = d3 + ((size_t)k * L + i) * n;
        ora_baseconv(c, midx, M, qidx, L, srcp, dstp, 1);
        for (int i = 0; i < L; i++) {
            u64 q = c->mod[i];
            u64 tq = c->t % q;
            u64* o = d3 + ((size_t)k * L + i) * n;
            for (int x = 0; x < n; x++) o[x] = ora_mulmod(o[x], tq, q);
        }
    }
    free(ext); free(d); free(e);
}

static void ks_coeff(const ora_ctx* c, int lvl, const u64* cx_coeff, const u64* key, int klvl, u64* p0, u64* p1) {
    int n = c->n, L = lvl + 1;
    size_t P = (size_t)L * n;
    u64* cx = (u64*)malloc(sizeof(u64) * P);
    memcpy(cx, cx_coeff, sizeof(u64) * P);
    for (int i = 0; i < L; i++) ora_ntt(c, i, cx + (size_t)i * n);
    ora_keyswitch(c, lvl, cx, key, klvl, p0, p1);
    for (int i = 0; i < L; i++) { ora_intt(c, i, p0 + (size_t)i * n); ora_intt(c, i, p1 + (size_t)i * n); }
    free(cx);
}

void ora_bfv_relin(const ora_ctx* c, int lvl, const u64* d3, const u64* rlk, int klvl, u64* out2) {
    int n = c->n, L = lvl + 1;
    size_t P = (size_t)L * n;
    u64* p0 = (u64*)malloc(sizeof(u64) * P);
    u64* p1 = (u64*)malloc(sizeof(u64) * P);
    ks_coeff(c, lvl, d3 + 2 * P, rlk, klvl, p0, p1);
    for (int i = 0; i < L; i++) {
        ora_vec_add(c, i, d3 + (size_t)i * n, p0 + (size_t)i * n, out2 + (size_t)i * n);
        ora_vec_add(c, i, d3 + P + (size_t)i * n, p1 + (size_t)i * n, out2 + P + (size_t)i * n);
    }
    free(p0); free(p1);
}

void ora_bfv_rotate(const ora_ctx* c, int lvl, const u64* in2, u64 g, const u64* glk, int klvl, u64* out2) {
    int n = c->n, L = lvl + 1;
    size_t P = (size_t)L * n;
    u64* p0 = (u64*)malloc(sizeof(u64) * P);
    u64* p1 = (u64*)malloc(sizeof(u64) * P);
    ks_coeff(c, lvl, in2 + P, glk, klvl, p0, p1);
    for (int i = 0; i < L; i++) {
        ora_vec_add(c, i, p0 + (size_t)i * n, in2 + (size_t)i * n, p0 + (size_t)i * n);
        ora_automorph_coeff(c, i, g, p0 + (size_t)i * n, out2 + (size_t)i * n);
        ora_automorph_coeff(c, i, g, p1 + (size_t)i * n, out2 + P + (size_t)i * n);
    }
    free(p0); free(p1);
}

void ora_bfv_rescale(const ora_ctx* c, int lvl, const u64* in2, int npoly, u64* out) {
    int n = c->n, L = lvl + 1;
    for (int p = 0; p < npoly; p++)
        div_round_last(c, lvl, in2 + (size_t)p * L * n, out + (size_t)p * lvl * n, 0);
}

void ora_bfv_mult_relin(const ora_ctx* c, int lvl, const u64* a, const u64* b,
                        const u64* rlk, int klvl, u64* out2) {
    int n = c->n, L = lvl + 1;
    u64* d3 = (u64*)malloc(sizeof(u64) * 3 * L * n);
    ora_bfv_mult(c, lvl, a, b, d3);
    ora_bfv_relin(c, lvl, d3, rlk, klvl, out2);
    free(d3);
}

/* ------------------------------------------------------------------ ring-t plaintexts
 * A ring-t plaintext is ONE coefficient-domain limb (abi: level 0, plug-in/lattigo/acc/c_struct_import_export.go:179-184):
 *   CKKS: the scaled message reduced mod q_0  -> lifted as a CENTRED integer to every q_i, then NTT;
 *   BFV : the message mod t.  multiply: used directly as residues mod q_i (Lattigo bfv mulPlaintextRingT);
 *         add/sub: scaled up by Q/t with rounding (Lattigo bfv scaleUp): u = (m*[Q]_t + t/2) mod t,
 *         out_i = (u - t/2) * (-t^-1) mod q_i  ==  round(m*Q/t) mod q_i. */
void ora_lift_centered(const ora_ctx* c, int src_mi, int lvl, const u64* pt, u64* out) {
    int n = c->n;
    u64 q0 = c->mod[src_mi], half = q0 >> 1;
    for (int i = 0; i <= lvl; i++) {
        u64 qi = c->mod[i];
        for (int x = 0; x < n; x++) {
            u64 v = pt[x];
            out[(size_t)i * n + x] = v > half ? submod(0, (q0 - v) % qi, qi) : v % qi;
        }
    }
}

void ora_bfv_scale_up(const ora_ctx* c, int lvl, const u64* pt, u64* out) {
    int n = c->n;
    u64 t = c->t, thalf = t >> 1;
    u64 qmodt = 1 % t;
    for (int i = 0; i <= lvl; i++) qmodt = ora_mulmod(qmodt, c->mod[i] % t, t);
    for (int i = 0; i <= lvl; i++) {
        u64 qi = c->mod[i];
        u64 neg_tinv = qi - invmod(t % qi, qi);
        u64 thalf_q = thalf % qi;
        for (int x = 0; x < n; x++) {
            u64 u = (ora_mulmod(pt[x] % t, qmodt, t) + thalf) % t;
            out[(size_t)i * n + x] = ora_mulmod(submod(u % qi, thalf_q, qi), neg_tinv, qi);
        }
    }
}

/* op: 0 add, 1 sub, 2 mul.  ct [polys][L][N] NTT domain, pt ring-t limb mod q_0 */
void ora_ckks_plain_ringt(const ora_ctx* c, int op, int lvl, int polys, const u64* ct, const u64* pt, u64* out) {
    int n = c->n, L = lvl + 1;
    u64* lift = (u64*)malloc(sizeof(u64) * L * n);
    ora_lift_centered(c, 0, lvl, pt, lift);
    for (int i = 0; i < L; i++) ora_ntt(c, i, lift + (size_t)i * n);
    memcpy(out, ct, sizeof(u64) * polys * L * n);
    for (int i = 0; i < L; i++) {
        if (op == 0) ora_vec_add(c, i, ct + (size_t)i * n, lift + (size_t)i * n, out + (size_t)i * n);
        else if (op == 1) ora_vec_sub(c, i, ct + (size_t)i * n, lift + (size_t)i * n, out + (size_t)i * n);
        else
            for (int p = 0; p < polys; p++)
                ora_vec_mul(c, i, ct + ((size_t)p * L + i) * n, lift + (size_t)i * n, out + ((size_t)p * L + i) * n);
    }
    free(lift);
}

/* op: 0 add, 1 sub, 2 mul.  ct [polys][L][N] coefficient domain, pt ring-t limb mod t */
void ora_bfv_plain_ringt(const ora_ctx* c, int op, int lvl, int polys, const u64* ct, const u64* pt, u64* out) {
    int n = c->n, L = lvl + 1;
    u64* tmp = (u64*)malloc(sizeof(u64) * L * n);
    memcpy(out, ct, sizeof(u64) * polys * L * n);
    if (op == 0 || op == 1) {
        ora_bfv_scale_up(c, lvl, pt, tmp);
        for (int i = 0; i < L; i++) {
            if (op == 0) ora_vec_add(c, i, ct + (size_t)i * n, tmp + (size_t)i * n, out + (size_t)i * n);
            else ora_vec_sub(c, i, ct + (size_t)i * n, tmp + (size_t)i * n, out + (size_t)i * n);
        }
    } else {
        for (int i = 0; i < L; i++) {
            u64 qi = c->mod[i];
            u64* pn = tmp + (size_t)i * n;
            for (int x = 0; x < n; x++) pn[x] = pt[x] % qi;
            ora_ntt(c, i, pn);
            for (int p = 0; p < polys; p++) {
                u64* o = out + ((size_t)p * L + i) * n;
                ora_ntt(c, i, o);
                ora_vec_mul(c, i, o, pn, o);
                ora_intt(c, i, o);
            }
        }
    }
    free(tmp);
}
