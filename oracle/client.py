"""Test-only client-side crypto (keygen / encode / encrypt / decrypt / decode) on top of the CPU oracle.

TEST INFRASTRUCTURE ONLY.  The reference keeps this functionality inside Lattigo (fhe_ops_lib, SURVEY §2 #9,
out of scope for the GPU executor); it exists here solely so that the reference's own assertion
(decrypt(op(enc x)) == plain op, unittests/test_gpu_bfv.cpp:332-335, test_gpu_ckks.cpp:37-43) can be applied
to oracle and GPU results.  Key format follows plug-in/lattigo/acc/c_struct_import_export.go:41-135:
beta = ceil((level+1)/k) digits, each a degree-1 pair over Q[0..level] u P, NTT domain, non-Montgomery.
"""
import numpy as np

from .pyoracle import Oracle

GALOIS_GEN = 5  # frontend/custom_task.py:44


def galois_element_for_col_rotation(step, n):
    return pow(GALOIS_GEN, step % (n // 2), 2 * n)


def galois_element_for_row_rotation(n):
    return 2 * n - 1


class Client:
    def __init__(self, oracle: Oracle, seed=0, hamming=None):
        self.o = oracle
        self.n = oracle.n
        self.rng = np.random.default_rng(seed)
        n = self.n
        if hamming is None:
            s = self.rng.integers(-1, 2, size=n)
        else:
            s = np.zeros(n, dtype=np.int64)
            idx = self.rng.choice(n, size=hamming, replace=False)
            s[idx] = self.rng.choice([-1, 1], size=hamming)
        self.s = s.astype(np.int64)
        self.nqp = oracle.nq + oracle.np_
        # secret in NTT domain for every modulus of Q u P
        self.s_ntt = np.stack([oracle.ntt(i, self._lift(self.s, oracle.mod[i])) for i in range(self.nqp)])

    # ---- helpers
    @staticmethod
    def _lift(v, q):
        return np.mod(np.asarray(v, dtype=np.int64), np.int64(q)).astype(np.uint64)

    def _uniform(self, q):
        return self.rng.integers(0, q, size=self.n, dtype=np.uint64)

    def _gauss(self):
        return np.rint(self.rng.normal(0, 3.2, size=self.n)).astype(np.int64)

    def _mi(self, j, lvl):
        """limb j of a compact (Q[0..lvl] then P) polynomial -> oracle modulus index"""
        return j if j <= lvl else self.o.nq + (j - lvl - 1)

    def secret_ntt_permuted(self, g):
        """pi_g(s) in NTT domain for every modulus"""
        return np.stack([self.o.automorph_ntt(g, self.s_ntt[i]) for i in range(self.nqp)])

    # ---- evaluation keys (Lattigo genSwitchingKey): key_d = (-a*s_out + e + [limb in digit d] P*s_in, a)
    def gen_switching_key(self, s_in_ntt, s_out_ntt, klvl):
        o = self.o
        k = o.np_
        beta = (klvl + 1 + k - 1) // k
        ncomp = klvl + 1 + k
        key = np.zeros((beta, 2, ncomp, self.n), dtype=np.uint64)
        for d in range(beta):
            e = self._gauss()
            for j in range(ncomp):
                mi = self._mi(j, klvl)
                q = o.mod[mi]
                a = self._uniform(q)
                b = o.vec("neg", mi, o.vec("mul", mi, a, s_out_ntt[mi]))
                b = o.vec("add", mi, b, o.ntt(mi, self._lift(e, q)))
                if j <= klvl and d * k <= j < (d + 1) * k:
                    pmod = 1
                    for pp in o.p:
                        pmod = pmod * (pp % q) % q
                    ps = o.vec("mul", mi, s_in_ntt[mi], np.full(self.n, pmod, dtype=np.uint64))
                    b = o.vec("add", mi, b, ps)
                key[d, 0, j] = b
                key[d, 1, j] = a
        return key

    def gen_relin_key(self, klvl):
        s2 = np.stack([self.o.vec("mul", i, self.s_ntt[i], self.s_ntt[i]) for i in range(self.nqp)])
        return self.gen_switching_key(s2, self.s_ntt, klvl)

    def gen_galois_key(self, g, klvl):
        """Lattigo rotation key for element g: switches from s to pi_{g^-1}(s) (then the evaluator permutes by g)."""
        ginv = pow(g, -1, 2 * self.n)
        return self.gen_switching_key(self.s_ntt, self.secret_ntt_permuted(ginv), klvl)

    # ---- generic RLWE encryption of an integer plaintext polynomial given per-limb residues
    def _encrypt_limbs(self, m_limbs, lvl, ntt_domain):
        o = self.o
        L = lvl + 1
        ct = np.zeros((2, L, self.n), dtype=np.uint64)
        e = self._gauss()
        for i in range(L):
            q = o.mod[i]
            a = self._uniform(q)  # uniform in NTT domain == uniform in coefficient domain
            c0 = o.vec("neg", i, o.vec("mul", i, a, self.s_ntt[i]))
            c0 = o.vec("add", i, c0, o.ntt(i, self._lift(e, q)))
            if ntt_domain:
                c0 = o.vec("add", i, c0, o.ntt(i, m_limbs[i]))
                ct[0, i], ct[1, i] = c0, a
            else:
                c0 = o.vec("add", i, o.intt(i, c0), m_limbs[i])
                ct[0, i], ct[1, i] = c0, o.intt(i, a)
        return ct

    def _phase_bigint(self, ct, ntt_domain):
        """[c0 + c1*s (+ c2*s^2)]_Q as centred Python ints (CRT)."""
        o = self.o
        npoly, L, n = ct.shape
        limbs = []
        for i in range(L):
            acc = np.zeros(n, dtype=np.uint64)
            spow = None
            for k in range(npoly):
                ck = ct[k, i] if ntt_domain else o.ntt(i, ct[k, i])
                if k == 0:
                    term = ck
                else:
                    spow = self.s_ntt[i] if spow is None else o.vec("mul", i, spow, self.s_ntt[i])
                    term = o.vec("mul", i, ck, spow)
                acc = o.vec("add", i, acc, term)
            limbs.append(o.intt(i, acc))
        Q = 1
        for i in range(L):
            Q *= o.mod[i]
        res = [0] * n
        for i in range(L):
            qi = o.mod[i]
            Qi = Q // qi
            w = Qi * pow(Qi % qi, -1, qi)
            li = limbs[i].tolist()
            for x in range(n):
                res[x] += li[x] * w
        half = Q // 2
        out = []
        for x in range(n):
            v = res[x] % Q
            out.append(v - Q if v > half else v)
        return out, Q

    # ---- BFV (t prime, t == 1 mod 2N): Lattigo bfv encoder index matrix
    def _bfv_index(self):
        n = self.n
        logn = n.bit_length() - 1
        m = 2 * n
        idx = np.zeros(n, dtype=np.int64)
        pos = 1

        def brv(x):
            return int(format(x, "0%db" % logn)[::-1], 2)

        for i in range(n // 2):
            idx[i] = brv((pos - 1) >> 1)
            idx[i | (n >> 1)] = brv((m - pos - 1) >> 1)
            pos = pos * GALOIS_GEN % m
        return idx

    def _t_ctx(self):
        if not hasattr(self, "_tctx"):   # one assignment: custom executors call this from several runtime threads at once
            self._tctx = (Oracle(self.n, [self.o.t], [], 0), self._bfv_index())
        return self._tctx

    def bfv_encode(self, values):
        to, idx = self._t_ctx()
        v = np.zeros(self.n, dtype=np.uint64)
        vals = np.asarray(values, dtype=np.uint64) % np.uint64(self.o.t)
        v[idx[: len(vals)]] = vals
        return to.intt(0, v)  # coefficients mod t

    def bfv_decode(self, coeffs_mod_t):
        to, idx = self._t_ctx()
        ev = to.ntt(0, np.asarray(coeffs_mod_t, dtype=np.uint64))
        return ev[idx]

    def bfv_encrypt(self, values, lvl):
        """coefficient-domain BFV ciphertext of Delta*m, Delta = floor(Q_lvl/t)."""
        o = self.o
        m = self.bfv_encode(values).tolist()
        Q = 1
        for i in range(lvl + 1):
            Q *= o.mod[i]
        delta = Q // o.t
        limbs = [np.array([(delta % o.mod[i]) * mx % o.mod[i] for mx in m], dtype=np.uint64) for i in range(lvl + 1)]
        return self._encrypt_limbs(limbs, lvl, ntt_domain=False)

    def bfv_decrypt(self, ct):
        ph, Q = self._phase_bigint(np.asarray(ct), ntt_domain=False)
        t = self.o.t
        m = [((2 * t * v + Q) // (2 * Q)) % t for v in ph]
        return self.bfv_decode(np.array(m, dtype=np.uint64))

    # ---- CKKS canonical embedding (slot j <-> zeta^{5^j})
    def _ckks_slots_k(self):
        n = self.n
        k = np.zeros(n // 2, dtype=np.int64)
        pos = 1
        for j in range(n // 2):
            k[j] = (pos - 1) // 2
            pos = pos * GALOIS_GEN % (2 * n)
        return k

    def ckks_encode_coeffs(self, z, scale):
        n = self.n
        k = self._ckks_slots_k()
        allv = np.zeros(n, dtype=np.complex128)
        z = np.asarray(z, dtype=np.complex128)
        allv[k[: len(z)]] = z
        allv[(n - 1) - k[: len(z)]] = np.conj(z)  # exponent -(2k+1) == 2(n-1-k)+1 mod 2n
        zeta = np.exp(1j * np.pi * np.arange(n) / n)
        m = np.fft.fft(allv) / n * np.conj(zeta)
        return np.rint(m.real * scale).astype(object)

    def ckks_decode_coeffs(self, coeffs, scale, nslots=None):
        n = self.n
        k = self._ckks_slots_k()
        c = np.array([float(x) for x in coeffs], dtype=np.float64) / scale
        zeta = np.exp(1j * np.pi * np.arange(n) / n)
        allv = np.fft.ifft(c * zeta) * n
        z = allv[k]
        return z if nslots is None else z[:nslots]

    def ckks_encode_ringt(self, z, scale):
        """ring-t plaintext: the scaled message polynomial as ONE limb mod q_0 (centred representation)"""
        m = self.ckks_encode_coeffs(z, scale)
        return np.array([int(x) % self.o.mod[0] for x in m], dtype=np.uint64)

    def ckks_encode_ntt(self, z, lvl, scale):
        """full CKKS plaintext: [lvl+1][N] NTT-domain limbs"""
        m = self.ckks_encode_coeffs(z, scale)
        return np.stack([self.o.ntt(i, np.array([int(x) % self.o.mod[i] for x in m], dtype=np.uint64))
                         for i in range(lvl + 1)])

    def ckks_encrypt(self, z, lvl, scale):
        o = self.o
        m = self.ckks_encode_coeffs(z, scale)
        limbs = [np.array([int(x) % o.mod[i] for x in m], dtype=np.uint64) for i in range(lvl + 1)]
        return self._encrypt_limbs(limbs, lvl, ntt_domain=True)

    def ckks_decrypt(self, ct, scale, nslots=None):
        ph, _ = self._phase_bigint(np.asarray(ct), ntt_domain=True)
        return self.ckks_decode_coeffs(ph, scale, nslots)


def mean_precision_bits(want, got):
    """fhe_ops_lib/precision.cpp:103-186 metric: log2(1/mean|delta|) on real and imaginary parts."""
    d = np.asarray(want) - np.asarray(got)
    re = np.mean(np.abs(d.real)) + 1e-300
    im = np.mean(np.abs(d.imag)) + 1e-300
    return float(np.log2(1 / re)), float(np.log2(1 / im))
