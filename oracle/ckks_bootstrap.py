"""CKKS bootstrapping on the CPU oracle (TEST INFRASTRUCTURE, like the rest of oracle/): the operator program that the
reference's `bootstrap` node stands for (reference: mega_ag_runners/gpu/mega_ag_executors_gpu.cu:410-426 hands the
ciphertext to HEonGPU's regular_bootstrapping_v2; parameters frontend/custom_task.py:383-468, rotation planner
frontend/bootstrap_params.py:104-263).  PARITY UNPINNED: HEonGPU / Lattigo are absent from /root/reference, so what is
restated here is the published algorithm (Cheon-Han-Kim-Kim-Song bootstrapping with the FFT-factorised CoeffsToSlots /
SlotsToCoeffs of Chen-Chillotti-Song and the cosine + double-angle EvalMod of Han-Ki), with the reference's parameter
structure: CtS depth / StC depth / K / double-angle count / message ratio, one prime per linear-transform level with the
plaintext diagonals encoded at that prime, EvalMod at scale ~ q.  What it pins is the reference's own assertion:
decrypt(bootstrap(ct)) == decrypt(ct) to >= 10 bits of mean precision (unittests/test_gpu_ckks.cpp:763-781,
unittests/test_cpu_ckks.cpp:37-43).

Every step is a composition of oracle operators that the GPU executor already reproduces bit for bit (rotate, conjugate,
ct x pt, ct x ct + relin + rescale, limb-wise add / constant ops, centred lift), so the same program replayed on the
device must give identical residues.

Math (n = N/2 slots, t_k = m_k + i*m_{k+n} the packed coefficients, z = U t the slots, U[j][k] = zeta_j^k):
  U   = L_n ... L_4 L_2 B      (radix-2 layers of the "special FFT", B = bit reversal)
  U^-1 = (1/n) B L'_2 ... L'_n
CoeffsToSlots applies (1/n) L'_2...L'_n (no B: the coefficients land in bit-reversed slot order, which the slot-wise
EvalMod does not care about), SlotsToCoeffs applies L_n...L_2 to that order.  Each layer has the three diagonals
{0, +len/2, -len/2}; consecutive layers are merged (`depth` matrices) and each merged matrix is applied with baby-step /
giant-step rotations.
"""
from collections import namedtuple

import numpy as np

from .client import galois_element_for_col_rotation, galois_element_for_row_rotation

Ct = namedtuple("Ct", "data level scale")   # data: [2][level+1][N] NTT-domain residues
# a ciphertext times P over Q_level u P (rows: q_0..q_level, then the special primes), NTT domain: what a key switch yields
# before its division by P.  Sums of such stay exact; ONE rounded division (moddown) per sum instead of one per term.
ExtCt = namedtuple("ExtCt", "data level scale")


# ------------------------------------------------------------------------------------------------ evaluator
class Evaluator:
    """Scale-tracking CKKS evaluator over the oracle's operators; keys come from the test client on demand."""

    def __init__(self, oracle, client, key_level):
        self.o, self.c, self.klvl = oracle, client, key_level
        self.n = oracle.n
        self.rlk = client.gen_relin_key(key_level)
        self.glk = {}
        self.counts = {"rotate": 0, "mult": 0, "mul_plain": 0}

    def q(self, level):
        return self.o.mod[level]

    def _key(self, g):
        if g not in self.glk:
            self.glk[g] = self.c.gen_galois_key(g, self.klvl)
        return self.glk[g]

    def _limbs(self, ct, fn):
        d = ct.data
        return np.stack([np.stack([fn(pl, j, d[pl, j]) for j in range(ct.level + 1)]) for pl in range(d.shape[0])])

    def add(self, a, b):
        assert a.level == b.level and abs(a.scale / b.scale - 1) < 1e-9, (a.level, b.level, a.scale, b.scale)
        return Ct(self._limbs(a, lambda pl, j, x: self.o.vec("add", j, x, b.data[pl, j])), a.level, a.scale)

    def sub(self, a, b):
        assert a.level == b.level and abs(a.scale / b.scale - 1) < 1e-9
        return Ct(self._limbs(a, lambda pl, j, x: self.o.vec("sub", j, x, b.data[pl, j])), a.level, a.scale)

    def drop(self, a, level):
        assert level <= a.level
        return Ct(a.data[:, : level + 1].copy(), level, a.scale)

    def rescale(self, a):
        return Ct(self.o.ckks_rescale(a.level, a.data), a.level - 1, a.scale / self.q(a.level))

    def mul(self, a, b):
        """ct x ct + relinearise + rescale"""
        lvl = min(a.level, b.level)
        a, b = self.drop(a, lvl), self.drop(b, lvl)
        self.counts["mult"] += 1
        return Ct(self.o.ckks_mult_relin_rescale(lvl, a.data, b.data, self.rlk, self.klvl), lvl - 1,
                  a.scale * b.scale / self.q(lvl))

    def rotate(self, a, r):
        r %= self.n // 2
        if r == 0:
            return a
        g = galois_element_for_col_rotation(r, self.n)
        self.counts["rotate"] += 1
        return Ct(self.o.ckks_rotate(a.level, a.data, g, self._key(g), self.klvl), a.level, a.scale)

    # --- extended (Q u P) ciphertexts: the operands of the double-hoisted linear transform
    # (Lattigo v4 ckks/linear_transform.go MultiplyByDiagMatrixBSGS, rlwe GadgetProductNoModDown / ModDownQPtoQNTT)
    def _mi(self, level, tl):
        """modulus index of row tl of an extended polynomial at `level`"""
        return tl if tl <= level else self.o.nq + (tl - level - 1)

    def lift_ext(self, a):
        """(P c0, P c1) on the Q rows, zero on the P rows: the same ciphertext, times P"""
        L, T = a.level + 1, a.level + 1 + self.o.np_
        d = np.zeros((2, T, self.n), dtype=np.uint64)
        for j in range(L):
            pm = 1
            for pp in self.o.p:
                pm = pm * (pp % self.q(j)) % self.q(j)
            for pl in range(2):
                d[pl, j] = self.o.vec("mul", j, a.data[pl, j], self._const(pm, j))
        return ExtCt(d, a.level, a.scale)

    def rotate_ext(self, a, r):
        """rotation WITHOUT the division by P: automorphism of (P c0 + ks0, ks1), ks = gadget product of c1 with the key"""
        r %= self.n // 2
        if r == 0:
            return self.lift_ext(a)
        g = galois_element_for_col_rotation(r, self.n)
        self.counts["rotate"] += 1
        L = a.level + 1
        acc = self.o.gadget_product(a.level, a.data[1], self._key(g), self.klvl)
        c0p = self.lift_ext(a).data[0]
        for j in range(L):
            acc[0, j] = self.o.vec("add", j, acc[0, j], c0p[j])
        out = np.stack([np.stack([self.o.automorph_ntt(g, acc[pl, tl]) for tl in range(acc.shape[1])]) for pl in range(2)])
        return ExtCt(out, a.level, a.scale)

    def mul_plain_ext(self, a, pt, pt_scale):
        """extended ciphertext times an NTT-domain plaintext over the same moduli [level+1+np][N]"""
        self.counts["mul_plain"] += 1
        T = a.data.shape[1]
        d = np.stack([np.stack([self.o.vec("mul", self._mi(a.level, tl), a.data[pl, tl], pt[tl]) for tl in range(T)]) for pl in range(2)])
        return ExtCt(d, a.level, a.scale * pt_scale)

    def add_ext(self, a, b):
        assert a.level == b.level and abs(a.scale / b.scale - 1) < 1e-9
        T = a.data.shape[1]
        d = np.stack([np.stack([self.o.vec("add", self._mi(a.level, tl), a.data[pl, tl], b.data[pl, tl]) for tl in range(T)]) for pl in range(2)])
        return ExtCt(d, a.level, a.scale)

    def moddown(self, a):
        """the rounded division by P that ends a key switch, on a sum of extended ciphertexts"""
        return Ct(self.o.moddown(a.level, a.data), a.level, a.scale)

    def encode_ext(self, z, level, scale):
        """plaintext over Q_level u P: the same integer polynomial reduced at the special primes too"""
        m = self.c.ckks_encode_coeffs(z, scale)
        T = level + 1 + self.o.np_
        return np.stack([self.o.ntt(self._mi(level, tl), np.array([int(x) % self.o.mod[self._mi(level, tl)] for x in m], dtype=np.uint64))
                         for tl in range(T)])

    def conj(self, a):
        g = galois_element_for_row_rotation(self.n)
        self.counts["rotate"] += 1
        return Ct(self.o.ckks_rotate(a.level, a.data, g, self._key(g), self.klvl), a.level, a.scale)

    def mul_plain(self, a, pt, pt_scale):
        """every polynomial times an NTT-domain plaintext [level+1][N]; no rescale"""
        self.counts["mul_plain"] += 1
        return Ct(self._limbs(a, lambda pl, j, x: self.o.vec("mul", j, x, pt[j])), a.level, a.scale * pt_scale)

    def _const(self, value, j):
        return np.full(self.n, int(value) % self.q(j), dtype=np.uint64)

    def mul_int(self, a, k):
        """times an integer: exact, scale unchanged"""
        return Ct(self._limbs(a, lambda pl, j, x: self.o.vec("mul", j, x, self._const(k, j))), a.level, a.scale)

    def mul_const(self, a, c, const_scale):
        """times the real constant c encoded as round(c * const_scale); no rescale"""
        assert abs(c * const_scale) < 4.6e18, "constant out of range: the modulus chain does not match the level plan"
        k = int(round(c * const_scale))
        return Ct(self._limbs(a, lambda pl, j, x: self.o.vec("mul", j, x, self._const(k, j))), a.level, a.scale * const_scale)

    def add_const(self, a, c):
        """plus the real constant c in every slot (a constant polynomial is constant in the NTT domain too)"""
        k = int(round(c * a.scale))
        d = a.data.copy()
        for j in range(a.level + 1):
            d[0, j] = self.o.vec("add", j, d[0, j], self._const(k, j))
        return Ct(d, a.level, a.scale)

    def mul_by_i(self, a, sign=1):
        """times +-i: the monomial +-X^(N/2), exact"""
        mono = np.zeros(self.n, dtype=np.int64)
        mono[self.n // 2] = sign
        pts = [self.o.ntt(j, np.mod(mono, np.int64(self.q(j))).astype(np.uint64)) for j in range(a.level + 1)]
        return Ct(self._limbs(a, lambda pl, j, x: self.o.vec("mul", j, x, pts[j])), a.level, a.scale)

    def encode(self, z, level, scale):
        return self.c.ckks_encode_ntt(z, level, scale)


# ------------------------------------------------------------------------------------------------ DFT layers
def _rot_group(n_slots, m):
    g, out = 1, []
    for _ in range(n_slots):
        out.append(g)
        g = g * 5 % m
    return out


def layer_diagonals(n_slots, length, inverse):
    """The three diagonals of one radix-2 layer on n_slots points (diag k: d[t] multiplies x[t+k])."""
    m = 4 * n_slots                      # = 2N
    rg = _rot_group(n_slots, m)
    lenh, lenq = length // 2, length * 4
    d0 = np.zeros(n_slots, dtype=np.complex128)
    dp = np.zeros(n_slots, dtype=np.complex128)
    dm = np.zeros(n_slots, dtype=np.complex128)
    for t in range(n_slots):
        j = t % length
        if j < lenh:
            w = np.exp(2j * np.pi * ((rg[j] % lenq) * (m // lenq)) / m)
            d0[t] = 1
            dp[t] = 1 if inverse else w           # inverse: a + b ; forward: u + w*v
        else:
            jj = j - lenh
            if inverse:
                w = np.exp(2j * np.pi * ((lenq - (rg[jj] % lenq)) * (m // lenq)) / m)
                d0[t] = -w                        # (a - b) * w
                dm[t] = w
            else:
                w = np.exp(2j * np.pi * ((rg[jj] % lenq) * (m // lenq)) / m)
                d0[t] = -w                        # u - w*v
                dm[t] = 1
    if lenh % n_slots == (-lenh) % n_slots:          # the widest layer: +n/2 and -n/2 are the same rotation
        return {0: d0, lenh % n_slots: dp + dm}
    return {0: d0, lenh % n_slots: dp, (-lenh) % n_slots: dm}


def compose(first, second, n_slots):
    """diagonals of (second o first): apply `first`, then `second`"""
    out = {}
    for i, a in second.items():           # (A B) diag i+j : A_i[t] * B_j[t+i]
        for j, b in first.items():
            k = (i + j) % n_slots
            out[k] = out.get(k, 0) + a * np.roll(b, -i)
    return {k: v for k, v in out.items() if np.max(np.abs(v)) > 1e-300}


def merged_matrices(n_slots, depth, inverse):
    """`depth` merged layer groups in application order (CoeffsToSlots: inverse layers len = n..2; SlotsToCoeffs: forward
    layers len = 2..n), grouped like frontend/bootstrap_params.py:104-119 (ceil(remaining / groups left))."""
    log_n = n_slots.bit_length() - 1
    lengths = [1 << l for l in range(log_n, 0, -1)] if inverse else [1 << l for l in range(1, log_n + 1)]
    sizes, left = [], log_n
    for i in range(depth):
        s = -(-left // (depth - i))
        sizes.append(s)
        left -= s
    if not inverse:
        sizes = sizes[::-1]
    mats, pos = [], 0
    for s in sizes:
        m = None
        for length in lengths[pos: pos + s]:
            lay = layer_diagonals(n_slots, length, inverse)
            m = lay if m is None else compose(m, lay, n_slots)
        mats.append(m)
        pos += s
    return mats


def apply_plain(diags, x):
    """numpy model of a diagonal-form matrix"""
    return sum(d * np.roll(x, -k) for k, d in diags.items())


def bsgs_sets(ks, n_slots, n1):
    """giant steps (multiples of n1) and baby steps (< n1) of a diagonal index set"""
    giants = sorted({((k % n_slots) // n1) * n1 % n_slots for k in ks})
    babies = sorted({(k % n_slots) % n1 for k in ks})
    return giants, babies


def bsgs_split(ks, n_slots, ratio=2.0):
    """The baby-step count the reference's planner picks (frontend/bootstrap_params.py:193-207: the first power of two at
    which #babies / #giants reaches `ratio`, the previous one if it overshoots) -- the caller's Galois keys exist for
    exactly the rotations this choice needs."""
    n1 = 1
    while n1 < n_slots:
        giants, babies = bsgs_sets(ks, n_slots, n1)
        nb_g, nb_b = len(giants) - 1, len(babies) - 1
        if nb_g == 0 or nb_b / nb_g == ratio:
            return n1
        if nb_b / nb_g > ratio:
            return n1 // 2
        n1 <<= 1
    return 1


def rotations_of(diags, n_slots, ratio=2.0):
    """non-zero rotations one merged matrix needs (frontend/bootstrap_params.py:210-230)"""
    ks = sorted(diags)
    if len(ks) < 3:
        return sorted(k for k in ks if k)
    giants, babies = bsgs_sets(ks, n_slots, bsgs_split(ks, n_slots, ratio))
    return sorted({r for r in giants + babies if r})


def linear_transform(ev, ct, diags, ratio=2.0, plains=None, rescale=True, n_slots=None, double_hoist=True):
    """ct <- M ct for M in diagonal form; the diagonals are encoded at the scale of the ciphertext's top prime, so the single
    rescale at the end leaves the scale unchanged.  Consumes one level.  Fewer than three diagonals: one rotation each;
    otherwise baby-step / giant-step with the planner's split:
        M x = sum_g rot_g( sum_b rot_{-g}(d_{g+b}) . rot_b(x) ).
    double_hoist (Lattigo's MultiplyByDiagMatrixBSGS): the baby-step rotations stay over Q u P (no division by P), the
    plaintexts are encoded over Q u P too, each giant step's inner sum is divided by P once, rotated without division, and the
    sum over the giant steps is divided once more: (giant steps + 1) rounded divisions instead of one per rotation."""
    # `plains`: {k: NTT-domain plaintext [level+1 (+np)][N]} replaces this module's own encoding of (rot_{-g} of) diagonal k --
    # the parity tests pass the device library's plaintexts so that both sides multiply by the same integers
    # n_slots: period of the diagonals (sparse packing: shorter than N/2; they are tiled over the N/2 slots when encoded)
    n = ev.n // 2
    period = n_slots or n
    tile = (lambda v: np.tile(v, n // len(v)))
    ks = sorted(diags)
    pt_scale = float(ev.q(ct.level))
    acc = None
    if len(ks) < 3:
        for k in ks:
            pt = plains[k] if plains is not None else ev.encode(tile(diags[k]), ct.level, pt_scale)
            term = ev.mul_plain(ev.rotate(ct, k), pt, pt_scale)
            acc = term if acc is None else ev.add(acc, term)
        return ev.rescale(acc) if rescale else acc
    n1 = bsgs_split(ks, period, ratio)
    babies = {}
    by_giant = {}
    for k in ks:
        by_giant.setdefault((k // n1) * n1, []).append(k)
    if double_hoist:
        for g, klist in sorted(by_giant.items()):
            inner = None
            for k in klist:
                b = k - g
                if b not in babies:
                    babies[b] = ev.rotate_ext(ct, b)
                pt = plains[k] if plains is not None else ev.encode_ext(tile(np.roll(diags[k], g)), ct.level, pt_scale)
                assert len(pt) == ct.level + 1 + ev.o.np_, "double hoisting: plaintexts must cover the special primes"
                term = ev.mul_plain_ext(babies[b], pt, pt_scale)
                inner = term if inner is None else ev.add_ext(inner, term)
            if g % n:
                inner = ev.rotate_ext(ev.moddown(inner), g)
            acc = inner if acc is None else ev.add_ext(acc, inner)
        acc = ev.moddown(acc)
        return ev.rescale(acc) if rescale else acc
    for g, klist in sorted(by_giant.items()):
        inner = None
        for k in klist:
            b = k - g
            if b not in babies:
                babies[b] = ev.rotate(ct, b)
            pt = plains[k] if plains is not None else ev.encode(tile(np.roll(diags[k], g)), ct.level, pt_scale)   # rot_{-g}(diag)
            term = ev.mul_plain(babies[b], pt, pt_scale)
            inner = term if inner is None else ev.add(inner, term)
        inner = ev.rotate(inner, g)
        acc = inner if acc is None else ev.add(acc, inner)
    return ev.rescale(acc) if rescale else acc


# ------------------------------------------------------------------------------------------------ EvalMod
def chebyshev_coeffs(fn, degree):
    return np.polynomial.chebyshev.chebinterpolate(fn, degree)


def eval_chebyshev(ev, u, coeffs):
    """sum_k coeffs[k] T_k(u) for u in [-1,1], len(coeffs) a power of two; depth log2(len(coeffs)) levels: full binary
    Chebyshev splitting p = hi * T_half + lo down to degree-1 leaves, with exact top-down target scales (a leaf's constant
    is encoded at exactly the scale that makes the node's rescale land on the scale its parent asked for)."""
    k = len(coeffs).bit_length() - 1
    assert len(coeffs) == 1 << k and k >= 1
    powers = {1: u}
    for j in range(1, k):                     # T_{2^j} = 2 T_{2^(j-1)}^2 - 1
        p = powers[1 << (j - 1)]
        sq = ev.mul(p, p)
        powers[1 << j] = ev.add_const(ev.mul_int(sq, 2), -1.0)

    def rec(c, level_out, scale_out):
        if len(c) == 2:
            t1 = ev.drop(u, level_out + 1)
            cs = scale_out * ev.q(level_out + 1) / t1.scale
            r = ev.rescale(ev.mul_const(t1, float(c[1]), cs))
            r = Ct(r.data, r.level, scale_out)           # exact by construction (up to the rounding of the constant)
            return ev.add_const(r, float(c[0]))
        half = len(c) // 2
        hi = np.zeros(half)
        lo = np.array(c[:half], dtype=np.float64)
        hi[0] = c[half]
        for j in range(1, half):                  # T_{half+j} = 2 T_half T_j - T_{half-j}
            hi[j] = 2 * c[half + j]
            lo[half - j] -= c[half + j]
        th = ev.drop(powers[half], level_out + 1)
        h = rec(hi, level_out + 1, scale_out * ev.q(level_out + 1) / th.scale)
        prod = ev.mul(h, th)
        prod = Ct(prod.data, prod.level, scale_out)
        return ev.add(prod, rec(lo, level_out, scale_out))

    level_out = u.level - k
    return rec(np.asarray(coeffs, dtype=np.float64), level_out, float(ev.q(level_out + 1)))


def eval_monomial(ev, u, coeffs):
    """sum_k coeffs[k] u^k, len(coeffs) a power of two: the binary splitting of eval_chebyshev in the monomial basis
    (p = hi * u^half + lo), same exact top-down target scales"""
    k = len(coeffs).bit_length() - 1
    assert len(coeffs) == 1 << k and k >= 1
    powers = {1: u}
    for j in range(1, k):
        p = powers[1 << (j - 1)]
        powers[1 << j] = ev.mul(p, p)

    def rec(c, level_out, scale_out):
        if len(c) == 2:
            t1 = ev.drop(u, level_out + 1)
            cs = scale_out * ev.q(level_out + 1) / t1.scale
            r = ev.rescale(ev.mul_const(t1, float(c[1]), cs))
            r = Ct(r.data, r.level, scale_out)
            return ev.add_const(r, float(c[0]))
        half = len(c) // 2
        th = ev.drop(powers[half], level_out + 1)
        h = rec(np.array(c[half:], dtype=np.float64), level_out + 1, scale_out * ev.q(level_out + 1) / th.scale)
        prod = ev.mul(h, th)
        prod = Ct(prod.data, prod.level, scale_out)
        return ev.add(prod, rec(np.array(c[:half], dtype=np.float64), level_out, scale_out))

    level_out = u.level - k
    return rec(np.asarray(coeffs, dtype=np.float64), level_out, float(ev.q(level_out + 1)))


def arcsine_coeffs(degree):
    """Taylor coefficients of arcsin up to the (odd) degree, padded to a power of two: sum_k C(2k,k) / (4^k (2k+1)) y^(2k+1)"""
    n = 1
    while n < degree + 1:
        n *= 2
    c = np.zeros(n)
    binom, pow4 = 1.0, 1.0
    k = 0
    while 2 * k + 1 <= degree:
        if k > 0:
            binom = binom * (2 * k) * (2 * k - 1) / (k * k)
            pow4 *= 4.0
        c[2 * k + 1] = binom / (pow4 * (2 * k + 1))
        k += 1
    return c


def eval_mod(ev, u, K, double_angle, coeffs=None, asin=None, sine_deg=30):
    """u = v/K with v = I + eps (I integer, |I| < K, |eps| small) -> sin(2 pi v) ~ 2 pi eps, via the Chebyshev interpolant of
    cos(2 pi (K u - 1/4) / 2^r) on [-1, 1] (2^ceil(log2(sine_deg+1)) coefficients) and r double-angle steps y <- 2 y^2 - 1;
    with `asin` (monomial coefficients of the arcsine series) the result is arcsin(sin(2 pi v)) = 2 pi eps beyond the sine's
    own linear range (reference: btp_eval_mod_arcsine_deg, gpu_wrapper.cu:100-103)."""
    r = double_angle
    if coeffs is None:
        m = 1
        while m < sine_deg + 1:
            m *= 2
        coeffs = chebyshev_coeffs(lambda x: np.cos(2 * np.pi * (K * x - 0.25) / (1 << r)), m - 1)
    y = eval_chebyshev(ev, u, coeffs)
    for _ in range(r):
        sq = ev.mul(y, y)
        y = ev.add_const(ev.mul_int(sq, 2), -1.0)
    if asin is not None:
        y = eval_monomial(ev, y, asin)
    return y


# ------------------------------------------------------------------------------------------------ bootstrap
class Bootstrapper:
    def __init__(self, ev, cts_depth=4, stc_depth=3, K=16, double_angle=3, message_ratio=256.0, out_scale=None,
                 plains=None, coeffs=None, sine_deg=30, arcsine_deg=0, asin=None, double_hoist=True):
        """plains: {("cts"|"stc", matrix index): {k: plaintext}}, coeffs (Chebyshev coefficients) and asin (arcsine monomial
        coefficients) override this module's own floating-point constants with another implementation's (see linear_transform)."""
        self.ev = ev
        self.double_hoist = double_hoist
        self.plains, self.coeffs = plains, coeffs
        self.sine_deg, self.arcsine_deg = sine_deg, arcsine_deg
        self.cheb_depth = max(1, int(sine_deg).bit_length())               # ceil(log2(sine_deg + 1))
        self.asin_depth = int(arcsine_deg).bit_length() if arcsine_deg > 0 else 0
        self.asin = asin if asin is not None else (arcsine_coeffs(arcsine_deg) if arcsine_deg > 0 else None)
        self.n = ev.n // 2
        self.K, self.r, self.mr = K, double_angle, message_ratio
        self.cts_depth, self.stc_depth = cts_depth, stc_depth
        self.cts = merged_matrices(self.n, cts_depth, inverse=True)
        self.stc = merged_matrices(self.n, stc_depth, inverse=False)
        # constants folded into the first matrix of each transform: 1/n (inverse FFT), 1/2 (t + conj t), 1/K (unit interval)
        g = 1.0 / (2.0 * self.n * K)
        self.cts[0] = {k: d * g for k, d in self.cts[0].items()}
        self.out_scale = out_scale

    def expected_constants(self, in_scale, top_level):
        """This module's OWN floating-point constants for a bootstrap of a level-0 ciphertext at `in_scale`: every diagonal of
        every CoeffsToSlots / SlotsToCoeffs matrix encoded the way linear_transform() would encode it (level, scale, giant-step
        pre-rotation), keyed like `plains`, and the 32 Chebyshev coefficients.  Nothing here comes from the device library:
        tests/test_gpu_bootstrap.py compares the device plan's constants against these within a stated tolerance."""
        ev = self.ev
        n = self.n
        q0 = ev.q(0)
        c = max(1, int(round(q0 / (self.mr * in_scale))))
        d1 = in_scale * c

        def encode_matrix(diags, level):
            ks = sorted(diags)
            out = {}
            n1 = bsgs_split(ks, n, 2.0) if len(ks) >= 3 else 0
            for k in ks:
                g = (k // n1) * n1 if n1 else 0
                enc = ev.encode_ext if (self.double_hoist and n1) else ev.encode
                out[k] = enc(np.roll(diags[k], g), level, float(ev.q(level)))
            return out

        plains = {}
        level = top_level
        for i, m in enumerate(self.cts):
            plains[("cts", i)] = encode_matrix(m, level)
            level -= 1
        u_level = level
        level -= self.cheb_depth + self.r + self.asin_depth   # EvalMod: cosine interpolant + r double-angle steps (+ arcsine)
        natural = self.evalmod_out_scale(u_level) * 2 * np.pi * d1 / q0
        stc = list(self.stc)
        if self.out_scale is not None:
            kappa = self.out_scale / natural
            stc[0] = {k: d * kappa for k, d in stc[0].items()}
        for i, m in enumerate(stc):
            plains[("stc", i)] = encode_matrix(m, level)
            level -= 1
        coeffs = chebyshev_coeffs(lambda x: np.cos(2 * np.pi * (self.K * x - 0.25) / (1 << self.r)), (1 << self.cheb_depth) - 1)
        return plains, np.asarray(coeffs, dtype=np.float64), level

    def mod_raise(self, ct, top_level):
        """level-0 ciphertext -> same polynomials (centred mod q_0) over Q_top"""
        o = self.ev.o
        q0 = o.mod[0]
        out = np.zeros((2, top_level + 1, self.ev.n), dtype=np.uint64)
        for pl in range(2):
            coef = o.intt(0, ct.data[pl, 0]).tolist()
            cen = [x - q0 if x > q0 // 2 else x for x in coef]
            for j in range(top_level + 1):
                qj = o.mod[j]
                out[pl, j] = o.ntt(j, np.array([x % qj for x in cen], dtype=np.uint64))
        return out

    def evalmod_out_scale(self, level_in):
        """scale bookkeeping of eval_mod without data (it does not depend on the values)"""
        ev = self.ev
        level = level_in - self.cheb_depth        # 2^cheb_depth Chebyshev coefficients
        s = float(ev.q(level + 1))
        for _ in range(self.r):
            s = s * s / ev.q(level)
            level -= 1
        if self.asin_depth:
            s = float(ev.q(level - self.asin_depth + 1))
        return s

    def key_switch(self, ct, key, klvl):
        """(c0, c1) under s_in -> (c0 + ks0, ks1) under s_out with a generic switching key"""
        o = self.ev.o
        k0, k1 = o.keyswitch(ct.level, ct.data[1], key, klvl)
        d = np.stack([np.stack([o.vec("add", j, ct.data[0, j], k0[j]) for j in range(ct.level + 1)]), k1])
        return Ct(d, ct.level, ct.scale)

    def bootstrap(self, ct, top_level, swk_dts=None, swk_std=None):
        """swk_dts / swk_std: optional sparse-secret encapsulation (reference: custom_task.py:1989-1996): the ciphertext is
        switched to an ephemeral sparse secret before ModRaise (key at level 0) and back afterwards (key at the top level),
        so the integer part I of the raised polynomial stays within K for a dense main secret."""
        ev = self.ev
        assert ct.level == 0
        q0 = ev.q(0)
        # 1. scale the message up to q0 / message_ratio (integer factor: exact, no level)
        c = max(1, int(round(q0 / (self.mr * ct.scale))))
        d1 = ct.scale * c                      # plaintext becomes d1 * m
        ct = ev.mul_int(ct, c)
        if swk_dts is not None:
            ct = self.key_switch(ct, swk_dts, 0)
        # 2. ModRaise: polynomial P = d1*m + q0*I over the full chain; read with scale q0 its slots decode P / q0
        x = Ct(self.mod_raise(ct, top_level), top_level, float(q0))
        if swk_std is not None:
            x = self.key_switch(x, swk_std, top_level)
        # 3. CoeffsToSlots -> packed coefficients t / (2K) in bit-reversed order
        for i, m in enumerate(self.cts):
            x = linear_transform(ev, x, m, plains=self.plains[("cts", i)] if self.plains else None, double_hoist=self.double_hoist)
        xc = ev.conj(x)
        u_re = ev.add(x, xc)                                   # Re(t)/K
        u_im = ev.mul_by_i(ev.sub(x, xc), -1)                  # Im(t)/K
        # 4. EvalMod on both halves: sin(2 pi v) ~ 2 pi d1 m_k / q0
        y_re = eval_mod(ev, u_re, self.K, self.r, self.coeffs, self.asin, self.sine_deg)
        y_im = eval_mod(ev, u_im, self.K, self.r, self.coeffs, self.asin, self.sine_deg)
        y = ev.add(y_re, ev.mul_by_i(y_im, 1))
        # 5. SlotsToCoeffs.  The slots then hold (2 pi d1 / q0) * z at scale y.scale, i.e. z at scale y.scale * 2 pi d1 / q0;
        #    a requested output scale is met by folding the ratio into the first matrix (the caller pre-sets the scale on
        #    its output handle: it never crosses the ABI, unittests/test_gpu_ckks.cpp:406-410)
        natural = y.scale * 2 * np.pi * d1 / q0
        assert abs(y.scale / self.evalmod_out_scale(u_re.level) - 1) < 1e-9
        stc = list(self.stc)
        if self.out_scale is not None:
            kappa = self.out_scale / natural
            stc[0] = {k: d * kappa for k, d in stc[0].items()}
            natural = self.out_scale
        for i, m in enumerate(stc):
            y = linear_transform(ev, y, m, plains=self.plains[("stc", i)] if self.plains else None, double_hoist=self.double_hoist)
        return Ct(y.data, y.level, natural)


class SparseBootstrapper(Bootstrapper):
    """Sparse packing: 2^log_slots < N/2 slots (the reference also tests log_slots = 11 at N = 2^16, unittests/fixture.hpp:
    152-162).  The plaintext lives in the subring of X^gap (gap = (N/2)/slots), its slot vector has period `slots`.
      SubSum: ct += rot_{2^i}(ct), i = log_slots .. logN-2 -- the trace onto the subring: the integer polynomial I of the
        raised ciphertext loses its coefficients off the subring, the rest is multiplied by gap (folded into CoeffsToSlots);
      CoeffsToSlots on `slots` points; the last matrix M is applied as  y = P1 x + P2 conj(x)  with diagonals of period
        2*slots: first half Re(Mx) = (M x + conj(M) conj(x))/2, second half Im(Mx) = (-i M x + i conj(M) conj(x))/2, so ONE
        EvalMod serves both coefficient halves ("repack imag to real", frontend/bootstrap_params.py:74);
      SlotsToCoeffs: first matrix = (first group of forward layers) o R with the two-diagonal repack R = {0, slots}
        (bootstrap_params.py:233, :121-133): t[k] = y[k] + i y[k+slots]."""

    def __init__(self, ev, log_slots, cts_depth=4, stc_depth=3, K=16, double_angle=3, message_ratio=256.0, out_scale=None,
                 plains=None, coeffs=None, sine_deg=30, arcsine_deg=0, asin=None, double_hoist=True):
        """plains keys: ("cts", i) for the leading matrices, ("p1",), ("p2",), ("stc", i)"""
        self.ev = ev
        self.double_hoist = double_hoist
        self.plains, self.coeffs = plains, coeffs
        self.sine_deg, self.arcsine_deg = sine_deg, arcsine_deg
        self.cheb_depth = max(1, int(sine_deg).bit_length())
        self.asin_depth = int(arcsine_deg).bit_length() if arcsine_deg > 0 else 0
        self.asin = asin if asin is not None else (arcsine_coeffs(arcsine_deg) if arcsine_deg > 0 else None)
        self.n = ev.n // 2
        self.ns = ns = 1 << log_slots
        self.log_slots = log_slots
        assert ns < self.n
        self.K, self.r, self.mr = K, double_angle, message_ratio
        self.out_scale = out_scale
        gap = self.n // ns
        cts = merged_matrices(ns, cts_depth, inverse=True)
        last = cts.pop()
        half = lambda a, b: np.concatenate([a, b])
        self.p1 = {k: half(0.5 * d, -0.5j * d) for k, d in last.items()}
        self.p2 = {k: half(0.5 * np.conj(d), 0.5j * np.conj(d)) for k, d in last.items()}
        g = 1.0 / (ns * gap * K)              # 1/slots (inverse FFT), 1/gap (SubSum), 1/K (unit interval)
        if cts:
            cts[0] = {k: d * g for k, d in cts[0].items()}
        else:
            self.p1 = {k: d * g for k, d in self.p1.items()}
            self.p2 = {k: d * g for k, d in self.p2.items()}
        self.cts = cts
        # SlotsToCoeffs: forward layers on `slots` points, grouped as for the dense case; the first group is preceded by R
        log_ns = log_slots
        lengths = [1 << l for l in range(1, log_ns + 1)]
        sizes, left = [], log_ns
        for i in range(stc_depth):
            sz = -(-left // (stc_depth - i))
            sizes.append(sz)
            left -= sz
        sizes = sizes[::-1]
        ones, eye = np.ones(ns), 1j * np.ones(ns)
        m = {0: half(ones, eye), ns: half(eye, ones)}          # R on period 2*slots
        mats, pos = [], 0
        for gi, sz in enumerate(sizes):
            for length in lengths[pos: pos + sz]:
                lay = layer_diagonals(ns, length, False)
                if gi == 0:
                    lay = {k: np.tile(d, 2) for k, d in lay.items()}
                    m = compose(m, lay, 2 * ns)
                else:
                    m = lay if m is None else compose(m, lay, ns)
            mats.append(m)
            m = None
            pos += sz
        self.stc = mats

    def bootstrap(self, ct, top_level, swk_dts=None, swk_std=None):
        ev = self.ev
        ns = self.ns
        assert ct.level == 0
        q0 = ev.q(0)
        c = max(1, int(round(q0 / (self.mr * ct.scale))))
        d1 = ct.scale * c
        ct = ev.mul_int(ct, c)
        if swk_dts is not None:
            ct = self.key_switch(ct, swk_dts, 0)
        x = Ct(self.mod_raise(ct, top_level), top_level, float(q0))
        if swk_std is not None:
            x = self.key_switch(x, swk_std, top_level)
        for i in range(self.log_slots, ev.n.bit_length() - 2):          # SubSum
            x = ev.add(x, ev.rotate(x, 1 << i))
        pl = self.plains or {}
        for i, m in enumerate(self.cts):
            x = linear_transform(ev, x, m, n_slots=ns, plains=pl.get(("cts", i)), double_hoist=self.double_hoist)
        a = linear_transform(ev, x, self.p1, rescale=False, n_slots=2 * ns, plains=pl.get(("p1",)), double_hoist=self.double_hoist)
        b = linear_transform(ev, ev.conj(x), self.p2, rescale=False, n_slots=2 * ns, plains=pl.get(("p2",)), double_hoist=self.double_hoist)
        u = ev.rescale(ev.add(a, b))                                    # [Re(t)/K | Im(t)/K], period 2*slots
        y = eval_mod(ev, u, self.K, self.r, self.coeffs, self.asin, self.sine_deg)
        natural = y.scale * 2 * np.pi * d1 / q0
        stc = list(self.stc)
        if self.out_scale is not None:
            kappa = self.out_scale / natural
            stc[0] = {k: d * kappa for k, d in stc[0].items()}
            natural = self.out_scale
        y = linear_transform(ev, y, stc[0], n_slots=2 * ns, plains=pl.get(("stc", 0)), double_hoist=self.double_hoist)
        for i, m in enumerate(stc[1:], 1):
            y = linear_transform(ev, y, m, n_slots=ns, plains=pl.get(("stc", i)), double_hoist=self.double_hoist)
        return Ct(y.data, y.level, natural)
