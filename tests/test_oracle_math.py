"""Pins the CPU oracle by mathematics (the reference holds no golden vectors for this path):
NTT == evaluation at odd powers of psi, NTT-convolution == the reference's negacyclic schoolbook definition
(fhe_ops_lib/utils.cpp:87-102), base conversion == big-integer CRT, rescale == big-integer rounded division."""
import os

import numpy as np
import pytest

from lattisense_amd import params
from oracle.pyoracle import Oracle, lib

Q3 = params.BFV_DEFAULT[8192]["q"]
P1 = params.BFV_DEFAULT[8192]["p"]


def brv(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2)


def test_default_moduli_are_ntt_primes():
    L = lib()
    for sets in (params.BFV_DEFAULT, params.CKKS_DEFAULT):
        for n, v in sets.items():
            for q in v["q"] + v["p"]:
                assert L.ora_is_prime(q) and q % (2 * n) == 1
    for q in params.CKKS_BOOTSTRAP_65536["q"] + params.CKKS_BOOTSTRAP_65536["p"]:
        assert L.ora_is_prime(q) and q % (2 * 65536) == 1


def test_primitive_root_is_smallest_generator():
    L = lib()
    for q in [17, 257, 65537, 1073750017, Q3[0]]:
        g = L.ora_primitive_root(q)
        # order of g is q-1: check against factorisation by brute force for small q only
        if q < 70000:
            seen, x = set(), 1
            for _ in range(q - 1):
                x = x * g % q
                seen.add(x)
            assert len(seen) == q - 1
            for h in range(2, g):
                x, ok = 1, True
                s2 = set()
                for _ in range(q - 1):
                    x = x * h % q
                    s2.add(x)
                assert len(s2) < q - 1
    # psi is a primitive 2N-th root
    n = 8192
    psi = L.ora_psi(Q3[0], n)
    assert pow(psi, n, Q3[0]) == Q3[0] - 1


@pytest.mark.parametrize("n", [16, 64, 256])
def test_ntt_is_evaluation_at_odd_psi_powers(n):
    q = [Q3[0]]
    o = Oracle(n, q, [], 0)
    logn = n.bit_length() - 1
    rng = np.random.default_rng(1)
    a = rng.integers(0, q[0], size=n, dtype=np.uint64)
    A = o.ntt(0, a)
    psi = lib().ora_psi(q[0], n)
    al = [int(x) for x in a]
    for i in range(n):
        e = 2 * brv(i, logn) + 1
        w = pow(psi, e, q[0])
        acc = 0
        for j in reversed(range(n)):
            acc = (acc * w + al[j]) % q[0]
        assert acc == int(A[i])
    assert np.array_equal(o.intt(0, A), a)


@pytest.mark.parametrize("n", [32, 128])
def test_ntt_convolution_is_negacyclic_schoolbook(n):
    q = Q3[1]
    o = Oracle(n, [q], [], 0)
    rng = np.random.default_rng(2)
    a = rng.integers(0, q, size=n, dtype=np.uint64)
    b = rng.integers(0, q, size=n, dtype=np.uint64)
    c = o.intt(0, o.vec("mul", 0, o.ntt(0, a), o.ntt(0, b)))
    al, bl = [int(x) for x in a], [int(x) for x in b]
    want = [0] * n
    for i in range(n):
        for j in range(n):
            k = i + j
            if k < n:
                want[k] = (want[k] + al[i] * bl[j]) % q
            else:
                want[k - n] = (want[k - n] - al[i] * bl[j]) % q
    assert [int(x) for x in c] == want


def _crt(limbs, mods):
    S = 1
    for m in mods:
        S *= m
    out = []
    for x in range(len(limbs[0])):
        v = 0
        for li, m in zip(limbs, mods):
            Si = S // m
            v += int(li[x]) * Si * pow(Si % m, -1, m)
        out.append(v % S)
    return out, S


@pytest.mark.parametrize("centered", [0, 1])
def test_baseconv_matches_bigint(centered):
    n = 64
    q = params.BFV_DEFAULT[16384]["q"]
    p = params.BFV_DEFAULT[16384]["p"]
    o = Oracle(n, [x for x in q], [x for x in p], 0)  # moduli need not be NTT-friendly for this n? they are (==1 mod 2^15)
    rng = np.random.default_rng(3)
    sidx = [0, 1, 2]
    didx = [3, 4, 5, 6, 7]
    src = np.stack([rng.integers(0, o.mod[i], size=n, dtype=np.uint64) for i in sidx])
    # boundary: x = 0.  (x within ~S*2^-52 of S is where the float64 quotient estimate of the published
    # algorithm can be off by one - inherent to it, so not asserted against the big-integer value.)
    src[:, 0] = 0
    dst = o.baseconv(sidx, didx, src, centered)
    vals, S = _crt(src, [o.mod[i] for i in sidx])
    for j, mi in enumerate(didx):
        for x in range(n):
            v = vals[x]
            if centered:
                v = (v + S // 2) % S - S // 2
            assert int(dst[j][x]) == v % o.mod[mi]


def test_ckks_rescale_is_rounded_division():
    n = 32
    q = params.CKKS_DEFAULT[16384]["q"][:3]
    o = Oracle(n, q, [], 0)
    rng = np.random.default_rng(4)
    lvl = 2
    coeff = np.stack([rng.integers(0, q[i], size=n, dtype=np.uint64) for i in range(3)])
    ct = np.stack([o.ntt(i, coeff[i]) for i in range(3)])[None]
    out = o.ckks_rescale(lvl, np.ascontiguousarray(ct))
    vals, S = _crt(coeff, q)
    ql = q[2]
    for i in range(2):
        got = o.intt(i, out[0, i])
        for x in range(n):
            v = vals[x]
            r = ((v % ql) + (ql - 1) // 2) % ql - (ql - 1) // 2   # centred remainder
            assert (v - r) % ql == 0
            assert int(got[x]) == ((v - r) // ql) % q[i]


def test_automorphism_ntt_matches_coefficient_domain():
    n = 64
    q = Q3[0]
    o = Oracle(n, [q], [], 0)
    rng = np.random.default_rng(5)
    a = rng.integers(0, q, size=n, dtype=np.uint64)
    for g in (5, 25, 2 * n - 1, pow(5, 7, 2 * n)):
        lhs = o.automorph_ntt(g, o.ntt(0, a))
        rhs = o.ntt(0, o.automorph_coeff(0, g, a))
        assert np.array_equal(lhs, rhs)


def test_fast_mulmod_build_agrees(tmp_path):
    """the -O3 -march=native timing build of the oracle (bench.py cpu_baseline) computes the same residues as the checker
    build: mulmod on edge operands and a whole CKKS HMult+relin+rescale"""
    import ctypes
    import subprocess
    import sys
    from lattisense_amd import params
    from oracle import pyoracle
    fast = pyoracle.build_fast(str(tmp_path))
    L = ctypes.CDLL(fast)
    L.ora_mulmod.restype = ctypes.c_uint64
    L.ora_mulmod.argtypes = [ctypes.c_uint64] * 3
    ref = pyoracle.lib()
    ref.ora_mulmod.restype = ctypes.c_uint64
    ref.ora_mulmod.argtypes = [ctypes.c_uint64] * 3
    rng = np.random.default_rng(3)
    for q in params.CKKS_DEFAULT[65536]["q"][:3] + params.CKKS_BOOTSTRAP_65536["p"][:2] + [3, 5, 65537]:
        xs = [0, 1, q - 1, q // 2] + [int(v) for v in rng.integers(0, q, size=200, dtype=np.uint64)]
        for a in xs[:40]:
            for b in xs[::7]:
                assert L.ora_mulmod(a, b, q) == ref.ora_mulmod(a, b, q) == a * b % q
    # a whole operator in a child process that loads the fast build through LS_ORACLE_LIB
    code = ("import numpy as np, hashlib\nfrom lattisense_amd import params\nfrom oracle.pyoracle import Oracle\n"
            "P = params.CKKS_DEFAULT[16384]\nn, q, p = 2048, P['q'][:4], P['p']\no = Oracle(n, q, p, 0)\n"
            "rng = np.random.default_rng(1)\n"
            "def ct():\n    c = np.empty((2, 4, n), dtype=np.uint64)\n"
            "    for j, m in enumerate(q): c[:, j] = rng.integers(0, m, size=(2, n), dtype=np.uint64)\n    return c\n"
            "k = np.empty((2, 2, 6, n), dtype=np.uint64)\n"
            "for j, m in enumerate(q + p): k[:, :, j] = rng.integers(0, m, size=(2, 2, n), dtype=np.uint64)\n"
            "print(hashlib.sha256(o.ckks_mult_relin_rescale(3, ct(), ct(), k, 3).tobytes()).hexdigest())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env_lib in (None, fast):
        env = dict(os.environ, PYTHONPATH=root)
        env.pop("LS_ORACLE_LIB", None)
        if env_lib:
            env["LS_ORACLE_LIB"] = env_lib
        outs.append(subprocess.check_output([sys.executable, "-c", code], env=env, text=True).strip())
    assert outs[0] == outs[1] and len(outs[0]) == 64
