"""Replays the HIP NTT kernel's phase functions on the CPU (one simulated thread at a time) and checks them
bit-exactly against the oracle, for both butterfly engines (integer Shoup; exact FP64-FMA for q < 2^47).
This validates the kernel's tile/sub-pass/twiddle indexing and the FP64 exactness argument without a GPU;
the real kernel is checked on the GPU in tests/test_gpu_*.py."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from lattisense_amd import params
from oracle.pyoracle import Oracle

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lattisense_amd", "csrc")


@pytest.fixture(scope="module")
def emu():
    so = os.path.join(CSRC, "libls_emu.so")
    srcs = [os.path.join(CSRC, f) for f in ("emu_ntt.cpp", "tables.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("ntt_core.h", "ntt_r16.h", "ntt_plan.h", "modarith.h", "tables.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-DLSA_EMULATE"] + os.environ.get("LSA_EXTRA_FLAGS", "").split() + ["-o", so] + srcs)
    L = ctypes.CDLL(so)
    L.lsa_emu_ntt.restype = ctypes.c_int
    return L


def run(emu, n, mods, data, rows, mod_of, inverse, tau, fp64=1):
    batch = data.shape[0]
    arr = (ctypes.c_uint64 * len(mods))(*mods)
    mo = (ctypes.c_ubyte * len(mod_of))(*mod_of)
    emu.lsa_emu_ntt(ctypes.c_int(n), arr, len(mods), data.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                    batch, ctypes.c_longlong(rows * n), rows, mo, len(mod_of), int(inverse), tau, int(fp64))


def _check(emu, logn, tau, mods, fp64):
    n = 1 << logn
    o = Oracle(n, mods, [], 0)
    rng = np.random.default_rng(logn)
    batch, rows = 2, len(mods)
    data = np.stack([np.stack([rng.integers(0, mods[r], size=n, dtype=np.uint64) for r in range(rows)])
                     for _ in range(batch)])
    data[0, :, 0] = 0
    data[0, :, 1] = np.array([m - 1 for m in mods], dtype=np.uint64)   # extreme residues
    data[1, :, :] = np.array([m - 1 for m in mods], dtype=np.uint64)[:, None]  # all-max limb: worst-case growth
    want = np.stack([np.stack([o.ntt(r, data[b, r]) for r in range(rows)]) for b in range(batch)])
    got = data.copy()
    run(emu, n, mods, got, rows, list(range(rows)), 0, tau, fp64)
    assert np.array_equal(got, want)
    run(emu, n, mods, got, rows, list(range(rows)), 1, tau, fp64)
    assert np.array_equal(got, data)


# (logn, tau_max): single pass (incl. the whole-limb 2^13 / 2^14 shapes), two passes with several (mu_a, mu_b) splits incl. odd logn
@pytest.mark.parametrize("logn,tau", [(8, 12), (10, 12), (12, 12), (13, 12), (14, 12), (11, 8), (13, 10), (15, 12), (13, 13), (14, 14)])
def test_integer_engine_matches_oracle(emu, logn, tau):
    B = params.CKKS_BOOTSTRAP_65536
    _check(emu, logn, tau, B["q"][:2] + B["p"][:1], fp64=0)   # 60-, 40-, 61-bit


@pytest.mark.parametrize("logn,tau", [(9, 12), (10, 12), (12, 12), (13, 12), (14, 12), (15, 12), (13, 10), (13, 13), (14, 14)])
def test_fp64_engine_matches_oracle(emu, logn, tau):
    # 46/45-bit (largest primes the FP64 engine accepts in the default chains), 40- and 39-bit, plus a 60-bit limb that
    # must keep using the integer engine inside the same launch
    D = params.CKKS_DEFAULT[65536]
    B = params.CKKS_BOOTSTRAP_65536
    mods = [D["q"][1], D["q"][3], B["q"][1], B["q"][10], B["q"][0]]
    assert max(m.bit_length() for m in mods[:4]) <= 46
    _check(emu, logn, tau, mods, fp64=1)
    _check(emu, logn, tau, mods, fp64=0)
    _check(emu, logn, tau, mods, fp64=3)   # the interleaved workgroup order of mixed-engine launches


@pytest.mark.parametrize("logn", [14, 15, 16, 17, 18])
def test_radix16_squared_passes_match_oracle(emu, logn):
    """the 8- and 7-stage passes of two-pass plans through ntt_r16.h (both passes at N = 2^14 .. 2^16, the first at 2^17) and the
    9-stage second pass of N = 2^17 as three radix-8 groups per point: both engines, forward and inverse, raw FP64 hand-off"""
    D = params.CKKS_DEFAULT[65536]
    B = params.CKKS_BOOTSTRAP_65536
    if logn >= 17:   # (2^18: nine-stage first pass on the staged kernel, nine-stage second pass as three radix-8 groups)
        P = params.ckks_n17_chain()
        mods = [m for m in (P["q"][0], P["q"][1], P["p"][0]) if (m - 1) % (2 << logn) == 0]
        if len(mods) < 2:
            pytest.skip("the generated N = 2^17 chain has no two primes = 1 mod 2^%d" % (logn + 1))
    else:
        mods = [D["q"][1], B["q"][10], B["q"][0], B["p"][0]]   # 46- and 39-bit (FP64 engine), 60- and 61-bit (integer engine)
    _check(emu, logn, 12, mods, fp64=4 | 1)
    _check(emu, logn, 12, mods, fp64=4)
    _check(emu, logn, 12, mods, fp64=4 | 3)


def test_skipped_rows_get_no_workgroups(emu):
    """rows whose modulus is LSA_ROW_SKIP (0xFF) are left untouched and the grid is compacted to the active rows"""
    B = params.CKKS_BOOTSTRAP_65536
    mods = B["q"][:3]
    n = 1 << 13
    o = Oracle(n, mods, [], 0)
    rng = np.random.default_rng(4)
    rows, mod_of = 5, [0, 0xFF, 2, 0xFF, 1]
    data = np.stack([np.stack([rng.integers(0, mods[m if m != 0xFF else 0], size=n, dtype=np.uint64) for m in mod_of]) for _ in range(2)])
    got = data.copy()
    run(emu, n, mods, got, rows, mod_of, 0, 12, 1)
    for b in range(2):
        for r, m in enumerate(mod_of):
            want = data[b, r] if m == 0xFF else o.ntt(m, data[b, r])
            assert np.array_equal(got[b, r], want), (b, r)
