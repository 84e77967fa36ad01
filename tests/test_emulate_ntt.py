"""Replays the HIP NTT kernel's phase functions on the CPU (one simulated thread at a time) and checks them
bit-exactly against the oracle.  This validates the kernel's tile/sub-pass/twiddle indexing without a GPU;
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
    deps = srcs + [os.path.join(CSRC, f) for f in ("ntt_core.h", "ntt_plan.h", "modarith.h", "tables.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-DLSA_EMULATE", "-o", so] + srcs)
    L = ctypes.CDLL(so)
    L.lsa_emu_ntt.restype = ctypes.c_int
    return L


def run(emu, n, mods, data, rows, mod_of, inverse, tau):
    batch = data.shape[0]
    arr = (ctypes.c_uint64 * len(mods))(*mods)
    mo = (ctypes.c_ubyte * len(mod_of))(*mod_of)
    emu.lsa_emu_ntt(ctypes.c_int(n), arr, len(mods), data.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                    batch, ctypes.c_longlong(rows * n), rows, mo, len(mod_of), int(inverse), tau)


# (logn, tau_max): single pass, two passes with several (mu_a, mu_b) splits incl. odd logn
@pytest.mark.parametrize("logn,tau", [(8, 12), (10, 12), (12, 12), (13, 12), (14, 12), (11, 8), (13, 10), (15, 12)])
def test_emulated_kernel_matches_oracle(emu, logn, tau):
    n = 1 << logn
    mods = [m for m in (params.CKKS_BOOTSTRAP_65536["q"][:2] + params.CKKS_BOOTSTRAP_65536["p"][:1])]
    o = Oracle(n, mods, [], 0)
    rng = np.random.default_rng(logn)
    batch, rows = 2, 3
    data = np.stack([np.stack([rng.integers(0, mods[r], size=n, dtype=np.uint64) for r in range(rows)])
                     for _ in range(batch)])
    data[0, 0, :4] = [0, mods[0] - 1, 1, mods[0] - 2]
    want = np.stack([np.stack([o.ntt(r, data[b, r]) for r in range(rows)]) for b in range(batch)])
    got = data.copy()
    run(emu, n, mods, got, rows, [0, 1, 2], 0, tau)
    assert np.array_equal(got, want)
    run(emu, n, mods, got, rows, [0, 1, 2], 1, tau)
    assert np.array_equal(got, data)
