"""Whole-limb (single-pass, LSA_NTT_WIDE=1) against two-pass NTT at N = 2^13 / 2^14 on both butterfly engines.
Usage: LSA_NTT_WIDE=0|1 python tools/probe_wide.py   (prints algorithmic GB/s, forward + inverse per iteration)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lattisense_amd import params  # noqa: E402
from lattisense_amd.device import DeviceContext, ALGO_CKKS  # noqa: E402

CASES = [("N=2^14 fp64 engine (CKKS 35-bit primes)", 16384, params.CKKS_DEFAULT[16384]["q"][1:9]),
         ("N=2^14 integer engine (BFV 54-57-bit primes)", 16384, params.BFV_DEFAULT[16384]["q"][:4] + params.BFV_DEFAULT[16384]["q"][:4]),
         ("N=2^13 integer engine (BFV 54-55-bit primes)", 8192, params.BFV_DEFAULT[8192]["q"][:3] + params.BFV_DEFAULT[8192]["q"][:3])]
for name, n, mods in CASES:
    uniq = sorted(set(mods))
    ctx = DeviceContext(ALGO_CKKS, n, uniq, [])
    mo = [uniq.index(m) for m in mods]
    rows = len(mods)
    batch = (1 << 30) // (rows * n * 8)     # 1 GiB of limbs
    rng = np.random.default_rng(1)
    one = np.stack([rng.integers(0, m, size=n, dtype=np.uint64) for m in mods])
    buf = ctx.upload(np.broadcast_to(one, (batch, rows, n)).copy())
    for it in range(2):
        ctx.ntt(buf, batch, rows, mo, inverse=False)
        ctx.ntt(buf, batch, rows, mo, inverse=True)
    ctx.sync()
    reps = 10
    t0 = time.perf_counter()
    for it in range(reps):
        ctx.ntt(buf, batch, rows, mo, inverse=False)
        ctx.ntt(buf, batch, rows, mo, inverse=True)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    back = ctx.download(buf, (batch, rows, n))
    assert np.array_equal(back[0], one) and np.array_equal(back[-1], one)
    print(f"LSA_NTT_WIDE={os.environ.get('LSA_NTT_WIDE', 'auto')} {name}: {2 * batch * rows * 16.0 * n / dt / 1e9:.0f} GB/s algorithmic")
    buf.free()
    ctx.close()
