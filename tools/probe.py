"""Hardware ceilings used in DESIGN.md: streaming copy GB/s, 64-bit Montgomery-multiply rate, NTT GB/s by size."""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lattisense_amd import params  # noqa: E402
from lattisense_amd._native import check, lib  # noqa: E402
from lattisense_amd.device import ALGO_CKKS, DeviceContext  # noqa: E402


def timed(ctx, fn, iters=10, warm=2):
    L = lib()
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    check(L.lsa_event_create(ctx.h, ctypes.byref(e0)))
    check(L.lsa_event_create(ctx.h, ctypes.byref(e1)))
    for _ in range(warm):
        fn()
    ctx.sync()
    check(L.lsa_event_record(ctx.h, e0, ctx.stream))
    for _ in range(iters):
        fn()
    check(L.lsa_event_record(ctx.h, e1, ctx.stream))
    ms = ctypes.c_float()
    check(L.lsa_event_elapsed_ms(ctx.h, e0, e1, ctypes.byref(ms)))
    return ms.value / iters


def main():
    out = {}
    B = params.CKKS_BOOTSTRAP_65536
    ctx = DeviceContext(ALGO_CKKS, 65536, B["q"][:13], B["p"][:4])
    L = lib()
    nw = 1 << 28  # 2 GiB each
    a, b = ctx.alloc(nw), ctx.alloc(nw)
    ms = timed(ctx, lambda: check(L.lsa_probe_copy(ctx.h, b.ptr, a.ptr, nw, ctx.stream)))
    out["copy_GBps"] = 2 * nw * 8 / ms / 1e6
    nthreads_words = 256 * 256 * 16 * 4
    iters = 512
    ms = timed(ctx, lambda: check(L.lsa_probe_mulhi(ctx.h, a.ptr, nthreads_words, iters, ctx.stream)))
    out["montmul_per_s"] = nthreads_words * iters / (ms * 1e-3)
    a.free(); b.free()
    for logn, nprimes, batch in [(14, 4, 1024 * 2), (16, 13, 64), (16, 13, 256), (13, 3, 4096), (12, 3, 8192)]:
        n = 1 << logn
        c2 = ctx if logn == 16 else DeviceContext(ALGO_CKKS, n, B["q"][:nprimes], B["p"][:1])
        rows = nprimes
        buf = c2.alloc(batch * rows * n)
        mo = list(range(nprimes))
        f = timed(c2, lambda: c2.ntt(buf, batch, rows, mo, False))
        i = timed(c2, lambda: c2.ntt(buf, batch, rows, mo, True))
        gb = batch * rows * n * 16 / 1e9
        out["ntt_logn%d_rows%d" % (logn, batch * rows)] = {"fwd_ms": f, "inv_ms": i, "fwd_GBps": gb / f * 1e3,
                                                          "inv_GBps": gb / i * 1e3}
        buf.free()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
