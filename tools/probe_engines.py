"""NTT GB/s by butterfly engine at N=2^16 / 2^14 (integer Montgomery vs FP64-FMA), same launch shapes."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lattisense_amd import params
from lattisense_amd.device import ALGO_CKKS, DeviceContext
from tools.probe import timed

out = {}
D = params.CKKS_DEFAULT[65536]
for logn, batch in [(16, 64), (14, 256)]:
    n = 1 << logn
    ctx = DeviceContext(ALGO_CKKS, n, D["q"][:13], D["p"])
    for name, mods in [("fp64_small_primes", list(range(1, 13))), ("int_big_primes", [0, 13, 14, 15, 16] * 2 + [0, 13])]:
        rows = len(mods)
        buf = ctx.alloc(batch * rows * n)
        for fp in (1, 0):
            ctx.set_fp64_ntt(fp)
            f = timed(ctx, lambda: ctx.ntt(buf, batch, rows, mods, False))
            i = timed(ctx, lambda: ctx.ntt(buf, batch, rows, mods, True))
            gb = batch * rows * n * 16 / 1e9
            out["logn%d_%s_fp64=%d" % (logn, name, fp)] = (round(gb / f * 1e3), round(gb / i * 1e3))
        buf.free()
print(json.dumps(out, indent=1))
