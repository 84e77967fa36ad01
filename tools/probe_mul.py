"""64-bit modular-multiply issue ceilings: Montgomery (as used by the integer engine) vs Shoup/Harvey (precomputed quotient)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lattisense_amd import params
from lattisense_amd._native import check, lib
from lattisense_amd.device import ALGO_CKKS, DeviceContext
from tools.probe import timed
B = params.CKKS_BOOTSTRAP_65536
ctx = DeviceContext(ALGO_CKKS, 1 << 12, B["q"][:3], B["p"][:1])
L = lib()
nw = 256 * 256 * 16 * 4
a = ctx.alloc(nw)
# lane-operations per second at full rate: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 3.93e13
for name, iters, ops in (("montgomery", 512, "11 32-bit multiplies + adds"), ("shoup", -512, "10 32-bit multiplies + adds"),
                         ("fp64 (6 double operations)", (1 << 20) + 512, "6 double ops")):
    ms = timed(ctx, lambda: check(L.lsa_probe_mulhi(ctx.h, a.ptr, nw, iters, ctx.stream)))
    rate = nw * 512 / (ms * 1e-3)
    print(name, "%.3e modmul/s" % rate, "(%s; %.1f full-rate lane-op slots per modmul)" % (ops, 3.93e13 / rate))
