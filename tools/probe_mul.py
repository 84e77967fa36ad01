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
for name, iters in (("montgomery", 512), ("shoup", -512)):
    ms = timed(ctx, lambda: check(L.lsa_probe_mulhi(ctx.h, a.ptr, nw, iters, ctx.stream)))
    print(name, "%.3e modmul/s" % (nw * 512 / (ms * 1e-3)))
