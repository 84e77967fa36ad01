"""Forward NTT launches for PMC collection: 4 transforms on FP64-engine limbs, then 4 on integer-engine limbs (N=2^16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lattisense_amd import params
from lattisense_amd.device import ALGO_CKKS, DeviceContext

D = params.CKKS_DEFAULT[65536]
n, batch = 1 << 16, 256
ctx = DeviceContext(ALGO_CKKS, n, D["q"][:13], D["p"])
for mods in (list(range(1, 13)), [0, 13, 14, 15, 16] * 2 + [0, 13]):
    buf = ctx.alloc(batch * len(mods) * n)
    for _ in range(4):
        ctx.ntt(buf, batch, len(mods), mods, False)
    ctx.synchronize()
    buf.free()
