"""Per-phase shader-clock breakdown of one NTT pass (needs the variant build `python -m lattisense_amd.build --variant stamps -DLSA_NTT_DIAG_STAMPS`, selected with LSA_NATIVE_LIB=lattisense_amd/variants/libstamps.so).
Stamps per workgroup: 0 start, 1 loads issued, 2 tile in LDS (barrier), 3/4 sub-pass 1 done / barrier, 5/6 sub-pass 2, 7 stores issued."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lattisense_amd import params
from lattisense_amd._native import lib
from lattisense_amd.device import ALGO_CKKS, DeviceContext

D = params.CKKS_DEFAULT[65536]
n, batch = 1 << 16, 64
ctx = DeviceContext(ALGO_CKKS, n, D["q"][:13], D["p"])
stamps = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
out = {}
for name, mods in (("fp64", list(range(1, 13))), ("int", [0, 13, 14, 15, 16] * 2 + [0, 13])):
    buf = ctx.upload(np.random.default_rng(1).integers(0, 1 << 44, batch * len(mods) * n, dtype=np.uint64))
    for inverse in (False, True):
        for _ in range(3):
            ctx.ntt(buf, batch, len(mods), mods, inverse)
        torch.cuda.synchronize()
        lib().lsa_debug_set_ntt_stamps(ctx.h, ctypes.c_void_p(stamps.data_ptr()))
        ctx.ntt(buf, batch, len(mods), mods, inverse)   # the stamps of the LAST pass of this transform survive
        torch.cuda.synchronize()
        lib().lsa_debug_set_ntt_stamps(ctx.h, None)
        t = stamps.cpu().numpy().reshape(8192, 8).astype(np.int64)
        d = np.diff(t, axis=1)
        # steady state: skip the first dispatch wave
        sel = slice(2048, 8192)
        names = ["load_issue", "load_wait+bar", "sub1", "bar1", "sub2", "bar2", "store_issue"]
        out["%s_%s" % (name, "inv" if inverse else "fwd")] = dict(
            {k: int(np.median(d[sel, i])) for i, k in enumerate(names)}, total=int(np.median(t[sel, 7] - t[sel, 0])),
            span_all=int(t[:, 7].max() - t[:, 0].min()))
    buf.free()
print(json.dumps(out, indent=1))
