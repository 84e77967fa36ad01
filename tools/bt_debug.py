"""Step-by-step comparison of the device bootstrap with the oracle program (LSA_BT_STOP diagnostic of bootstrap.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lattisense_amd import params
from lattisense_amd.device import ALGO_CKKS, DeviceContext, BootstrapPlan
from oracle.client import Client, mean_precision_bits
from oracle.pyoracle import Oracle
from oracle.ckks_bootstrap import *

B = params.CKKS_BOOTSTRAP_65536
N = 1 << 10
o = Oracle(N, B["q"], B["p"], 0)
c = Client(o, seed=21, hamming=32)
ctx = DeviceContext(ALGO_CKKS, N, B["q"], B["p"])
top = len(B["q"]) - 1
D = float(2 ** 40)
plan = BootstrapPlan(ctx, in_scale=D, out_scale=D)
ev = Evaluator(o, c, top)
rlk = ctx.upload_key(ev.rlk, top)
keys = {e: c.gen_galois_key(e, top) for e in plan.galois_elements}
ev.glk = dict(keys)      # the oracle must rotate with the SAME keys (fresh ones differ in their noise)
glk = {e: ctx.upload_key(k, top) for e, k in keys.items()}
rng = np.random.default_rng(1)
z = rng.uniform(-1, 1, N // 2) + 1j * rng.uniform(-1, 1, N // 2)
ct0 = c.ckks_encrypt(z, 0, D)
plains = {}
for i in range(plan.n_matrices):
    _, _, _, pts = plan.matrix(i)
    plains[("cts", i) if i < plan.n_cts else ("stc", i - plan.n_cts)] = pts
bt = Bootstrapper(ev, out_scale=D, plains=plains, coeffs=plan.chebyshev(), double_hoist=plan.double_hoist)
# oracle intermediates
inter = []
x = Ct(ct0, 0, D)
cc = max(1, int(round(ev.q(0) / (bt.mr * x.scale))))
x = ev.mul_int(x, cc); inter.append(("mul_int", x))
x = Ct(bt.mod_raise(x, top), top, float(ev.q(0))); inter.append(("mod_raise", x))
for i, m in enumerate(bt.cts):
    x = linear_transform(ev, x, m, plains=plains[("cts", i)], double_hoist=plan.double_hoist); inter.append(("cts%d" % i, x))
xc = ev.conj(x)
u_re = ev.add(x, xc); inter.append(("u_re", u_re))
u_im = ev.mul_by_i(ev.sub(x, xc), -1); inter.append(("u_im", u_im))
y_re = eval_mod(ev, u_re, bt.K, bt.r, bt.coeffs); inter.append(("y_re", y_re))
y_im = eval_mod(ev, u_im, bt.K, bt.r, bt.coeffs)
y = ev.add(y_re, ev.mul_by_i(y_im, 1)); inter.append(("y", y))
dev_in = ctx.upload(ct0[None])
for step, (name, want) in enumerate(inter, 1):
    os.environ["LSA_BT_STOP"] = str(step)
    out = plan.run(dev_in, 1, rlk, glk)
    got = ctx.download(out, (1, 2, plan.out_level + 1, N))[0]
    lv = min(want.level, plan.out_level)
    ok = np.array_equal(got[:, : lv + 1], want.data[:, : lv + 1])
    print(step, name, "level", want.level, "MATCH" if ok else "DIFF", [bool(np.array_equal(got[p, j], want.data[p, j])) for p in range(2) for j in range(lv + 1)][:6])
    if not ok:
        zw = c.ckks_decrypt(want.data[:, : lv + 1], want.scale)
        zg = c.ckks_decrypt(got[:, : lv + 1], want.scale)
        print("decrypt want/got agree bits:", mean_precision_bits(zw, zg), "| first slots", zw[:2], zg[:2])
        lvm, n1, ks, _ = plan.matrix(step - 3) if name.startswith("cts") else (None, None, None, None)
        print("device matrix level", lvm, "n1", n1, "ks", ks)
        if ks:
            from oracle.ckks_bootstrap import bsgs_split
            print("oracle n1", bsgs_split(ks, N // 2), "oracle ks", sorted(bt.cts[step - 3]))
        break
