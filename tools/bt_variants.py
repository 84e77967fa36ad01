import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from lattisense_amd import params
from lattisense_amd.device import ALGO_CKKS, DeviceContext, BootstrapPlan
from oracle.client import Client, mean_precision_bits
from oracle.pyoracle import Oracle
from oracle.ckks_bootstrap import Bootstrapper, SparseBootstrapper, Ct, Evaluator
B = params.CKKS_BOOTSTRAP_65536
for logn, cd, sd, K, r, ls in [(10, 3, 3, 16, 3, 0), (10, 2, 2, 16, 3, 0), (11, 4, 3, 12, 2, 0), (12, 3, 2, 16, 3, 8), (10, 4, 3, 16, 3, 5), (11, 2, 3, 16, 3, 7)]:
    N = 1 << logn
    o = Oracle(N, B["q"], B["p"], 0); c = Client(o, seed=logn + cd, hamming=32)
    ctx = DeviceContext(ALGO_CKKS, N, B["q"], B["p"])
    top = len(B["q"]) - 1; D = float(2 ** 40)
    try:
        plan = BootstrapPlan(ctx, cd, sd, K, r, 256.0, D, D, log_slots=ls)
        ev = Evaluator(o, c, top)
        keys = {e: c.gen_galois_key(e, top) for e in plan.galois_elements}
        rlk = ctx.upload_key(ev.rlk, top); glk = {e: ctx.upload_key(k, top) for e, k in keys.items()}
        probe = c.ckks_encrypt(np.zeros(N // 2), 0, D)
        plan.run(ctx.upload(probe[None]), 1, rlk, glk).free()
    except Exception as e:
        print((logn, cd, sd, K, r, ls), "REFUSED:", str(e)[:110]); ctx.close(); continue
    ev = Evaluator(o, c, top)
    keys = {e: c.gen_galois_key(e, top) for e in plan.galois_elements}
    ev.glk = dict(keys)
    rlk = ctx.upload_key(ev.rlk, top); glk = {e: ctx.upload_key(k, top) for e, k in keys.items()}
    ns = (1 << ls) if ls else N // 2
    rng = np.random.default_rng(logn)
    z = rng.uniform(-1, 1, ns) + 1j * rng.uniform(-1, 1, ns)
    ct = c.ckks_encrypt(np.tile(z, (N // 2) // ns), 0, D)
    got = ctx.download(plan.run(ctx.upload(ct[None]), 1, rlk, glk), (1, 2, plan.out_level + 1, N))[0]
    if plan.sparse:
        bt = SparseBootstrapper(ev, ls, cd, sd, K, r, 256.0, out_scale=D, plains=plan.oracle_plains(), coeffs=plan.chebyshev(), double_hoist=plan.double_hoist)
    else:
        bt = Bootstrapper(ev, cd, sd, K, r, 256.0, out_scale=D, plains=plan.oracle_plains(), coeffs=plan.chebyshev(), double_hoist=plan.double_hoist)
    want = bt.bootstrap(Ct(ct, 0, D), top)
    prec = mean_precision_bits(z, c.ckks_decrypt(got, D)[:ns])
    print((logn, cd, sd, K, r, ls), "out level", plan.out_level, "bit-exact", bool(np.array_equal(got, want.data)), "prec %.1f/%.1f" % prec, "keys", len(keys))
    plan.close(); ctx.close()
