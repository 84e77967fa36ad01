"""Mean precision of a device bootstrap against the ring degree, same parameters otherwise (reference bootstrap chain 25Q+5P,
dense main secret of weight 192 behind the sparse-secret encapsulation, weight-32 ephemeral secret, scale 2^40): how much of the
19.7 -> 13.9-bit drop between the toy and the full-size test is the ring degree itself.  Message-level only (no oracle walk)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lattisense_amd import params
from lattisense_amd.device import ALGO_CKKS, BootstrapPlan, DeviceContext
from oracle.client import Client, mean_precision_bits
from oracle.pyoracle import Oracle

B = params.CKKS_BOOTSTRAP_65536
for log_n in [int(x) for x in (sys.argv[1:] or ["10", "11", "12", "13"])]:
    N = 1 << log_n
    t0 = time.time()
    o = Oracle(N, B["q"], B["p"], 0)
    c = Client(o, seed=5, hamming=min(192, N // 4))
    sparse = Client(o, seed=6, hamming=32)
    ctx = DeviceContext(ALGO_CKKS, N, B["q"], B["p"])
    top = len(B["q"]) - 1
    D = float(2 ** 40)
    plan = BootstrapPlan(ctx, in_scale=D, out_scale=D)
    rlk = ctx.upload_key(c.gen_relin_key(top), top)
    glk = {e: ctx.upload_key(c.gen_galois_key(e, top), top) for e in plan.galois_elements}
    kd = ctx.upload_key(c.gen_switching_key(c.s_ntt, sparse.s_ntt, 0), 0)
    ks = ctx.upload_key(c.gen_switching_key(sparse.s_ntt, c.s_ntt, top), top)
    rng = np.random.default_rng(log_n)
    z = rng.uniform(-1, 1, N // 2) + 1j * rng.uniform(-1, 1, N // 2)
    ct = np.stack([c.ckks_encrypt(z, 0, D)])
    out = plan.run(ctx.upload(ct), 1, rlk, glk, kd, ks)
    got = ctx.download(out, (1, 2, plan.out_level + 1, N))
    re, im = mean_precision_bits(z, c.ckks_decrypt(got[0], D))
    print("N=2^%d: mean precision %.2f / %.2f bits (real / imaginary), %d Galois keys, %.0f s" % (log_n, re, im, len(glk), time.time() - t0), flush=True)
    plan.close()
