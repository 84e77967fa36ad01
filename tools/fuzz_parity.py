"""Randomised GPU-vs-oracle differential run over parameter shapes the fixed tests do not enumerate: ring degrees 2^12..2^16,
1..4 special primes, prime chains cut from the reference's CKKS / BFV / bootstrap sets (46-, 56-, 40-, 60-, 61-bit limbs mixed),
random levels and key levels, edge-value inputs (zeros, q-1 everywhere).  Every comparison is bit-exact.
usage: python tools/fuzz_parity.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lattisense_amd import params
from lattisense_amd.device import ALGO_BFV, ALGO_CKKS, DeviceContext
from oracle.pyoracle import Oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
C = params.CKKS_DEFAULT[65536]
B = params.CKKS_BOOTSTRAP_65536
F = params.BFV_DEFAULT[32768] if 32768 in params.BFV_DEFAULT else params.BFV_DEFAULT[16384]
pool_q = C["q"][:8] + B["q"][:4] + B["q"][13:17] + B["q"][21:24]
pool_p = C["p"] + B["p"]


def rand_limbs(mods, shape, n, mode):
    out = np.empty((*shape, len(mods), n), dtype=np.uint64)
    for i, m in enumerate(mods):
        if mode == 0:
            out[..., i, :] = rng.integers(0, m, size=(*shape, n), dtype=np.uint64)
        elif mode == 1:
            out[..., i, :] = m - 1
        else:
            out[..., i, :] = 0
    return out


t0 = time.time()
done = {"ckks_hmult": 0, "ckks_rotate": 0, "bfv_hmult": 0, "bfv_rotate": 0}
for case in range(cases):
    bfv = case % 3 == 2
    logn = int(rng.integers(12, 16 if bfv else 17))   # CKKS: up to 2^16 (the chains' primes are 1 mod 2^17), BFV: up to 2^15
    n = 1 << logn
    nq = int(rng.integers(2, 9))
    np_ = int(rng.integers(1, 5))
    if bfv:
        q = [int(x) for x in rng.permutation(F["q"])[: min(nq, len(F["q"]))]]
        p = [int(x) for x in rng.permutation(F["p"] + B["p"][:2])[:np_]]
        t = 65537
    else:
        q = [int(x) for x in rng.permutation(pool_q)[:nq]]
        p = [int(x) for x in rng.permutation(pool_p)[:np_]]
        t = 0
    nq = len(q)
    o = Oracle(n, q, p, t)
    # the NTT plan switch is read when the context is built: unset = per launch, 1 = whole-limb single pass at 2^13 / 2^14
    os.environ.pop("LSA_NTT_WIDE", None)
    if case % 2:
        os.environ["LSA_NTT_WIDE"] = "1"
    ctx = DeviceContext(ALGO_BFV if bfv else ALGO_CKKS, n, q, p, t)
    klvl = nq - 1
    lvl = int(rng.integers(1, nq))          # >= 1 (rescale needs a level)
    mode = case % 7 if case % 7 < 3 else 0  # mostly random values, sometimes q-1 / zero
    batch = int(rng.integers(1, 4))
    A = rand_limbs(q[: lvl + 1], (batch, 2), n, mode)
    Bc = rand_limbs(q[: lvl + 1], (batch, 2), n, 0)
    beta = (klvl + 1 + len(p) - 1) // len(p)
    key = rand_limbs(q[: klvl + 1] + p, (beta, 2), n, 0)
    k = ctx.upload_key(key, klvl)
    da, db = ctx.upload(A), ctx.upload(Bc)
    g = int(pow(5, int(rng.integers(1, n // 2)), 2 * n)) if case % 2 else 2 * n - 1
    if bfv:
        got = ctx.download(ctx.bfv_mult_relin(lvl, da, db, k, batch), (batch, 2, lvl + 1, n))
        for b in range(batch):
            assert np.array_equal(got[b], o.bfv_mult_relin(lvl, A[b], Bc[b], key, klvl)), ("bfv_hmult", case, n, q, p, lvl)
        done["bfv_hmult"] += batch
        got = ctx.download(ctx.bfv_rotate(lvl, da, g, k, batch), (batch, 2, lvl + 1, n))
        for b in range(batch):
            assert np.array_equal(got[b], o.bfv_rotate(lvl, A[b], g, key, klvl)), ("bfv_rotate", case, n, q, p, lvl, g)
        done["bfv_rotate"] += batch
    else:
        got = ctx.download(ctx.ckks_mult_relin_rescale(lvl, da, db, k, batch), (batch, 2, lvl, n))
        for b in range(batch):
            assert np.array_equal(got[b], o.ckks_mult_relin_rescale(lvl, A[b], Bc[b], key, klvl)), ("ckks_hmult", case, n, q, p, lvl)
        done["ckks_hmult"] += batch
        got = ctx.download(ctx.ckks_rotate(lvl, da, g, k, batch), (batch, 2, lvl + 1, n))
        for b in range(batch):
            assert np.array_equal(got[b], o.ckks_rotate(lvl, A[b], g, key, klvl)), ("ckks_rotate", case, n, q, p, lvl, g)
        done["ckks_rotate"] += batch
    ctx.close()
print("fuzz_parity: %d cases, all bit-exact: %s  (%.1f s)" % (cases, done, time.time() - t0))
