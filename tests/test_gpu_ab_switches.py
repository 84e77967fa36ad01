"""The run-time A/B switches that select an older form of an operator must not change a single
residue: every form is compared with the CPU oracle on the same inputs, in one process (the switches are read per call)."""
import numpy as np
import pytest

from lattisense_amd import params
from tests.gpu_util import need_gpu

pytestmark = pytest.mark.gpu


def _rand(rng, mods, shape, n):
    out = np.empty((*shape, len(mods), n), dtype=np.uint64)
    for i, m in enumerate(mods):
        out[..., i, :] = rng.integers(0, m, size=(*shape, n), dtype=np.uint64)
    return out


@pytest.mark.parametrize("switch", ["LSA_ROT_SCATTER", "LSA_KSMAC_XCD", "LSA_KS_FUSED"])
@pytest.mark.parametrize("logn", [13, 16])
def test_ckks_rotate_and_hmult_with_a_switch_off(switch, logn, monkeypatch):
    """N = 2^13 (7 + 6... staged / one-pass shapes) and N = 2^16 (the radix-16-squared passes, the fused second pass + key MAC)"""
    need_gpu()
    from lattisense_amd.device import ALGO_CKKS, DeviceContext
    from oracle.pyoracle import Oracle
    C = params.CKKS_DEFAULT[65536]
    n = 1 << logn
    q, p = C["q"][:6], C["p"][:2]
    rng = np.random.default_rng(logn)
    o = Oracle(n, q, p, 0)
    lvl, klvl, batch = 4, 5, 2
    A = _rand(rng, q[: lvl + 1], (batch, 2), n)
    B = _rand(rng, q[: lvl + 1], (batch, 2), n)
    beta = (klvl + 1 + len(p) - 1) // len(p)
    key = _rand(rng, q[: klvl + 1] + p, (beta, 2), n)
    g = int(pow(5, 77, 2 * n))
    want_rot = [o.ckks_rotate(lvl, A[b], g, key, klvl) for b in range(batch)]
    want_mul = [o.ckks_mult_relin_rescale(lvl, A[b], B[b], key, klvl) for b in range(batch)]
    for value in ("0", None):          # the older form first, then the default, in the same process
        if value is None:
            monkeypatch.delenv(switch, raising=False)
        else:
            monkeypatch.setenv(switch, value)
        ctx = DeviceContext(ALGO_CKKS, n, q, p)
        k = ctx.upload_key(key, klvl)
        da, db = ctx.upload(A), ctx.upload(B)
        got = ctx.download(ctx.ckks_rotate(lvl, da, g, k, batch), (batch, 2, lvl + 1, n))
        for b in range(batch):
            assert np.array_equal(got[b], want_rot[b]), (switch, value, "rotate")
        got = ctx.download(ctx.ckks_mult_relin_rescale(lvl, da, db, k, batch), (batch, 2, lvl, n))
        for b in range(batch):
            assert np.array_equal(got[b], want_mul[b]), (switch, value, "hmult")
        ctx.close()


def test_bfv_mult_relin_with_the_folded_steps_off(monkeypatch):
    need_gpu()
    from lattisense_amd.device import ALGO_BFV, DeviceContext
    from oracle.pyoracle import Oracle
    F = params.BFV_DEFAULT[16384]
    n, t = 1 << 13, 65537
    q, p = F["q"][:4], F["p"]
    rng = np.random.default_rng(5)
    o = Oracle(n, q, p, t)
    lvl, klvl, batch = 3, 3, 2
    A = _rand(rng, q, (batch, 2), n)
    B = _rand(rng, q, (batch, 2), n)
    beta = (klvl + 1 + len(p) - 1) // len(p)
    key = _rand(rng, q + p, (beta, 2), n)
    want = [o.bfv_mult_relin(lvl, A[b], B[b], key, klvl) for b in range(batch)]
    for value in ("0", None):
        if value is None:
            monkeypatch.delenv("LSA_BFV_FOLD", raising=False)
        else:
            monkeypatch.setenv("LSA_BFV_FOLD", value)
        ctx = DeviceContext(ALGO_BFV, n, q, p, t)
        k = ctx.upload_key(key, klvl)
        got = ctx.download(ctx.bfv_mult_relin(lvl, ctx.upload(A), ctx.upload(B), k, batch), (batch, 2, lvl + 1, n))
        for b in range(batch):
            assert np.array_equal(got[b], want[b]), value
        ctx.close()
