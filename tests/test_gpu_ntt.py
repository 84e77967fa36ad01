"""GPU parity (bit-exact) of the batched NTT/INTT kernels against the CPU oracle — BASELINE config #2 shape
(BFV N=2^14, 4 RNS primes) at an oracle-sized batch, plus every pass shape (single pass / two passes, odd logn)."""
import numpy as np
import pytest

from lattisense_amd import params
from tests.gpu_util import need_gpu, rand_ct

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("logn,wide", [(10, 0), (12, 0), (13, 0), (14, 0), (15, 0), (16, 0), (13, 1), (14, 1)])
def test_ntt_roundtrip_and_parity(logn, wide, monkeypatch):
    """wide = 1: the whole-limb single-pass plan of N = 2^13 / 2^14 (512 / 1024-thread workgroups, LSA_NTT_WIDE=1)."""
    need_gpu()
    monkeypatch.setenv("LSA_NTT_WIDE", str(wide))
    from lattisense_amd.device import DeviceContext, ALGO_CKKS
    from oracle.pyoracle import Oracle
    n = 1 << logn
    B = params.CKKS_BOOTSTRAP_65536
    q, p = B["q"][:3], B["p"][:1]          # 60-, 40-, 40-bit and a 61-bit prime
    mods = q + p
    ctx = DeviceContext(ALGO_CKKS, n, q, p)
    o = Oracle(n, q, p, 0)
    rng = np.random.default_rng(logn)
    batch, polys = 3, 2
    data = rand_ct(rng, mods, polys, n, batch)
    data[0, 0, 0, :4] = [0, mods[0] - 1, 1, mods[0] - 2]
    buf = ctx.upload(data)
    mod_of = list(range(len(mods)))
    ctx.ntt(buf, batch, polys * len(mods), mod_of, inverse=False)
    got = ctx.download(buf, data.shape)
    # oracle on a subset of rows for the big sizes (it is O(N log N) per row in C, fast enough for all here)
    for b in range(batch):
        for pl in range(polys):
            for i in range(len(mods)):
                assert np.array_equal(got[b, pl, i], o.ntt(i, data[b, pl, i])), (b, pl, i)
    ctx.ntt(buf, batch, polys * len(mods), mod_of, inverse=True)
    back = ctx.download(buf, data.shape)
    assert np.array_equal(back, data)


def test_ntt_skip_rows_and_strides():
    need_gpu()
    from lattisense_amd.device import DeviceContext, ALGO_BFV
    from oracle.pyoracle import Oracle
    P = params.BFV_DEFAULT[16384]
    n = 16384
    ctx = DeviceContext(ALGO_BFV, n, P["q"], P["p"], P["t"])
    o = Oracle(n, P["q"], P["p"], P["t"])
    rng = np.random.default_rng(5)
    data = rand_ct(rng, P["q"][:4], 1, n, 2)   # [2][1][4][N]
    buf = ctx.upload(data)
    ctx.ntt(buf, 2, 4, [0, 0xFF, 2, 0xFF], inverse=False)
    got = ctx.download(buf, data.shape)
    for b in range(2):
        assert np.array_equal(got[b, 0, 0], o.ntt(0, data[b, 0, 0]))
        assert np.array_equal(got[b, 0, 1], data[b, 0, 1])
        assert np.array_equal(got[b, 0, 2], o.ntt(2, data[b, 0, 2]))
        assert np.array_equal(got[b, 0, 3], data[b, 0, 3])


def test_bfv_config2_shape_linearity_full_batch():
    """BASELINE configs[1] at full size (batch 1024 cts x 2 polys x 4 limbs, N=2^14): size-independent properties —
    INTT(NTT(x)) == x and NTT(x+y) == NTT(x)+NTT(y) — plus oracle parity on sampled rows."""
    need_gpu()
    from lattisense_amd.device import DeviceContext, ALGO_BFV
    from oracle.pyoracle import Oracle
    P = params.BFV_DEFAULT[16384]
    n, L, batch = 16384, 4, 1024
    q = P["q"][:L]
    ctx = DeviceContext(ALGO_BFV, n, P["q"], P["p"], P["t"])
    o = Oracle(n, P["q"], P["p"], P["t"])
    rng = np.random.default_rng(9)
    x = rand_ct(rng, q, 2, n, batch)
    y = rand_ct(rng, q, 2, n, batch)
    bx, by = ctx.upload(x), ctx.upload(y)
    bs = ctx.addsub(0, L - 1, 2, bx, by, batch)
    rows = 2 * L
    for b in (bx, by, bs):
        ctx.ntt(b, batch, rows, list(range(L)))
    lhs = ctx.download(bs, x.shape)
    bsum = ctx.addsub(0, L - 1, 2, bx, by, batch)
    rhs = ctx.download(bsum, x.shape)
    assert np.array_equal(lhs, rhs)
    fx = ctx.download(bx, x.shape)
    for (b, pl, i) in [(0, 0, 0), (511, 1, 2), (1023, 1, 3)]:
        assert np.array_equal(fx[b, pl, i], o.ntt(i, x[b, pl, i]))
    ctx.ntt(bx, batch, rows, list(range(L)), inverse=True)
    assert np.array_equal(ctx.download(bx, x.shape), x)
