"""GPU parity of the BFV operators against the CPU oracle (bit-exact), and the reference's own assertion on the GPU
output: decrypt == x*y mod t / rotated vector (unittests/test_gpu_bfv.cpp:332-335, :486, :551)."""
import numpy as np
import pytest

from lattisense_amd import params
from tests.gpu_util import need_gpu

pytestmark = pytest.mark.gpu


def _setup(n, nq=4):
    from lattisense_amd.device import DeviceContext, ALGO_BFV
    from oracle.pyoracle import Oracle
    P = params.BFV_DEFAULT[16384]
    q, p, t = P["q"][:nq], P["p"], P["t"]
    return DeviceContext(ALGO_BFV, n, q, p, t), Oracle(n, q, p, t), q, p, t


@pytest.mark.parametrize("n,lvl", [(1024, 3), (4096, 1)])
def test_bfv_mult_relin_bit_exact_and_decrypts(n, lvl):
    need_gpu()
    from oracle.client import Client
    ctx, o, q, p, t = _setup(n)
    assert ctx.moduli == o.mod           # same auxiliary basis on both sides
    c = Client(o, seed=n + 1)
    klvl = 3
    rlk = c.gen_relin_key(klvl)
    rng = np.random.default_rng(0)
    batch = 2
    xs = [rng.integers(0, t, size=n, dtype=np.uint64) for _ in range(batch)]
    ys = [rng.integers(0, t, size=n, dtype=np.uint64) for _ in range(batch)]
    A = np.stack([c.bfv_encrypt(x, lvl) for x in xs])
    Bc = np.stack([c.bfv_encrypt(y, lvl) for y in ys])
    k = ctx.upload_key(rlk, klvl)
    da, db = ctx.upload(A), ctx.upload(Bc)
    d3 = ctx.bfv_mult(lvl, da, db, batch)
    want_d3 = np.stack([o.bfv_mult(lvl, A[i], Bc[i]) for i in range(batch)])
    assert np.array_equal(ctx.download(d3, want_d3.shape), want_d3)
    z = ctx.bfv_relin(lvl, d3, k, batch)
    want_z = np.stack([o.bfv_relin(lvl, want_d3[i], rlk, klvl) for i in range(batch)])
    got = ctx.download(z, want_z.shape)
    assert np.array_equal(got, want_z)
    z2 = ctx.bfv_mult_relin(lvl, da, db, k, batch)
    assert np.array_equal(ctx.download(z2, want_z.shape), want_z)
    # same-operand multiply (single input edge, mega_ag_executors_gpu.cu:178-186)
    sq = ctx.bfv_mult(lvl, da, da, batch)
    want_sq = np.stack([o.bfv_mult(lvl, A[i], A[i]) for i in range(batch)])
    assert np.array_equal(ctx.download(sq, want_sq.shape), want_sq)
    for i in range(batch):
        assert np.array_equal(c.bfv_decrypt(got[i]), xs[i] * ys[i] % np.uint64(t))


def test_bfv_rotate_and_rescale():
    need_gpu()
    from oracle.client import Client, galois_element_for_col_rotation, galois_element_for_row_rotation
    n, lvl, klvl = 2048, 2, 3
    ctx, o, q, p, t = _setup(n)
    c = Client(o, seed=77)
    x = np.arange(n, dtype=np.uint64) % np.uint64(t)
    A = c.bfv_encrypt(x, lvl)[None]
    da = ctx.upload(A)
    h = n // 2
    for step, g in [(1, galois_element_for_col_rotation(1, n)), (33, galois_element_for_col_rotation(33, n)),
                    (None, galois_element_for_row_rotation(n))]:
        glk = c.gen_galois_key(g, klvl)
        k = ctx.upload_key(glk, klvl)
        out = ctx.bfv_rotate(lvl, da, g, k, 1)
        want = o.bfv_rotate(lvl, A[0], g, glk, klvl)[None]
        got = ctx.download(out, want.shape)
        assert np.array_equal(got, want)
        exp = (np.concatenate([x[h:], x[:h]]) if step is None
               else np.concatenate([np.roll(x[:h], -step), np.roll(x[h:], -step)]))
        assert np.array_equal(c.bfv_decrypt(got[0]), exp)
        ctx.destroy_key(k)
    rs = ctx.bfv_rescale(lvl, 2, da, 1)
    want = o.bfv_rescale(lvl, A[0])[None]
    got = ctx.download(rs, want.shape)
    assert np.array_equal(got, want)
    assert np.array_equal(c.bfv_decrypt(got[0]), x)


@pytest.mark.parametrize("tag", ["default_n8192_k1", "default_n16384_k2", "default_n32768_k3"])
def test_bfv_every_level_mult_relin(tag):
    """BFV parameter sets of the reference (fixture.hpp:63-85 + frontend/parameter.json) incl. the 12-limb chain whose
    Q -> QMul extension has 12 source limbs; every level, ring degree shrunk to 1024 for the oracle."""
    need_gpu()
    from lattisense_amd.device import DeviceContext, ALGO_BFV
    from oracle.pyoracle import Oracle
    from tests.gpu_util import rand_ct
    P = params.BFV_DEFAULT[{"default_n8192_k1": 8192, "default_n16384_k2": 16384, "default_n32768_k3": 32768}[tag]]
    q, p, t = P["q"], P["p"], P["t"]
    n = 1024
    ctx = DeviceContext(ALGO_BFV, n, q, p, t)
    o = Oracle(n, q, p, t)
    assert ctx.moduli == o.mod
    rng = np.random.default_rng(len(q))
    klvl = len(q) - 1
    beta = (klvl + 1 + len(p) - 1) // len(p)
    key = np.empty((beta, 2, klvl + 1 + len(p), n), dtype=np.uint64)
    for j, m in enumerate(q + p):
        key[:, :, j, :] = rng.integers(0, m, size=(beta, 2, n), dtype=np.uint64)
    k = ctx.upload_key(key, klvl)
    levels = range(0, len(q)) if len(q) <= 6 else [0, 1, 5, 8, 11]
    for lvl in levels:
        A, Bc = rand_ct(rng, q[: lvl + 1], 2, n, 2), rand_ct(rng, q[: lvl + 1], 2, n, 2)
        out = ctx.bfv_mult_relin(lvl, ctx.upload(A), ctx.upload(Bc), k, 2)
        want = np.stack([o.bfv_mult_relin(lvl, A[i], Bc[i], key, klvl) for i in range(2)])
        assert np.array_equal(ctx.download(out, want.shape), want), (tag, lvl)
