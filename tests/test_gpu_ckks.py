"""GPU parity of the CKKS operators against the CPU oracle: bit-exact on identical inputs (real keys/ciphertexts from
the test client), and the reference's own message-level assertion on the GPU output (mean precision >= 10 bits,
unittests/test_gpu_ckks.cpp:37-43)."""
import numpy as np
import pytest

from lattisense_amd import params
from tests.gpu_util import need_gpu, rand_ct

pytestmark = pytest.mark.gpu


def _setup(n, nq, seed=3):
    from lattisense_amd.device import DeviceContext, ALGO_CKKS
    from oracle.client import Client
    from oracle.pyoracle import Oracle
    P = params.CKKS_DEFAULT[16384]
    q, p = P["q"][:nq], P["p"]
    return DeviceContext(ALGO_CKKS, n, q, p), Oracle(n, q, p, 0), q, p


@pytest.mark.parametrize("n,lvl", [(1024, 4), (4096, 2), (8192, 5)])
def test_mult_relin_rescale_bit_exact_and_decrypts(n, lvl):
    _mult_relin_rescale_case(n, lvl)


def _mult_relin_rescale_case(n, lvl):
    need_gpu()
    from oracle.client import Client, mean_precision_bits
    ctx, o, q, p = _setup(n, 6)
    c = Client(o, seed=n)
    klvl = 5
    rlk = c.gen_relin_key(klvl)
    scale = float(2 ** 34)
    rng = np.random.default_rng(1)
    batch = 3
    xs = [rng.uniform(-1, 1, n // 2) + 1j * rng.uniform(-1, 1, n // 2) for _ in range(batch)]
    ys = [rng.uniform(-1, 1, n // 2) + 1j * rng.uniform(-1, 1, n // 2) for _ in range(batch)]
    A = np.stack([c.ckks_encrypt(x, lvl, scale) for x in xs])
    Bc = np.stack([c.ckks_encrypt(y, lvl, scale) for y in ys])
    k = ctx.upload_key(rlk, klvl)
    da, db = ctx.upload(A), ctx.upload(Bc)
    # step by step
    d3 = ctx.ckks_mult(lvl, da, db, batch)
    want_d3 = np.stack([o.ckks_mult(lvl, A[i], Bc[i]) for i in range(batch)])
    assert np.array_equal(ctx.download(d3, want_d3.shape), want_d3)
    r2 = ctx.ckks_relin(lvl, d3, k, batch)
    want_r2 = np.stack([o.ckks_relin(lvl, want_d3[i], rlk, klvl) for i in range(batch)])
    assert np.array_equal(ctx.download(r2, want_r2.shape), want_r2)
    rs = ctx.ckks_rescale(lvl, 2, r2, batch)
    want_rs = np.stack([o.ckks_rescale(lvl, want_r2[i]) for i in range(batch)])
    assert np.array_equal(ctx.download(rs, want_rs.shape), want_rs)
    # fused entry point, with and without tiling
    for tb in (0, 1, 2):
        ctx.set_tile_batch(tb)
        out = ctx.ckks_mult_relin_rescale(lvl, da, db, k, batch)
        got = ctx.download(out, want_rs.shape)
        assert np.array_equal(got, want_rs)
    for i in range(batch):
        re, im = mean_precision_bits(xs[i] * ys[i], c.ckks_decrypt(got[i], scale * scale / q[lvl]))
        assert re >= 10 and im >= 10
    ctx.destroy_key(k)


def test_rotate_conjugate_bit_exact_and_decrypts():
    need_gpu()
    from oracle.client import (Client, galois_element_for_col_rotation, galois_element_for_row_rotation,
                               mean_precision_bits)
    n, lvl, klvl = 2048, 3, 4
    ctx, o, q, p = _setup(n, 5)
    c = Client(o, seed=21)
    scale = float(2 ** 34)
    rng = np.random.default_rng(2)
    batch = 2
    xs = [rng.uniform(-1, 1, n // 2) + 1j * rng.uniform(-1, 1, n // 2) for _ in range(batch)]
    A = np.stack([c.ckks_encrypt(x, lvl, scale) for x in xs])
    da = ctx.upload(A)
    for step, g in [(1, galois_element_for_col_rotation(1, n)), (-7, galois_element_for_col_rotation(-7, n)),
                    (None, galois_element_for_row_rotation(n))]:
        glk = c.gen_galois_key(g, klvl)
        k = ctx.upload_key(glk, klvl)
        out = ctx.ckks_rotate(lvl, da, g, k, batch)
        want = np.stack([o.ckks_rotate(lvl, A[i], g, glk, klvl) for i in range(batch)])
        got = ctx.download(out, want.shape)
        assert np.array_equal(got, want)
        for i in range(batch):
            exp = np.conj(xs[i]) if step is None else np.roll(xs[i], -step)
            re, im = mean_precision_bits(exp, c.ckks_decrypt(got[i], scale))
            assert re >= 10 and im >= 10
        ctx.destroy_key(k)


def test_add_sub_neg_drop_level():
    need_gpu()
    n, lvl = 1024, 3
    ctx, o, q, p = _setup(n, 4)
    rng = np.random.default_rng(4)
    batch = 2
    A, Bc = rand_ct(rng, q, 2, n, batch), rand_ct(rng, q, 2, n, batch)
    da, db = ctx.upload(A), ctx.upload(Bc)
    for op, name in [(0, "add"), (1, "sub"), (2, "neg")]:
        out = ctx.addsub(op, lvl, 2, da, db if op != 2 else None, batch)
        got = ctx.download(out, A.shape)
        for i in range(lvl + 1):
            want = np.stack([[o.vec(name, i, A[b, pl, i], Bc[b, pl, i]) for pl in range(2)] for b in range(batch)])
            assert np.array_equal(got[:, :, i], want)
    out = ctx.drop_level(lvl, 2, da, batch)
    assert np.array_equal(ctx.download(out, (batch, 2, lvl, n)), A[:, :, :lvl])


def test_deep_chain_keyswitch_bootstrap_primes():
    """digit width 5, 61-bit special primes, 60/40/39-bit chain (frontend/custom_task.py:387-420), L=12: beta=3 with a
    short last digit; uniform-random 'key' and ciphertext (parity does not need a decryptable key)."""
    need_gpu()
    from lattisense_amd.device import DeviceContext, ALGO_CKKS
    from oracle.pyoracle import Oracle
    Bp = params.CKKS_BOOTSTRAP_65536
    n, nq = 2048, 12
    q, p = Bp["q"][:nq], Bp["p"]
    lvl = klvl = nq - 1
    ctx = DeviceContext(ALGO_CKKS, n, q, p)
    o = Oracle(n, q, p, 0)
    rng = np.random.default_rng(6)
    batch = 2
    A, Bc = rand_ct(rng, q, 2, n, batch), rand_ct(rng, q, 2, n, batch)
    beta = (lvl + 1 + len(p) - 1) // len(p)
    key = np.empty((beta, 2, lvl + 1 + len(p), n), dtype=np.uint64)
    for j, m in enumerate(q + p):
        key[:, :, j, :] = rng.integers(0, m, size=(beta, 2, n), dtype=np.uint64)
    k = ctx.upload_key(key, klvl)
    out = ctx.ckks_mult_relin_rescale(lvl, ctx.upload(A), ctx.upload(Bc), k, batch)
    want = np.stack([o.ckks_mult_relin_rescale(lvl, A[i], Bc[i], key, klvl) for i in range(batch)])
    assert np.array_equal(ctx.download(out, want.shape), want)


def test_n17_deep_chain_hmult_bit_exact():
    """BASELINE configs[4] shape at full size: N=2^17, 25 Q-limbs + 5 special primes (generated chain, the reference has
    none for this degree), HMult+relin+rescale of one ciphertext pair against the oracle; also exercises the 8+9 stage
    pass split and limb mixes of both butterfly engines."""
    need_gpu()
    from lattisense_amd.device import DeviceContext, ALGO_CKKS
    from oracle.pyoracle import Oracle
    C = params.ckks_n17_chain()
    n, q, p = C["n"], C["q"], C["p"]
    lvl = klvl = len(q) - 1
    ctx = DeviceContext(ALGO_CKKS, n, q, p)
    o = Oracle(n, q, p, 0)
    rng = np.random.default_rng(17)
    A, Bc = rand_ct(rng, q, 2, n, 1), rand_ct(rng, q, 2, n, 1)
    beta = (lvl + 1 + len(p) - 1) // len(p)
    key = np.empty((beta, 2, lvl + 1 + len(p), n), dtype=np.uint64)
    for j, m in enumerate(q + p):
        key[:, :, j, :] = rng.integers(0, m, size=(beta, 2, n), dtype=np.uint64)
    k = ctx.upload_key(key, klvl)
    out = ctx.ckks_mult_relin_rescale(lvl, ctx.upload(A), ctx.upload(Bc), k, 1)
    want = o.ckks_mult_relin_rescale(lvl, A[0], Bc[0], key, klvl)[None]
    assert np.array_equal(ctx.download(out, want.shape), want)


# parameter sets of the reference's GPU tests (unittests/fixture.hpp:87-118, test_gpu_ckks.py:39-56): default chains and
# the custom N=8192 chain with ONE special prime (digit width 1: every digit is a single limb)
_CUSTOM_8192 = dict(q=[0x1FFFEC001, 0x3FFF4001, 0x3FFE8001, 0x40020001, 0x40038001, 0x3FFC0001], p=[0x800004001])


@pytest.mark.parametrize("tag", ["default_n8192", "custom_n8192_k1", "default_n16384_k2"])
def test_every_level_mult_relin_rescale_and_square(tag):
    """The reference generates one task per level (conftest.py:24-69); here every level >= 1 of each parameter set runs
    HMult+relin+rescale and a same-operand multiply (executors_gpu.cu:178-186) against the oracle, with ring degree
    shrunk to 1024 so the oracle stays fast (the prime chains, digit structure and level logic are the real ones)."""
    need_gpu()
    from lattisense_amd.device import DeviceContext, ALGO_CKKS
    from oracle.pyoracle import Oracle
    if tag == "default_n8192":
        P = params.CKKS_DEFAULT[8192]
        q, p = P["q"], P["p"]
    elif tag == "custom_n8192_k1":
        q, p = _CUSTOM_8192["q"], _CUSTOM_8192["p"]
    else:
        P = params.CKKS_DEFAULT[16384]
        q, p = P["q"], P["p"]
    n = 1024
    ctx = DeviceContext(ALGO_CKKS, n, q, p)
    o = Oracle(n, q, p, 0)
    rng = np.random.default_rng(len(q))
    klvl = len(q) - 1
    beta = (klvl + 1 + len(p) - 1) // len(p)
    key = np.empty((beta, 2, klvl + 1 + len(p), n), dtype=np.uint64)
    for j, m in enumerate(q + p):
        key[:, :, j, :] = rng.integers(0, m, size=(beta, 2, n), dtype=np.uint64)
    k = ctx.upload_key(key, klvl)     # one key at the top level serves every lower level
    batch = 2
    for lvl in range(1, len(q)):
        A, Bc = rand_ct(rng, q[: lvl + 1], 2, n, batch), rand_ct(rng, q[: lvl + 1], 2, n, batch)
        da, db = ctx.upload(A), ctx.upload(Bc)
        out = ctx.ckks_mult_relin_rescale(lvl, da, db, k, batch)
        want = np.stack([o.ckks_mult_relin_rescale(lvl, A[i], Bc[i], key, klvl) for i in range(batch)])
        assert np.array_equal(ctx.download(out, want.shape), want), (tag, lvl)
        sq = ctx.ckks_mult(lvl, da, da, batch)
        want_sq = np.stack([o.ckks_mult(lvl, A[i], A[i]) for i in range(batch)])
        assert np.array_equal(ctx.download(sq, want_sq.shape), want_sq), (tag, lvl)
        # unfused tails must give the same residues as the fused default
        from lattisense_amd._native import check, lib
        check(lib().lsa_set_fuse_tails(ctx.h, 0))
        out2 = ctx.ckks_mult_relin_rescale(lvl, da, db, k, batch)
        assert np.array_equal(ctx.download(out2, want.shape), want), (tag, lvl, "unfused")
        check(lib().lsa_set_fuse_tails(ctx.h, 1))
    # level 0 ciphertexts still multiply and relinearise (no rescale possible)
    A, Bc = rand_ct(rng, q[:1], 2, n, batch), rand_ct(rng, q[:1], 2, n, batch)
    d3 = ctx.ckks_mult(0, ctx.upload(A), ctx.upload(Bc), batch)
    r2 = ctx.ckks_relin(0, d3, k, batch)
    want = np.stack([o.ckks_relin(0, o.ckks_mult(0, A[i], Bc[i]), key, klvl) for i in range(batch)])
    assert np.array_equal(ctx.download(r2, want.shape), want)


def test_engine_and_stream_variants_agree():
    """integer vs FP64 butterfly engine, single vs dual stream: identical outputs on the headline operator"""
    need_gpu()
    from lattisense_amd._native import check, lib
    from lattisense_amd.device import DeviceContext, ALGO_CKKS
    P = params.CKKS_DEFAULT[65536]
    n, q, p = 8192, P["q"][:7], P["p"]
    lvl = 6
    ctx = DeviceContext(ALGO_CKKS, n, q, p)
    rng = np.random.default_rng(77)
    batch = 6
    A, Bc = rand_ct(rng, q, 2, n, batch), rand_ct(rng, q, 2, n, batch)
    beta = (lvl + 1 + len(p) - 1) // len(p)
    key = np.empty((beta, 2, lvl + 1 + len(p), n), dtype=np.uint64)
    for j, m in enumerate(q + p):
        key[:, :, j, :] = rng.integers(0, m, size=(beta, 2, n), dtype=np.uint64)
    k = ctx.upload_key(key, lvl)
    da, db = ctx.upload(A), ctx.upload(Bc)
    ref = ctx.download(ctx.ckks_mult_relin_rescale(lvl, da, db, k, batch), (batch, 2, lvl, n))
    for fp64, dual, tile in [(0, 0, 0), (1, 1, 2), (0, 1, 4), (1, 0, 1)]:
        ctx.set_fp64_ntt(fp64)
        check(lib().lsa_set_dual_stream(ctx.h, dual))
        ctx.set_tile_batch(tile)
        got = ctx.download(ctx.ckks_mult_relin_rescale(lvl, da, db, k, batch), (batch, 2, lvl, n))
        assert np.array_equal(got, ref), (fp64, dual, tile)


def test_headline_shape_full_size():
    """BASELINE configs[2] at full size (N=2^16, 13 Q + 4 P limbs): the fused operator against the oracle on one ciphertext,
    against the three separate operators, across butterfly engines / fused tails / operator tiles, and independent of the
    position in the batch."""
    need_gpu()
    from lattisense_amd._native import check, lib
    from lattisense_amd.device import DeviceContext, ALGO_CKKS
    from oracle.pyoracle import Oracle
    P = params.CKKS_DEFAULT[65536]
    n, q, p = 65536, P["q"][:13], P["p"]
    lvl = 12
    ctx = DeviceContext(ALGO_CKKS, n, q, p)
    rng = np.random.default_rng(2026)
    batch = 5
    A, Bc = rand_ct(rng, q, 2, n, batch), rand_ct(rng, q, 2, n, batch)
    A[3], Bc[3] = A[0], Bc[0]                      # the same ciphertexts at two batch positions
    beta = (lvl + 1 + len(p) - 1) // len(p)
    key = np.empty((beta, 2, lvl + 1 + len(p), n), dtype=np.uint64)
    for j, m in enumerate(q + p):
        key[:, :, j, :] = rng.integers(0, m, size=(beta, 2, n), dtype=np.uint64)
    k = ctx.upload_key(key, lvl)
    da, db = ctx.upload(A), ctx.upload(Bc)
    ref = ctx.download(ctx.ckks_mult_relin_rescale(lvl, da, db, k, batch), (batch, 2, lvl, n))
    assert np.array_equal(ref[0], ref[3])
    o = Oracle(n, q, p, 0)
    assert np.array_equal(ref[1], o.ckks_mult_relin_rescale(lvl, A[1], Bc[1], key, lvl))
    d3 = ctx.ckks_mult(lvl, da, db, batch)
    r2 = ctx.ckks_relin(lvl, d3, k, batch)
    sep = ctx.download(ctx.ckks_rescale(lvl, 2, r2, batch), (batch, 2, lvl, n))
    assert np.array_equal(sep, ref)                # merged ModDown+rescale tail == the two tails one after the other
    for fp64, fuse, tile in [(0, 1, 2), (1, 0, 5), (0, 0, 1)]:
        ctx.set_fp64_ntt(fp64)
        check(lib().lsa_set_fuse_tails(ctx.h, fuse))
        ctx.set_tile_batch(tile)
        got = ctx.download(ctx.ckks_mult_relin_rescale(lvl, da, db, k, batch), (batch, 2, lvl, n))
        assert np.array_equal(got, ref), (fp64, fuse, tile)


def test_rotate_headline_shape_full_size():
    """BASELINE configs[3] per-GPU shape at full size (CKKS N=2^16, 13 Q + 4 P limbs, Galois key switch): rotation by
    5^1 mod 2N (the element bench.py --workload rotate times) and conjugation (2N-1) against the oracle on one ciphertext,
    independent of the position in the batch, equal to the hoisted many-rotation entry point, and across butterfly engines /
    fused tails / operator tiles.  Uniform-random key: parity does not need a decryptable one."""
    need_gpu()
    from lattisense_amd._native import check, lib
    from lattisense_amd.device import DeviceContext, ALGO_CKKS
    from oracle.pyoracle import Oracle
    P = params.CKKS_DEFAULT[65536]
    n, q, p = 65536, P["q"][:13], P["p"]
    lvl = 12
    ctx = DeviceContext(ALGO_CKKS, n, q, p)
    o = Oracle(n, q, p, 0)
    rng = np.random.default_rng(2027)
    batch = 4
    A = rand_ct(rng, q, 2, n, batch)
    A[2] = A[0]                                    # the same ciphertext at two batch positions
    da = ctx.upload(A)
    beta = (lvl + 1 + len(p) - 1) // len(p)
    keys, raw = {}, {}
    for g in (5, 2 * n - 1):
        key = np.empty((beta, 2, lvl + 1 + len(p), n), dtype=np.uint64)
        for j, m in enumerate(q + p):
            key[:, :, j, :] = rng.integers(0, m, size=(beta, 2, n), dtype=np.uint64)
        raw[g] = key
        keys[g] = ctx.upload_key(key, lvl)
    refs = {}
    for g in keys:
        ref = ctx.download(ctx.ckks_rotate(lvl, da, g, keys[g], batch), (batch, 2, lvl + 1, n))
        assert np.array_equal(ref[0], ref[2])
        assert np.array_equal(ref[1], o.ckks_rotate(lvl, A[1], g, raw[g], lvl)), g
        refs[g] = ref
    outs = ctx.ckks_rotate_many(lvl, da, keys, batch)
    for g in keys:
        assert np.array_equal(ctx.download(outs[g], (batch, 2, lvl + 1, n)), refs[g]), g
    for fp64, fuse, tile in [(0, 1, 2), (1, 0, 3), (0, 0, 1)]:
        ctx.set_fp64_ntt(fp64)
        check(lib().lsa_set_fuse_tails(ctx.h, fuse))
        ctx.set_tile_batch(tile)
        got = ctx.download(ctx.ckks_rotate(lvl, da, 5, keys[5], batch), (batch, 2, lvl + 1, n))
        assert np.array_equal(got, refs[5]), (fp64, fuse, tile)


def test_hoisted_rotations_equal_stand_alone_rotations():
    """lsa_ckks_rotate_many: one decomposition of the input for several Galois elements; with the automorphism applied after
    the key switch every output is bit-identical to lsa_ckks_rotate (and hence to the oracle)."""
    need_gpu()
    from lattisense_amd.device import DeviceContext, ALGO_CKKS
    P = params.CKKS_DEFAULT[65536]
    n, q, p = 8192, P["q"][:6], P["p"][:2]
    lvl, klvl = 3, 5
    ctx = DeviceContext(ALGO_CKKS, n, q, p)
    rng = np.random.default_rng(91)
    batch = 3
    A = rand_ct(rng, q[: lvl + 1], 2, n, batch)
    da = ctx.upload(A)
    beta = (klvl + 1 + len(p) - 1) // len(p)
    keys, raw = {}, {}
    for g in (5, pow(5, 77, 2 * n), 2 * n - 1):
        key = np.empty((beta, 2, klvl + 1 + len(p), n), dtype=np.uint64)
        for j, m in enumerate(q + p):
            key[:, :, j, :] = rng.integers(0, m, size=(beta, 2, n), dtype=np.uint64)
        raw[g] = key
        keys[g] = ctx.upload_key(key, klvl)
    outs = ctx.ckks_rotate_many(lvl, da, keys, batch)
    from oracle.pyoracle import Oracle
    o = Oracle(n, q, p, 0)
    for g in keys:
        got = ctx.download(outs[g], (batch, 2, lvl + 1, n))
        single = ctx.download(ctx.ckks_rotate(lvl, da, g, keys[g], batch), (batch, 2, lvl + 1, n))
        assert np.array_equal(got, single)
        assert np.array_equal(got[1], o.ckks_rotate(lvl, A[1], g, raw[g], klvl))
