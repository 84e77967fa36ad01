"""Applies the reference's own assertion to the oracle: decrypt(op(encrypt x)) == plain op.
BFV: exact equality mod t (unittests/test_gpu_bfv.cpp:332-335, rotations :486,:551 with
fhe_ops_lib/utils.cpp:171-190 semantics); CKKS: mean precision >= 10 bits (test_gpu_ckks.cpp:37-43)."""
import numpy as np
import pytest

from lattisense_amd import params
from oracle.client import (Client, galois_element_for_col_rotation, galois_element_for_row_rotation,
                           mean_precision_bits)
from oracle.pyoracle import Oracle

N = 1024


def _bfv():
    P = params.BFV_DEFAULT[16384]
    o = Oracle(N, P["q"][:4], P["p"], P["t"])
    return o, Client(o, seed=7)


def _ckks(nq=5):
    P = params.CKKS_DEFAULT[16384]
    o = Oracle(N, P["q"][:nq], P["p"], 0)
    return o, Client(o, seed=11)


def vec_rotate_col(x, step):
    h = len(x) // 2
    return np.concatenate([np.roll(x[:h], -step), np.roll(x[h:], -step)])


def vec_rotate_row(x):
    h = len(x) // 2
    return np.concatenate([x[h:], x[:h]])


def test_bfv_encrypt_decrypt_roundtrip():
    o, c = _bfv()
    x = np.arange(N, dtype=np.uint64) * 37 % o.t
    assert np.array_equal(c.bfv_decrypt(c.bfv_encrypt(x, 3)), x)


@pytest.mark.parametrize("lvl", [1, 3])
def test_bfv_mult_relin(lvl):
    o, c = _bfv()
    rng = np.random.default_rng(0)
    x = rng.integers(0, o.t, size=N, dtype=np.uint64)
    y = rng.integers(0, o.t, size=N, dtype=np.uint64)
    rlk = c.gen_relin_key(3)
    cx, cy = c.bfv_encrypt(x, lvl), c.bfv_encrypt(y, lvl)
    d3 = o.bfv_mult(lvl, cx, cy)
    want = x * y % np.uint64(o.t)
    assert np.array_equal(c.bfv_decrypt(d3), want)          # ct3 is a legal output (test_gpu_bfv.cpp:288-312)
    z = o.bfv_relin(lvl, d3, rlk, 3)
    assert np.array_equal(c.bfv_decrypt(z), want)
    assert np.array_equal(o.bfv_mult_relin(lvl, cx, cy, rlk, 3), z)


def test_bfv_rotate_col_and_row():
    o, c = _bfv()
    lvl = 2
    x = np.arange(N, dtype=np.uint64)
    cx = c.bfv_encrypt(x, lvl)
    for step in (1, 5, N // 2 - 3):
        g = galois_element_for_col_rotation(step, N)
        glk = c.gen_galois_key(g, 3)
        z = o.bfv_rotate(lvl, cx, g, glk, 3)
        assert np.array_equal(c.bfv_decrypt(z), vec_rotate_col(x, step))
    g = galois_element_for_row_rotation(N)
    glk = c.gen_galois_key(g, 3)
    assert np.array_equal(c.bfv_decrypt(o.bfv_rotate(lvl, cx, g, glk, 3)), vec_rotate_row(x))


def test_bfv_rescale_keeps_message():
    o, c = _bfv()
    x = np.arange(N, dtype=np.uint64) * 3 % o.t
    z = o.bfv_rescale(3, c.bfv_encrypt(x, 3))
    assert z.shape == (2, 3, N)
    assert np.array_equal(c.bfv_decrypt(z), x)


def test_ckks_mult_relin_rescale_precision():
    o, c = _ckks()
    lvl = 4
    scale = float(2 ** 34)
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, N // 2) + 1j * rng.uniform(-1, 1, N // 2)
    y = rng.uniform(-1, 1, N // 2) + 1j * rng.uniform(-1, 1, N // 2)
    rlk = c.gen_relin_key(lvl)
    cx, cy = c.ckks_encrypt(x, lvl, scale), c.ckks_encrypt(y, lvl, scale)
    re, im = mean_precision_bits(x, c.ckks_decrypt(cx, scale))
    assert re >= 10 and im >= 10
    z = o.ckks_mult_relin_rescale(lvl, cx, cy, rlk, lvl)
    assert z.shape == (2, lvl, N)
    got = c.ckks_decrypt(z, scale * scale / o.q[lvl])
    re, im = mean_precision_bits(x * y, got)
    assert re >= 10 and im >= 10
    # a key exported at a higher level serves lower levels (custom_task.py:1266-1269)
    z2 = o.ckks_mult_relin_rescale(2, np.ascontiguousarray(cx[:, :3]), np.ascontiguousarray(cy[:, :3]), rlk, lvl)
    got2 = c.ckks_decrypt(z2, scale * scale / o.q[2])
    re, im = mean_precision_bits(x * y, got2)
    assert re >= 10 and im >= 10


def test_ckks_rotate_and_conjugate():
    o, c = _ckks()
    lvl = 3
    scale = float(2 ** 34)
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, N // 2) + 1j * rng.uniform(-1, 1, N // 2)
    cx = c.ckks_encrypt(x, lvl, scale)
    for step in (1, 20, -7):
        g = galois_element_for_col_rotation(step, N)
        glk = c.gen_galois_key(g, 4)
        z = o.ckks_rotate(lvl, cx, g, glk, 4)
        re, im = mean_precision_bits(np.roll(x, -step), c.ckks_decrypt(z, scale))
        assert re >= 10 and im >= 10
    g = galois_element_for_row_rotation(N)
    glk = c.gen_galois_key(g, 4)
    z = o.ckks_rotate(lvl, cx, g, glk, 4)
    re, im = mean_precision_bits(np.conj(x), c.ckks_decrypt(z, scale))
    assert re >= 10 and im >= 10
