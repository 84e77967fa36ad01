import numpy as np
import pytest


def need_gpu():
    from lattisense_amd._native import lib
    lib()  # fails loudly if the HIP library is not built


def rand_ct(rng, mods, polys, n, batch):
    out = np.empty((batch, polys, len(mods), n), dtype=np.uint64)
    for i, q in enumerate(mods):
        out[:, :, i, :] = rng.integers(0, q, size=(batch, polys, n), dtype=np.uint64)
    return out
