"""world_size-2 gloo test of the N>1 path: key broadcast from the ingesting rank, disjoint covering shards, no
steady-state collective, max-over-ranks timing.  The per-rank compute is the CPU oracle here (this is a CPU test of the
distributed plumbing; on the GPU box bench.py runs the same helpers with backend nccl = RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lattisense_amd import params
from lattisense_amd.sharding import (barrier, broadcast_key, init_process_group, max_over_ranks, shard_bounds,
                                     tensor_digest)


def test_shard_bounds_cover_and_are_disjoint():
    for n in (0, 1, 7, 256, 2048, 1001):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                a, b = shard_bounds(n, r, world)
                assert 0 <= a <= b <= n
                seen += list(range(a, b))
            assert seen == list(range(n))
            sizes = [shard_bounds(n, r, world)[1] - shard_bounds(n, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    init_process_group("gloo", rank, world)
    from oracle.client import Client
    from oracle.pyoracle import Oracle
    n, lvl = 1024, 2
    P = params.CKKS_DEFAULT[16384]
    q, p = P["q"][:3], P["p"]
    o = Oracle(n, q, p, 0)
    # rank 0 ingests the relinearisation key; the others receive it
    shape = ((lvl + 1 + len(p) - 1) // len(p), 2, lvl + 1 + len(p), n)
    if rank == 0:
        c = Client(o, seed=3)
        key = torch.from_numpy(c.gen_relin_key(lvl).view(np.int64))
    else:
        key = torch.zeros(shape, dtype=torch.int64)
    broadcast_key(key, src=0)
    digests = [None] * world
    dist.all_gather_object(digests, tensor_digest(key))
    assert len(set(digests)) == 1
    # the global batch is sharded by index; every rank derives the same synthetic ciphertext for a global index
    batch = 5
    a, b = shard_bounds(batch, rank, world)
    rlk = key.numpy().view(np.uint64)

    def ct(i, salt):
        rng = np.random.default_rng(1000 * salt + i)
        return np.stack([np.stack([rng.integers(0, q[j], size=n, dtype=np.uint64) for j in range(lvl + 1)])
                         for _ in range(2)])

    mine = {i: o.ckks_mult_relin_rescale(lvl, ct(i, 1), ct(i, 2), rlk, lvl) for i in range(a, b)}
    barrier()
    t = max_over_ranks(0.001 * (rank + 1))
    assert abs(t - 0.001 * world) < 1e-9
    gathered = [None] * world
    dist.all_gather_object(gathered, {i: tensor_digest(torch.from_numpy(v.view(np.int64))) for i, v in mine.items()})
    if rank == 0:
        merged = {}
        for gth in gathered:
            assert not (set(gth) & set(merged))
            merged.update(gth)
        assert sorted(merged) == list(range(batch))
        # same result as an unsharded run
        ref = {i: tensor_digest(torch.from_numpy(o.ckks_mult_relin_rescale(lvl, ct(i, 1), ct(i, 2), rlk, lvl).view(np.int64)))
               for i in range(batch)}
        assert merged == ref
        out.put("ok")
    dist.destroy_process_group()


def test_two_rank_key_broadcast_and_sharded_run():
    from oracle import pyoracle
    pyoracle.build()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert out.get(timeout=5) == "ok"
