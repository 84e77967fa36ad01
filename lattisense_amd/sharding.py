"""Multi-GPU plumbing for the hot path: ciphertext batches are independent units, so they are sharded by
index across ranks (one process per GPU) with NO steady-state collective; the only exchange is a one-time broadcast of
each evaluation key from the rank that ingested it (RCCL over xGMI on GPUs; gloo in the CPU tests)."""
import hashlib

import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world):
    """Contiguous block [start, stop) of rank `rank` when n_items units are split over `world` ranks."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def init_process_group(backend, rank, world, device=None):
    if world <= 1 or dist.is_initialized():
        return
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)


def broadcast_key(key_tensor, src=0):
    """One-time broadcast of an evaluation key (compact ABI layout, int64 view of the u64 limbs). In place."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(key_tensor, src=src)
    return key_tensor


def max_over_ranks(seconds, device="cpu"):
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def tensor_digest(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()
