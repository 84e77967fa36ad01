"""HBM ceilings by read:write mix (torch kernels on a 4 GiB buffer): write-only fill, read-only reduction, copy.
Context for kernels whose traffic is mostly writes (k_baseconv: 4 limbs read, 13 written)."""
import time

import torch

dev = torch.device("cuda", 0)
n = 1 << 29   # int64 elements: 4 GiB
a = torch.empty(n, dtype=torch.int64, device=dev)
b = torch.empty(n, dtype=torch.int64, device=dev)


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t = timed(lambda: a.fill_(3))
print(f"write-only fill : {8 * n / t / 1e9:.0f} GB/s")
t = timed(lambda: a.sum())
print(f"read-only sum   : {8 * n / t / 1e9:.0f} GB/s")
t = timed(lambda: b.copy_(a))
print(f"copy (1r + 1w)  : {16 * n / t / 1e9:.0f} GB/s total")
t = timed(lambda: torch.add(a, b, out=b))
print(f"add (2r + 1w)   : {24 * n / t / 1e9:.0f} GB/s total")
