"""PCIe rates of the box with pinned host memory: H2D alone, D2H alone, both at once on two streams (GB/s)."""
import time, torch
dev = torch.device("cuda", 0)
n = 256 << 20
h_in = torch.empty(n, dtype=torch.uint8).pin_memory()
h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
d_in = torch.empty(n, dtype=torch.uint8, device=dev)
d_out = torch.empty(n, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(h2d, d2h, reps=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if h2d:
            with torch.cuda.stream(s1):
                d_in.copy_(h_in, non_blocking=True)
        if d2h:
            with torch.cuda.stream(s2):
                h_out.copy_(d_out, non_blocking=True)
    torch.cuda.synchronize()
    return reps * n / (time.perf_counter() - t0) / 1e9
run(True, True, 3)
print("H2D alone %.1f GB/s" % run(True, False))
print("D2H alone %.1f GB/s" % run(False, True))
print("both: %.1f GB/s each direction" % run(True, True))
