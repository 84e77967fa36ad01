"""Time per phase of the device bootstrap at N=2^16 (LSA_BT_STOP diagnostic): cumulative ms at each checkpoint."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lattisense_amd import params
from lattisense_amd.device import ALGO_CKKS, BootstrapPlan, DeviceContext

B = params.CKKS_BOOTSTRAP_65536
n, q, p = 1 << 16, B["q"], B["p"]
top, np_ = len(q) - 1, len(p)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
ctx = DeviceContext(ALGO_CKKS, n, q, p, 0)
ctx.stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
plan = BootstrapPlan(ctx, in_scale=2.0 ** 40, out_scale=2.0 ** 40)
keep = []
def key(level):
    beta = (level + 1 + np_ - 1) // np_
    t = torch.randint(0, 1 << 38, (beta, 2, level + 1 + np_, n), dtype=torch.int64, device=dev)
    keep.append(t)
    return ctx.adopt_key(t.data_ptr(), level)
rlk = key(top); glk = {e: key(top) for e in plan.galois_elements}; dts, std = key(0), key(top)
x = torch.randint(0, 1 << 38, (batch, 2, 1, n), dtype=torch.int64, device=dev)
class Buf:
    def __init__(self, t): self.t, self.ptr = t, t.data_ptr()
names = ["mul_int", "mod_raise(+swk)", "cts0", "cts1", "cts2", "cts3", "u_re", "u_im", "evalmod x2", "recombine", "full (stc)"]
prev = 0.0
for step in list(range(1, 11)) + [0]:
    if step: os.environ["LSA_BT_STOP"] = str(step)
    else: os.environ.pop("LSA_BT_STOP", None)
    for _ in range(2):
        plan.run(Buf(x), batch, rlk, glk, dts, std).free()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        plan.run(Buf(x), batch, rlk, glk, dts, std).free()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
    print("%-16s cumulative %8.2f ms   phase %8.2f ms  (%.2f ms per ciphertext)" % (names[step - 1 if step else 10], ms, ms - prev, (ms - prev) / batch))
    prev = ms
