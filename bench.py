#!/usr/bin/env python3
"""bench.py — throughput of the RNS hot path on MI355X (BASELINE.json metric).

Default workload (the configuration the metric is quoted on, BASELINE.json configs[2]):
  CKKS N=2^16, level 12 (L=13 Q-limbs, k=4 special primes, beta=4 digits), HMult + relinearize + rescale,
  batch 256 ciphertext pairs per GPU, inputs/keys resident in HBM, synthetic uniform residues.
A "step" = one pass of the operator over the whole batch.  value = ciphertexts/s over all ranks.

Other workloads (--workload): ntt (configs[1]: BFV N=2^14, 4 primes, batch 1024 cts fwd+inv NTT; value = GB/s),
rotate (configs[3] shape per GPU), bfv_hmult (examples/benchmark_gpu shape), deep / deep17 (configs[4] deep-chain key switch at
N=2^16 / 2^17), bootstrap (the reference's bootstrap parameter set), task_ckks / task_bfv / task_conv (T2: the same
operators through run_fhe_gpu_task from host C structs, PCIe inclusive -- reported, never `value` of the default run).

Operator workloads time TWO regions in one run: the two-stream tile runner (alternate tiles of the batch on an auxiliary
stream; this is `value`) and a single-stream region (`single_stream`), on whose launch stream the `roofline` sample is taken.

Multi-GPU (torchrun, one rank per GPU): the ciphertext batch is sharded by index (weak scaling: `batch` per rank),
no steady-state collective; rank 0 "ingests" the evaluation key and broadcasts it once over RCCL/xGMI.

The JSON line also carries
  roofline     — the limb-transform passes (k_ntt_r16 / k_ntt_r8x3 / k_ntt_pass, the dominant kernel family): algorithmic bytes / HIP-event
                 duration, sampled live on the launch stream; `traffic` from the committed PMC passes of the same command
  cpu_baseline — the CPU oracle ("port") timed on rank 0's host cores on a bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="ckks_hmult", choices=["ckks_hmult", "ntt", "rotate", "bfv_hmult", "deep", "deep17", "task_ckks", "task_bfv", "task_conv", "bootstrap"])
    ap.add_argument("--batch", type=int, default=0, help="ciphertexts per GPU (0 = workload default)")
    ap.add_argument("--tile", type=int, default=-1, help="ciphertexts per kernel wave (-1 = library default)")
    ap.add_argument("--ntt-chunk-mib", type=int, default=-1, help="Infinity-Cache chunk of two-pass NTTs (-1 = default)")
    ap.add_argument("--int-ntt", action="store_true", help="force the integer butterfly engine (A/B)")
    ap.add_argument("--dual-stream", action="store_true", help="(default for the operator workloads) alternate tiles of an operator on an auxiliary stream")
    ap.add_argument("--single-stream", action="store_true", help="one stream for the timed region too (A/B); the roofline sample is always taken single-stream")
    ap.add_argument("--no-fuse", action="store_true", help="separate ModDown/rescale tail kernels (A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pinned-alloc", action="store_true", help="task workloads: caller buffers from lsa_host_alloc, copied without staging (A/B: slower on the measured hosts)")
    ap.add_argument("--register-in-place", action="store_true", help="task workloads: caller buffers pinned in place with lsa_host_register (A/B)")
    ap.add_argument("--prof-stride", type=int, default=4)
    ap.add_argument("--log-slots", type=int, default=0, help="bootstrap workload: log2 of the packed slots (0 = dense, N/2)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="--gpus N started plainly: overall deadline in seconds for the ranks this process starts")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launch path only: every rank joins a gloo group, rank 0 prints what it sees (no GPU work; CPU test)")
    return ap.parse_args()


def visible_gpu_count():
    """Number of GPUs this process may use, found WITHOUT loading a GPU runtime (no torch, no HIP: a launcher that has
    initialised HIP must not start other programs on this pool).  KFD topology nodes with SIMDs are the GPUs; the
    *_VISIBLE_DEVICES variables narrow the set.  None = cannot tell from here (no KFD sysfs): the ranks check for themselves."""
    import glob
    n = None
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if nodes:
        n = 0
        for f in nodes:
            try:
                for ln in open(f):
                    k, _, v = ln.partition(" ")
                    if k == "simd_count" and int(v) > 0:
                        n += 1
            except (OSError, ValueError):
                pass
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            listed = len([x for x in v.split(",") if x.strip() != ""])
            n = listed if n is None else min(n, listed)
    return n


def profiler_preloaded():
    """rocprofv3 & co. preload a tool library that initialises the GPU before main(): starting ranks from such a process is
    the fork+exec this pool forbids.  Counter / trace runs are --gpus 1 (or one profiler per rank under torchrun)."""
    pre = os.environ.get("LD_PRELOAD", "")
    return ("rocprof" in pre or "roctracer" in pre or "ROCP_TOOL_LIBRARIES" in os.environ
            or "ROCPROFILER_REGISTER_FORCE_LOAD" in os.environ or "ROCPROF_OUTPUT_PATH" in os.environ)


def self_launch(args):
    """`bench.py --gpus N` outside torchrun: start N fresh rank processes (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set)
    from a parent that never loads torch or HIP, poll them, and on the first failure (or the overall deadline) stop the
    others -- ranks blocked in the RCCL rendezvous / broadcast / barrier would otherwise sit there until the driver's
    timeout.  Returns the first non-zero exit code.  Rank 0 prints the JSON line on the shared stdout.  (The reference's
    multi-GPU usage is one device index per call, README.md:195-202; its harness examples/benchmark_gpu/benchmark_gpu.cpp:27-52
    takes the device the same way.)"""
    import signal
    import socket
    import subprocess
    n = args.gpus
    if profiler_preloaded():
        raise SystemExit("bench.py --gpus %d under a profiler preload: the launcher would start ranks from a GPU-initialised "
                         "process; profile with --gpus 1, or wrap each rank under torchrun" % n)
    if not args.dry_launch:
        have = visible_gpu_count()
        if have is not None and have < n:
            raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible" % (n, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # the proof the CPU test reads: nothing GPU-related was imported into the process that starts the ranks
    gpu_mods = sorted(m for m in sys.modules if m == "torch" or m.startswith("torch.") or m.startswith("lattisense_amd"))
    print("launcher: " + json.dumps({"pid": os.getpid(), "ranks": n, "port": port, "gpu_modules_loaded": gpu_mods}),
          file=sys.stderr, flush=True)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # own process group per rank: stopping a rank also stops whatever it started
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, start_new_session=True))

    def stop_all(sig):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)
                except (ProcessLookupError, PermissionError):
                    pass

    deadline = time.monotonic() + args.launch_timeout
    rc, why = 0, None
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                rc = abs(bad[0][1]) or 1
                why = "rank %d exited with code %d" % bad[0]
                break
            if all(c == 0 for c in codes):
                return 0
            if time.monotonic() > deadline:
                rc, why = 124, "no result after %.0f s (--launch-timeout)" % args.launch_timeout
                break
            time.sleep(0.05)
    except KeyboardInterrupt:
        rc, why = 130, "interrupted"
    print("launcher: %s; stopping the other ranks" % why, file=sys.stderr, flush=True)
    stop_all(signal.SIGTERM)
    t_kill = time.monotonic() + 5.0
    while any(p.poll() is None for p in procs) and time.monotonic() < t_kill:
        time.sleep(0.05)
    stop_all(signal.SIGKILL)
    for p in procs:
        p.wait()
    return rc


def dry_launch(args):
    """what --dry-launch ranks do: rendezvous over gloo exactly as the timed path does over RCCL, gather the ranks"""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if os.environ.get("LSA_DRY_FAIL_RANK") == str(rank):   # test hook: this rank dies before the rendezvous
        sys.exit(3)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = [None] * world
    dist.all_gather_object(seen, {"rank": rank, "local_rank": int(os.environ["LOCAL_RANK"]), "pid": os.getpid()})
    t = torch.tensor([float(rank)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "ranks": seen, "max_rank": t.item()}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def workload_config(name):
    from lattisense_amd import params
    if name == "ckks_hmult":
        P = params.CKKS_DEFAULT[65536]
        return dict(algo=1, n=65536, q=P["q"][:13], p=P["p"], t=0, level=12, batch=256,
                    label="CKKS N=2^16 L=13 k=4 HMult+relin+rescale", unit="ciphertexts/s",
                    metric="ckks_hmult_relin_rescale_throughput")
    if name == "rotate":
        P = params.CKKS_DEFAULT[65536]
        return dict(algo=1, n=65536, q=P["q"][:13], p=P["p"], t=0, level=12, batch=256,
                    label="CKKS N=2^16 L=13 k=4 rotate (Galois key-switch)", unit="ciphertexts/s",
                    metric="ckks_rotate_throughput")
    if name == "bfv_hmult":
        P = params.BFV_DEFAULT[16384]
        return dict(algo=0, n=16384, q=P["q"], p=P["p"], t=P["t"], level=3, batch=1024,
                    label="BFV N=2^14 level 3 mult+relin (examples/benchmark_gpu shape)", unit="ciphertexts/s",
                    metric="bfv_hmult_relin_throughput")
    if name == "deep":
        P = params.CKKS_BOOTSTRAP_65536
        return dict(algo=1, n=65536, q=P["q"], p=P["p"], t=0, level=24, batch=64,
                    label="CKKS N=2^16 25Q+5P deep-chain HMult+relin+rescale", unit="ciphertexts/s",
                    metric="ckks_deep_hmult_throughput")
    if name == "deep17":
        P = params.ckks_n17_chain()
        return dict(algo=1, n=1 << 17, q=P["q"], p=P["p"], t=0, level=24, batch=32,
                    label="CKKS N=2^17 25Q+5P (generated chain) HMult+relin+rescale", unit="ciphertexts/s",
                    metric="ckks_n17_deep_hmult_throughput")
    P = params.BFV_DEFAULT[16384]
    return dict(algo=0, n=16384, q=P["q"], p=P["p"], t=P["t"], level=3, batch=1024,
                label="BFV N=2^14 4 primes batch 1024 ct NTT+INTT", unit="GB/s", metric="ntt_intt_algorithmic_bandwidth")


def run_task_workload(args):
    """End-to-end through the drop-in boundary run_fhe_gpu_task with HOST buffers (export, pinned staging, H2D, batched
    operators, D2H, import): the PCIe-inclusive rate noted in DESIGN.md §6 — never the headline `value`."""
    import numpy as np
    from lattisense_amd.task import Argument, Ciphertext, FheTaskGpu, KeySwitchKey
    name = "ckks_n65536_l12_cmc_relin_rescale_x64" if args.workload == "task_ckks" else "bfv_n16384_l3_cmc_relin_x256"
    path = os.path.join(ROOT, "tests", "golden", "tasks_bench", name)
    g = json.load(open(os.path.join(path, "mega_ag.json")))
    P = g["parameter"]
    n, q, p = P["n"], P["q"][: P["max_level"] + 1], P["p"]
    n_op = len(g["outputs"])
    lvl = 12 if args.workload == "task_ckks" else 3
    out_lvl = lvl - 1 if args.workload == "task_ckks" else lvl
    rng = np.random.default_rng(0)

    def rand(shape_prefix, mods):
        out = np.empty((*shape_prefix, len(mods), n), dtype=np.uint64)
        for i, m in enumerate(mods):
            out[..., i, :] = rng.integers(0, m, size=(*shape_prefix, n), dtype=np.uint64)
        return out

    xs = [Ciphertext(rand((2,), q[: lvl + 1])) for _ in range(n_op)]
    ys = [Ciphertext(rand((2,), q[: lvl + 1])) for _ in range(n_op)]
    zs = [Ciphertext.empty(1, out_lvl, n) for _ in range(n_op)]
    beta = (lvl + 1 + len(p) - 1) // len(p)
    rlk = KeySwitchKey(rand((beta, 2), q[: lvl + 1] + p), lvl, len(p))
    t = FheTaskGpu(path)
    ins = [Argument("xs", xs), Argument("ys", ys), Argument("rlk_ntt", [rlk])]
    outs = [Argument("zs", zs)]
    pinned = []
    if args.register_in_place:   # A/B: the caller's own buffers pinned in place (lsa_host_register)
        from lattisense_amd.task import register_host
        for obj in xs + ys + zs:
            register_host(obj.data)
            pinned.append(obj.data)
    elif args.pinned_alloc:   # the caller keeps its limbs in pinned memory from the library's allocator: zero-copy ingestion
        from lattisense_amd.task import alloc_host

        def to_pinned(cts):
            out = []
            for ct in cts:
                a = alloc_host(ct.data.shape)
                a[...] = ct.data
                pinned.append(a)
                out.append(Ciphertext(a))
            return out
        xs, ys, zs = to_pinned(xs), to_pinned(ys), to_pinned(zs)
        ins = [Argument("xs", xs), Argument("ys", ys), Argument("rlk_ntt", [rlk])]
        outs = [Argument("zs", zs)]
    for _ in range(args.warmup):
        t.run(ins, outs)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t.run(ins, outs)
    dt = time.perf_counter() - t0
    st = t.last_run_stats()
    print(json.dumps({
        "metric": "task_end_to_end_throughput", "value": n_op * args.steps / dt, "unit": "ciphertexts/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": name + " through run_fhe_gpu_task (host buffers, PCIe-inclusive)", "n_op": n_op,
                   "gpu_nodes": st["gpu_nodes"], "batched_launch_groups": st["gpu_batches"],
                   "keys_last_run": t.last_run_keys(), "direct_copies_last_run": t.last_run_direct(),
                   "caller_buffers": "pinned in place" if args.register_in_place else ("library pinned allocator" if args.pinned_alloc else "pageable (staged)")},
        "roofline": None, "cpu_baseline": None}), flush=True)


def run_conv_workload(args):
    """The reference's application benchmark shape (examples/benchmark_convolution, config 4 in / 4 out channels of 32x32,
    3x3 kernel, CKKS N=16384 at level 2) end-to-end through run_fhe_gpu_task with host buffers: 71 rotations (NAF-shared),
    72 ct x pt products, a 72-term accumulation, rescale, bias.  Arguments are built from the task signature."""
    import numpy as np
    from lattisense_amd.task import Argument, Ciphertext, FheTaskGpu, GaloisKey, KeySwitchKey, Plaintext
    name = "ckks_n16384_conv2d_4in_4out_32x32_3x3"
    path = os.path.join(ROOT, "tests", "golden", "tasks_bench", name)
    g = json.load(open(os.path.join(path, "mega_ag.json")))
    sig = json.load(open(os.path.join(path, "task_signature.json")))
    P = g["parameter"]
    n, q, p = P["n"], P["q"][: P["max_level"] + 1], P["p"]
    rng = np.random.default_rng(0)

    def rand(shape_prefix, mods):
        out = np.empty((*shape_prefix, len(mods), n), dtype=np.uint64)
        for i, m in enumerate(mods):
            out[..., i, :] = rng.integers(0, m, size=(*shape_prefix, n), dtype=np.uint64)
        return out

    ins, outs = [], []
    for a in sig["online"]:
        count = int(np.prod(a["size"]))
        lvl = a["level"]
        if a["phase"] == "out":
            outs.append(Argument(a["id"], [Ciphertext.empty(1, lvl, n) for _ in range(count)]))
        elif a["type"] == "ct":
            ins.append(Argument(a["id"], [Ciphertext(rand((2,), q[: lvl + 1])) for _ in range(count)]))
        else:
            ins.append(Argument(a["id"], [Plaintext(rand((), q[: lvl + 1])) for _ in range(count)]))
    keys = {}
    for e, lvl in sig["key"]["glk"].items():
        beta = (lvl + 1 + len(p) - 1) // len(p)
        keys[int(e)] = KeySwitchKey(rand((beta, 2), q[: lvl + 1] + p), lvl, len(p))
    ins.append(Argument("glk_ntt", [GaloisKey(keys)]))
    t = FheTaskGpu(path)
    for _ in range(args.warmup):
        t.run(ins, outs)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t.run(ins, outs)
    dt = time.perf_counter() - t0
    st = t.last_run_stats()
    print(json.dumps({
        "metric": "conv2d_layer_end_to_end_rate", "value": args.steps / dt, "unit": "layers/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": name + " through run_fhe_gpu_task (host buffers, PCIe-inclusive)",
                   "compute_nodes": len(g["compute"]), "gpu_nodes": st["gpu_nodes"],
                   "batched_launch_groups": st["gpu_batches"]},
        "roofline": None, "cpu_baseline": None}), flush=True)


def run_bootstrap_workload(args):
    """CKKS bootstrapping at the reference's default bootstrap parameter set (N=2^16, 25 Q + 5 P primes, CtS depth 4, Cos1
    K=16 degree 30 with 3 double angles, StC depth 3, level 0 -> 9; frontend/custom_task.py:383-468), device-resident
    synthetic inputs and keys (the relinearisation key, one Galois key per planner rotation + conjugation, swk_dts / swk_std).
    A step bootstraps `batch` ciphertexts in one batched program."""
    import ctypes
    import torch
    from lattisense_amd import params
    from lattisense_amd._native import check, lib
    from lattisense_amd.device import ALGO_CKKS, BootstrapPlan, DeviceContext
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible and there is no CPU fallback")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    B = params.CKKS_BOOTSTRAP_65536
    n, q, p = 1 << 16, B["q"], B["p"]
    top, np_ = len(q) - 1, len(p)
    batch = args.batch or 16
    ctx = DeviceContext(ALGO_CKKS, n, q, p, 0, device=local_rank)
    ctx.stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L_ = lib()
    gen = torch.Generator(device=dev)
    gen.manual_seed(77)

    def uniform(shape_prefix, mods):
        out = torch.empty(*shape_prefix, len(mods), n, dtype=torch.int64, device=dev)
        for i, m in enumerate(mods):
            out[..., i, :] = torch.randint(0, m, (*shape_prefix, n), dtype=torch.int64, device=dev, generator=gen)
        return out

    t0 = time.perf_counter()
    plan = BootstrapPlan(ctx, in_scale=2.0 ** 40, out_scale=2.0 ** 40, log_slots=args.log_slots)
    t_plan = time.perf_counter() - t0
    keep = []

    def key(level):
        beta = (level + 1 + np_ - 1) // np_
        t = uniform((beta, 2), q[: level + 1] + p)
        keep.append(t)
        return ctx.adopt_key(t.data_ptr(), level)

    rlk = key(top)
    glk = {e: key(top) for e in plan.galois_elements}
    dts, std = key(0), key(top)
    x = uniform((batch, 2), q[:1])

    class Buf:
        def __init__(self, t):
            self.t, self.ptr = t, t.data_ptr()

    def step():
        return plan.run(Buf(x), batch, rlk, glk, dts, std)

    for _ in range(args.warmup):
        step().free()
    torch.cuda.synchronize()
    check(L_.lsa_profile_begin(ctx.h, args.prof_stride))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step().free()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    check(L_.lsa_profile_end(ctx.h))
    kinds = {0: "k_ntt_pass", 1: "k_baseconv", 2: "k_ks_mac", 3: "k_tensor", 4: "elementwise"}
    breakdown = {}
    for kid, name in kinds.items():
        ms, by = ctypes.c_double(), ctypes.c_double()
        ns, nl = ctypes.c_longlong(), ctypes.c_longlong()
        check(L_.lsa_profile_read(ctx.h, kid, ctypes.byref(ms), ctypes.byref(by), ctypes.byref(ns), ctypes.byref(nl)))
        if ns.value:
            breakdown[name] = {"est_ms_per_step": ms.value / ns.value * nl.value / args.steps, "avg_launch_us": ms.value / ns.value * 1e3,
                               "achieved_GBps": by.value / ms.value / 1e6, "launches_per_step": nl.value / args.steps}
    ntt = breakdown.get("k_ntt_pass")
    roofline = None
    if ntt:
        roofline = {"kernel": "k_ntt_pass", "bound": "hbm", "achieved": ntt["achieved_GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": ntt["achieved_GBps"] / HBM_PEAK_GBPS, "traffic": None, "avg_launch_us": ntt["avg_launch_us"],
                    "launches_per_step": ntt["launches_per_step"]}
    print(json.dumps({
        "metric": "ckks_bootstrap_throughput", "value": batch * args.steps / dt, "unit": "bootstraps/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "CKKS bootstrap N=2^16 25Q+5P (N16QP1546H192H32), %s, level 0 -> %d" % (
            "2^%d slots" % args.log_slots if plan.sparse else "dense packing", plan.out_level),
                   "batch_per_gpu": batch, "galois_keys": len(glk), "key_bytes_total": sum(t.numel() * 8 for t in keep),
                   "plan_build_s": t_plan, "ms_per_bootstrap": dt / args.steps / batch * 1e3},
        "roofline": roofline, "cpu_baseline": None, "kernel_breakdown": breakdown}), flush=True)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%s: launch one rank per GPU (torchrun --nproc-per-node %d, or "
                         "plain `bench.py --gpus %d`, which starts the ranks itself)" % (args.gpus, os.environ["WORLD_SIZE"], args.gpus, args.gpus))
    if args.dry_launch:
        if "WORLD_SIZE" not in os.environ:
            os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        return dry_launch(args)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and (args.workload == "bootstrap" or args.workload.startswith("task_")):
        raise SystemExit("workload %s is a single-device measurement (replicas only): run it with --gpus 1" % args.workload)
    if args.workload == "bootstrap":
        return run_bootstrap_workload(args)
    if args.workload == "task_conv":
        return run_conv_workload(args)
    if args.workload.startswith("task_"):
        return run_task_workload(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible and there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from lattisense_amd import sharding
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        sharding.init_process_group("nccl", rank, world, device=dev)   # backend "nccl" is RCCL on ROCm

    from lattisense_amd._native import check, lib
    from lattisense_amd.device import DeviceContext
    L_ = lib()
    cfg = workload_config(args.workload)
    n, lvl = cfg["n"], cfg["level"]
    L = lvl + 1
    batch = args.batch or cfg["batch"]
    ctx = DeviceContext(cfg["algo"], n, cfg["q"], cfg["p"], cfg["t"], device=local_rank)
    stream = torch.cuda.current_stream()
    ctx.stream = ctypes.c_void_p(stream.cuda_stream)
    if args.tile >= 0:
        ctx.set_tile_batch(args.tile)
    if args.ntt_chunk_mib >= 0:
        check(L_.lsa_set_ntt_chunk_mib(ctx.h, args.ntt_chunk_mib))
    if args.int_ntt:
        ctx.set_fp64_ntt(False)
    # The timed region runs the operator's tiles alternately on two streams (each kernel of the pipeline uses part of the chip:
    # two tiles in flight fill each other's gaps, +2-3 % on the headline, profiles/r03/ab_dual_stream_fused_build.log); per-kernel
    # durations are only attributable when one kernel runs at a time, so the roofline / kernel breakdown come from a SECOND,
    # single-stream measured region of the same run (reported next to the headline as `single_stream`).
    dual = args.workload != "ntt" and not args.single_stream
    check(L_.lsa_set_dual_stream(ctx.h, 1 if dual else 0))
    if args.no_fuse:
        check(L_.lsa_set_fuse_tails(ctx.h, 0))

    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)

    def uniform(shape_prefix, mods):
        """[..., len(mods), n] int64 tensor of uniform residues (synthetic data of the workload's shape)"""
        out = torch.empty(*shape_prefix, len(mods), n, dtype=torch.int64, device=dev)
        for i, m in enumerate(mods):
            out[..., i, :] = torch.randint(0, m, (*shape_prefix, n), dtype=torch.int64, device=dev, generator=gen)
        return out

    class Buf:  # adapter: torch tensor -> object with .ptr for DeviceContext methods
        def __init__(self, t):
            self.t, self.ptr = t, t.data_ptr()

    qs = cfg["q"][:L]
    key = None
    key_t = None
    if args.workload != "ntt":
        # evaluation key: rank 0 ingests, one-time broadcast to the other ranks over RCCL (xGMI), then read-only
        np_ = len(cfg["p"])
        beta = (L + np_ - 1) // np_
        kmods = qs + cfg["p"]
        if rank == 0:
            key_t = uniform((beta, 2), kmods)
        else:
            key_t = torch.empty(beta, 2, len(kmods), n, dtype=torch.int64, device=dev)
        sharding.broadcast_key(key_t, src=0)
        torch.cuda.synchronize()
        assert key_t.numel() * 8 == ctx.key_bytes(lvl)
        key = ctx.adopt_key(key_t.data_ptr(), lvl)

    a = uniform((batch, 2), qs)
    b = uniform((batch, 2), qs) if args.workload in ("ckks_hmult", "bfv_hmult", "deep", "deep17") else None
    if args.workload in ("ckks_hmult", "deep", "deep17"):
        out = torch.empty(batch, 2, lvl, n, dtype=torch.int64, device=dev)
    else:
        out = torch.empty(batch, 2, L, n, dtype=torch.int64, device=dev)
    g_rot = pow(5, 1, 2 * n)

    def step():
        if args.workload in ("ckks_hmult", "deep", "deep17"):
            ctx.ckks_mult_relin_rescale(lvl, Buf(a), Buf(b), key, batch, out=Buf(out))
        elif args.workload == "bfv_hmult":
            ctx.bfv_mult_relin(lvl, Buf(a), Buf(b), key, batch, out=Buf(out))
        elif args.workload == "rotate":
            check(L_.lsa_ckks_rotate(ctx.h, lvl, a.data_ptr(), g_rot, key, out.data_ptr(), batch, 2 * L * n, 2 * L * n,
                                     ctx.stream))
        else:
            mo = list(range(L))
            ctx.ntt(Buf(a), batch, 2 * L, mo, inverse=False)
            ctx.ntt(Buf(a), batch, 2 * L, mo, inverse=True)

    def barrier():
        torch.cuda.synchronize()
        sharding.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    if not dual:
        check(L_.lsa_profile_begin(ctx.h, args.prof_stride))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if not dual:
        check(L_.lsa_profile_end(ctx.h))
    dt = sharding.max_over_ranks(dt, device=dev)
    single = None
    if dual:
        # second measured region: one stream, HIP-event samples around the kernels (roofline, kernel breakdown)
        check(L_.lsa_set_dual_stream(ctx.h, 0))
        k2 = max(2, min(args.steps, 10))
        step()
        barrier()
        check(L_.lsa_profile_begin(ctx.h, args.prof_stride))
        t1 = time.perf_counter()
        for _ in range(k2):
            step()
        barrier()
        dt1 = time.perf_counter() - t1
        check(L_.lsa_profile_end(ctx.h))
        dt1 = sharding.max_over_ranks(dt1, device=dev)
        single = {"value": world * batch * k2 / dt1, "ms_per_step": dt1 / k2 * 1e3, "steps": k2}
    prof_steps = single["steps"] if single else args.steps

    ms_per_step = dt / args.steps * 1e3
    if args.workload == "ntt":
        # 2 transforms (fwd+inv) x batch*2*L limbs x 16N algorithmic bytes
        value = world * 2 * batch * 2 * L * 16.0 * n / (dt / args.steps) / 1e9
    else:
        value = world * batch * args.steps / dt

    # ---- roofline of the dominant kernel (k_ntt_pass), from the sampled HIP events of the timed region
    def prof(kind):
        ms, by = ctypes.c_double(), ctypes.c_double()
        ns, nl = ctypes.c_longlong(), ctypes.c_longlong()
        check(L_.lsa_profile_read(ctx.h, kind, ctypes.byref(ms), ctypes.byref(by), ctypes.byref(ns), ctypes.byref(nl)))
        return ms.value, by.value, ns.value, nl.value

    kinds = {0: "k_ntt_pass", 1: "k_baseconv", 2: "k_ks_mac", 3: "k_tensor", 4: "elementwise"}
    breakdown = {}
    for kid, name in kinds.items():
        ms, by, ns, nl = prof(kid)
        if ns:
            breakdown[name] = {"est_ms_per_step": ms / ns * nl / prof_steps, "avg_launch_us": ms / ns * 1e3,
                               "achieved_GBps": by / ms / 1e6, "launches_per_step": nl / prof_steps}
    ntt = breakdown.get("k_ntt_pass", None)
    ntt_primary = None
    if ntt:
        ms0, _, _, _ = prof(0)
        byp = ctypes.c_double()
        check(L_.lsa_profile_read_primary(ctx.h, 0, ctypes.byref(byp)))
        ntt_primary = byp.value / ms0 / 1e6
    roofline = None
    if ntt:
        roofline = {"kernel": "k_ntt_r16 / k_ntt_r8x3 / k_ntt_pass (limb-transform passes)", "bound": "hbm", "achieved": ntt["achieved_GBps"], "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": ntt["achieved_GBps"] / HBM_PEAK_GBPS, "traffic": None,
                    "avg_launch_us": ntt["avg_launch_us"], "launches_per_step": ntt["launches_per_step"],
                    # `achieved` counts the algorithmic bytes of everything these launches do: the transform passes (16 N / 2 per
                    # limb and pass) AND, for the fused second pass + key MAC launches, the key MAC's bytes as k_ks_mac counts them;
                    # the transforms' bytes alone over the same durations:
                    "achieved_transform_bytes_only": ntt_primary, "frac_transform_bytes_only": ntt_primary / HBM_PEAK_GBPS,
                    "sampling": "HIP event pair around one launch in %d (hash-picked), on the launch stream%s" % (
                        args.prof_stride, "; single-stream region of the same run (the timed region overlaps two tiles)" if single else "")}

    # HBM bytes per launch from the PMC counters cannot be collected inside this process (rocprofv3 wraps the process); they
    # come from the committed --pmc passes of this same command (tools/profile.sh + tools/pmc_traffic.py -> profiles/rNN/),
    # matched on workload, batch AND a hash of the kernel sources: a file collected on other kernels is not reported.
    if roofline:
        try:
            import glob
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from pmc_traffic import sources_hash
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic_k_ntt_pass.json")))
            pmc = json.load(open(cands[-1])) if cands else None
            if pmc and pmc["workload"] == args.workload and pmc["batch"] == batch:
                if pmc.get("kernel_sources_sha256") == sources_hash():
                    roofline["traffic"] = pmc["hbm_bytes_per_launch"]
                    roofline["traffic_source"] = os.path.relpath(cands[-1], ROOT)
                else:
                    roofline["traffic_source"] = "stale: %s was collected on other kernel sources" % os.path.relpath(cands[-1], ROOT)
                roofline["algorithmic_bytes_per_launch"] = ntt["achieved_GBps"] * 1e9 * ntt["avg_launch_us"] * 1e-6
        except Exception:
            pass

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.workload, cfg)

    if rank == 0:
        line = {
            "metric": cfg["metric"], "value": value, "unit": cfg["unit"], "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": cfg["label"], "batch_per_gpu": batch, "ring_degree": n, "q_limbs": L,
                       "special_primes": len(cfg["p"]), "sharding": "ciphertext batch by rank; key broadcast once"},
            "roofline": roofline, "cpu_baseline": cpu, "kernel_breakdown": breakdown,
            "tile_streams": 2 if dual else 1, "single_stream": single,
            "build_flags": L_.lsa_build_flags().decode(),   # "" = the product library (csrc/build_flags.h)
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(workload, cfg):
    """Times the CPU oracle (a plain-C restatement, `kind: port`) on a bounded sample of the same workload,
    one ciphertext per thread over the host cores available to this process."""
    import tempfile
    import numpy as np
    from oracle import pyoracle
    # the timing build of the oracle source: -O3 -march=native (same residues: tests/test_oracle_math.py), compiled here on
    # the machine that runs it
    flags = "gcc -O2 (checker build)"
    try:
        os.environ["LS_ORACLE_LIB"] = pyoracle.build_fast(tempfile.mkdtemp(prefix="lsa_cpu_baseline_"))
        flags = "gcc -O3 -march=native"
    except Exception:
        os.environ.pop("LS_ORACLE_LIB", None)
    Oracle = pyoracle.Oracle
    n, lvl = cfg["n"], cfg["level"]
    L = lvl + 1
    cores = max(1, min(len(os.sched_getaffinity(0)), 32))
    cpu_model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    o = Oracle(n, cfg["q"], cfg["p"], cfg["t"])
    rng = np.random.default_rng(0)
    qs = cfg["q"][:L]

    def rand_ct():
        ct = np.empty((2, L, n), dtype=np.uint64)
        for i, m in enumerate(qs):
            ct[:, i, :] = rng.integers(0, m, size=(2, n), dtype=np.uint64)
        return ct

    key = None
    if workload != "ntt":
        np_ = len(cfg["p"])
        beta = (L + np_ - 1) // np_
        key = np.empty((beta, 2, L + np_, n), dtype=np.uint64)
        for j, m in enumerate(qs + cfg["p"]):
            key[:, :, j, :] = rng.integers(0, m, size=(beta, 2, n), dtype=np.uint64)
    a, b = rand_ct(), rand_ct()
    per_thread = 200 if workload == "ntt" else 6   # ~10-20 s of CPU work per thread either way
    done = []

    def work():
        for _ in range(per_thread):
            if workload in ("ckks_hmult", "deep", "deep17"):
                o.ckks_mult_relin_rescale(lvl, a, b, key, lvl)
            elif workload == "bfv_hmult":
                o.bfv_mult_relin(lvl, a, b, key, lvl)
            elif workload == "rotate":
                o.ckks_rotate(lvl, a, 5, key, lvl)
            else:
                for p in range(2):
                    for i in range(L):
                        o.intt(i, o.ntt(i, a[p, i]))
            done.append(1)

    th = [threading.Thread(target=work) for _ in range(cores)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    units = len(done)
    if workload == "ntt":
        val, unit = units * 2 * 2 * L * 16.0 * n / dt / 1e9, "GB/s"
    else:
        val, unit = units / dt, "ciphertexts/s"
    return {"value": val, "unit": unit, "cores": cores, "kind": "port", "cpu_model": cpu_model, "host_cores_visible": os.cpu_count(),
            "sample": "%d ciphertext op(s) of the same shape over %d threads (%d each), oracle/ls_oracle.c restatement (%s; not Lattigo), %.1f s wall"
                      % (units, cores, per_thread, flags, dt)}


if __name__ == "__main__":
    main()
