"""CPU-side check of bench.py's N-rank launch: `bench.py --gpus N` outside torchrun starts N fresh rank processes itself
(before anything touches a GPU), each with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, and rank 0 reports n_gpus = N.
--dry-launch runs exactly that path with a gloo rendezvous and no GPU work.  (Reference usage: one device per call,
/root/reference/README.md:195-202; harness examples/benchmark_gpu/benchmark_gpu.cpp:27-52.)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e["MASTER_ADDR"] = "127.0.0.1"
    return e


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stdout + r.stderr
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["max_rank"] == 1.0
    assert sorted(x["rank"] for x in d["ranks"]) == [0, 1]
    assert sorted(x["local_rank"] for x in d["ranks"]) == [0, 1]
    assert len({x["pid"] for x in d["ranks"]}) == 2          # two fresh processes, neither is the launcher


def test_gpus_1_runs_in_process():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-launch"], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stdout + r.stderr
    assert _json_line(r.stdout)["n_gpus"] == 1


def test_torchrun_style_env_is_respected_and_mismatch_fails_loudly():
    # launched the driver's way: ranks already exist, --gpus must agree with WORLD_SIZE
    e = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_PORT="29611")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-launch"], capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode == 0 and _json_line(r.stdout)["n_gpus"] == 1
    e["WORLD_SIZE"] = "4"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def _launcher_line(err):
    lines = [ln for ln in err.splitlines() if ln.startswith("launcher: {")]
    assert len(lines) == 1, err
    return json.loads(lines[0][len("launcher: "):])


def test_launcher_process_never_loads_torch_or_the_gpu_library():
    # the process that starts the ranks must not have initialised a GPU runtime: it reports what it had imported
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stdout + r.stderr
    ln = _launcher_line(r.stderr)
    assert ln["gpu_modules_loaded"] == [] and ln["ranks"] == 2
    assert ln["pid"] not in {x["pid"] for x in _json_line(r.stdout)["ranks"]}


def _tagged_processes(tag):
    found = []
    for pid in os.listdir("/proc"):
        if not pid.isdigit():
            continue
        try:
            if ("LSA_TEST_TAG=" + tag).encode() in open("/proc/%s/environ" % pid, "rb").read():
                found.append(int(pid))
        except OSError:
            pass
    return found


def test_a_dead_rank_stops_its_siblings_and_the_launcher_fails_fast():
    import time
    import uuid
    tag = uuid.uuid4().hex
    e = dict(_env(), LSA_DRY_FAIL_RANK="1", LSA_TEST_TAG=tag)
    t0 = time.monotonic()
    # rank 1 exits 3 before the rendezvous; rank 0 would wait for it in init_process_group for minutes
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=120, env=e)
    dt = time.monotonic() - t0
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert "rank 1 exited with code 3" in r.stderr
    assert dt < 60, dt
    assert _tagged_processes(tag) == []          # no orphan rank left behind


def test_launcher_deadline():
    import time
    import uuid
    tag = uuid.uuid4().hex
    # world of 2 but this "torchrun-less" launch only ever gets one live rank: rank 1 is made to hang by pointing it at a
    # rendezvous nobody serves is not needed -- a 0.5 s deadline expires while the ranks are still importing torch
    e = dict(_env(), LSA_TEST_TAG=tag)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch", "--launch-timeout", "0.5"], capture_output=True,
                       text=True, timeout=120, env=e)
    assert r.returncode == 124 and "--launch-timeout" in r.stderr
    time.sleep(0.2)
    assert _tagged_processes(tag) == []


def test_device_count_comes_from_sysfs_and_the_visibility_variables():
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    old = {k: os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")}
    try:
        for k in old:
            os.environ.pop(k, None)
        base = bench.visible_gpu_count()     # None here (no KFD sysfs in the build container), the GPU count on a GPU box
        os.environ["HIP_VISIBLE_DEVICES"] = "0,1,2"
        os.environ["ROCR_VISIBLE_DEVICES"] = "0"
        assert bench.visible_gpu_count() == (1 if base is None else min(base, 1))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert "torch" not in [m for m in sys.modules if m == "torch"] or True   # (this test process may have torch; the launcher must not)


def test_without_devices_the_launch_fails_loudly():
    # one device allowed, two asked for: refused by the launcher itself, before any rank starts
    e = dict(_env(), HIP_VISIBLE_DEVICES="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode != 0 and "only 1 HIP device(s) visible" in r.stderr
    if os.path.exists("/sys/class/kfd/kfd/topology/nodes"):
        return
    # no KFD and no visibility variable: the launcher cannot count, the ranks find no device and fail; the launcher reports
    # the first failure and stops the other rank
    e = {k: v for k, v in _env().items() if not k.endswith("_VISIBLE_DEVICES")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode != 0 and "no HIP device visible" in r.stderr and "exited with code" in r.stderr


def test_refuses_to_start_ranks_under_a_profiler_preload():
    e = dict(_env(), LD_PRELOAD="")
    e["ROCP_TOOL_LIBRARIES"] = "/opt/rocm/lib/librocprofiler-sdk-tool.so"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=60, env=e)
    assert r.returncode != 0 and "profiler preload" in r.stderr
