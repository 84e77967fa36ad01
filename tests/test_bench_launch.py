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


def test_without_devices_the_launcher_refuses():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two devices visible")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode != 0 and "HIP device(s) visible" in r.stderr
