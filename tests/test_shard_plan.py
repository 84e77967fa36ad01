"""CPU-only: the multi-device plan behind run_fhe_gpu_task (lattisense_amd/csrc/shard_plan.h) -- independent subgraphs dealt out
to (device, lane pair) shards, every evaluation key uploaded once and copied device-to-device once per other distinct device --
exercised with a recording fake device layer (tests/cpp/test_shard_plan.cpp).  What it replaces: the reference's one run per
device, each exporting and uploading every key (/root/reference/README.md:195-202, mega_ag_runners/gpu/gpu_wrapper.cu:148-149)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_plan_and_key_fan_out(tmp_path):
    exe = str(tmp_path / "test_shard_plan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-fsanitize=address,undefined",
                           os.path.join(ROOT, "tests", "cpp", "test_shard_plan.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK shard_plan" in out.stdout


def test_entry_points_are_exported():
    from lattisense_amd.task import _task_lib
    L = _task_lib()
    assert L.lsa_task_set_devices and L.lsa_task_last_run_shards
    # without a task there is nothing to configure: error code, not a crash
    assert L.lsa_task_set_devices(None, None, 0) != 0
