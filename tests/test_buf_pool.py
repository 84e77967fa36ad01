"""CPU-only: the task runtime's device/pinned buffer pools are keyed by (device, lane) (tests/cpp/test_buf_pool.cpp, compiled
here with g++ against lattisense_amd/csrc/buf_pool.h and a fake allocator).  One task handle may run on any device
(/root/reference/README.md:195-202; mega_ag_runners/gpu/gpu_wrapper.cu:148-149)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_buf_pool_keying(tmp_path):
    exe = str(tmp_path / "test_buf_pool")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-fsanitize=address,undefined",
                           os.path.join(ROOT, "tests", "cpp", "test_buf_pool.cpp"), "-o", exe, "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK buf_pool" in out.stdout
