"""Compiles and runs the C++ SDK-level test (tests/cpp/test_fhe_task_gpu.cpp) against the in-tree shared library:
the host-side mirror of lattisense::FheTaskGpu in the reference's own language."""
import importlib.util
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_fhe_task_gpu(tmp_path):
    from lattisense_amd import build
    lib = build.LIB          # the library that shipped with the tree (the GPU box never rebuilds it)
    assert os.path.exists(lib), "liblattisense_amd.so is not built: run __graft_entry__.build()"
    libdir = os.path.dirname(lib)
    exe = str(tmp_path / "test_fhe_task_gpu")
    tl = build.torch_lib_dir()
    rpaths = [libdir] + ([tl] if tl else []) + ["/opt/rocm/lib"]
    cmd = ["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "test_fhe_task_gpu.cpp"), "-o", exe, "-L" + libdir, "-llattisense_amd"]
    for r in rpaths:
        cmd += ["-L" + r, "-Wl,-rpath," + r]
    cmd += ["-lamdhip64", "-lpthread"]
    subprocess.check_call(cmd)
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "tasks")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout
