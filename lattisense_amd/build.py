"""Builds the native library in-tree: hipcc --offload-arch=gfx950 per source, linked against the HIP runtime that
torch bundles (so one HIP runtime lives in a process that also imports torch for RCCL / device memory)."""
import importlib.util
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblattisense_amd.so")
HIP_SOURCES = ["kernels.hip", "context.hip", "ops.hip", "bootstrap.hip", "c_api.hip", "task_runtime.hip"]
CXX_SOURCES = ["tables.cpp", "task_graph.cpp"]
HEADERS = ["modarith.h", "ntt_core.h", "ntt_plan.h", "tables.h", "lsa_internal.h", "task_graph.h", "mini_json.h",
           "../../include/lattisense_amd.h", "../../include/lattisense_task.h"]


def torch_lib_dir():
    spec = importlib.util.find_spec("torch")
    if spec is None or spec.origin is None:
        return None
    d = os.path.join(os.path.dirname(spec.origin), "lib")
    return d if os.path.exists(os.path.join(d, "libamdhip64.so")) else None


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_native(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    srcs = [s for s in HIP_SOURCES + CXX_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    for src in srcs:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.replace(".", "_") + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + hdrs):
            if src.endswith(".hip"):
                cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                       "-fno-fast-math", "-Wall", "-Wno-unused-result"] + os.environ.get("LSA_EXTRA_FLAGS", "").split() + \
                      ["-c", sp, "-o", obj]
            else:
                cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-I/opt/rocm/include",
                       "-D__HIP_PLATFORM_AMD__", "-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        tl = torch_lib_dir()
        libdirs = ([tl] if tl else []) + ["/opt/rocm/lib"]
        cmd = ["g++", "-shared", "-o", LIB] + objs
        for d in libdirs:
            cmd += ["-L" + d, "-Wl,-rpath," + d]
        cmd += ["-lamdhip64", "-lpthread"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
