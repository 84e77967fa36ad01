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
HEADERS = ["build_flags.h", "modarith.h", "ntt_core.h", "ntt_r16.h", "ntt_plan.h", "tables.h", "lsa_internal.h", "task_graph.h", "mini_json.h", "buf_pool.h",
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
                       "-fno-fast-math", "-Wall", "-Wno-unused-result", "-c", sp, "-o", obj]   # no -D switches, whatever the environment says
            else:
                cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-I/opt/rocm/include",
                       "-D__HIP_PLATFORM_AMD__", "-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        tl = torch_lib_dir()
        libdirs = ([tl] if tl else []) + ["/opt/rocm/lib"]
        cmd = ["g++", "-shared", "-Wl,-soname,liblattisense_amd.so", "-o", LIB] + objs
        for d in libdirs:
            cmd += ["-L" + d, "-Wl,-rpath," + d]
        cmd += ["-lamdhip64", "-lpthread"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


def build_variant(name, flags, verbose=False):
    """A/B builds for measurement: the same library with extra compile flags, as lattisense_amd/variants/lib<name>.so
    (selected at run time with LSA_NATIVE_LIB=<path>; diagnostics only -- tests and bench use the default build).
    Objects are cached per variant; only sources whose text mentions one of the -D names are recompiled with the flags,
    plus the three that report the flag string (csrc/build_flags.h).  Wrong-result diagnostics (*_DIAG_NO_*, *_COPY_ONLY,
    *_COMPUTE_ONLY) get -DLSA_DIAG_BUILD and the library its own SONAME: it cannot stand in for the product."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    build_native()
    diag = any(t in flags for t in ("_DIAG_NO_", "_DIAG_COPY_ONLY", "_DIAG_COMPUTE_ONLY", "_DIAG_TW8"))
    if diag and "-DLSA_DIAG_BUILD" not in flags:
        flags = flags + " -DLSA_DIAG_BUILD"
    flags = flags + " -DLSA_VARIANT_NAME=" + name
    vdir = os.path.join(HERE, "variants")
    objdir = os.path.join(CSRC, "build", "variant_" + name)
    os.makedirs(vdir, exist_ok=True)
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    hdr_text = "".join(open(h).read() for h in hdrs if os.path.exists(h))
    macros = [f[2:].split("=")[0] for f in flags.split() if f.startswith("-D")]
    objs = []
    for src in [s for s in HIP_SOURCES + CXX_SOURCES if os.path.exists(os.path.join(CSRC, s))]:
        sp = os.path.join(CSRC, src)
        base_obj = os.path.join(CSRC, "build", src.replace(".", "_") + ".o")
        text = open(sp).read()
        affected = src.endswith(".hip") and (src in ("kernels.hip", "context.hip", "c_api.hip") or
                                             any(m in text for m in macros))
        if not affected:
            objs.append(base_obj)
            continue
        obj = os.path.join(objdir, src.replace(".", "_") + ".o")
        objs.append(obj)
        stamp = obj + ".flags"
        if _stale(obj, [sp] + hdrs) or not os.path.exists(stamp) or open(stamp).read() != flags:
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall",
                   "-Wno-unused-result"] + flags.split() + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
            open(stamp, "w").write(flags)
    out = os.path.join(vdir, "lib%s.so" % name)
    tl = torch_lib_dir()
    cmd = ["g++", "-shared", "-Wl,-soname,liblattisense_amd_%s_%s.so" % ("diag" if diag else "variant", name), "-o", out] + objs
    for d in ([tl] if tl else []) + ["/opt/rocm/lib"]:
        cmd += ["-L" + d, "-Wl,-rpath," + d]
    cmd += ["-lamdhip64", "-lpthread"]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2], verbose=True))
    else:
        print(build_native(force="--force" in sys.argv, verbose=True))
