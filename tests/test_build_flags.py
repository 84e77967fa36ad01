"""Build hygiene of the product library: it is compiled with none of the A/B / diagnostic switches, it says so through
lsa_build_flags(), the build ignores the environment, and a library that was built with switches cannot be loaded as the
product (only explicitly, through LSA_NATIVE_LIB).  CPU only: nothing here computes."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lattisense_amd", "csrc")


def test_product_library_reports_no_switches():
    from lattisense_amd import _native
    assert _native.build_flags() == ""


def test_every_switch_in_the_sources_is_reported():
    """a new #if defined(LSA_...) in a kernel source that build_flags.h does not know about would be invisible"""
    known = open(os.path.join(CSRC, "build_flags.h")).read()
    # not switches of the GPU product: the CPU replay harness, the in-tree ABI configuration, include guards / helper macros
    exempt = {"LSA_EMULATE", "LSA_WITH_NLOHMANN", "LSA_EMU_CHECK"}
    used = set()
    for f in os.listdir(CSRC):
        if not f.endswith((".hip", ".h", ".cpp")) or f == "build_flags.h":
            continue
        text = open(os.path.join(CSRC, f)).read()
        used |= set(re.findall(r"#\s*(?:if|ifdef|ifndef|elif)[^\n]*?\b(LSA_[A-Z0-9_]+)", text))
        used |= set(re.findall(r"defined\((LSA_[A-Z0-9_]+)\)", text))
    missing = sorted(m for m in used - exempt if m not in known)
    assert missing == [], missing


def test_the_product_build_ignores_the_environment():
    text = open(os.path.join(ROOT, "lattisense_amd", "build.py")).read()
    assert "LSA_EXTRA_FLAGS" not in text
    out = subprocess.run(["readelf", "-d", os.path.join(ROOT, "lattisense_amd", "liblattisense_amd.so")], capture_output=True, text=True)
    if out.returncode == 0:
        assert "liblattisense_amd.so" in "".join(l for l in out.stdout.splitlines() if "SONAME" in l)


def test_wrong_result_switches_need_the_diagnostic_umbrella(tmp_path):
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-DLSA_KS_DIAG_NO_MATH",
                        "-x", "hip", os.path.join(CSRC, "build_flags.h")], capture_output=True, text=True)
    assert r.returncode != 0 and "LSA_DIAG_BUILD" in r.stderr


def test_a_library_built_with_switches_is_refused_as_the_product(tmp_path):
    from lattisense_amd import build as nb
    nb.build_native()
    obj = str(tmp_path / "c_api_flagged.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-DLSA_AB_SWITCH=hygiene",
                           "-c", os.path.join(CSRC, "c_api.hip"), "-o", obj])
    objs = [obj if s == "c_api.hip" else os.path.join(CSRC, "build", s.replace(".", "_") + ".o")
            for s in nb.HIP_SOURCES + nb.CXX_SOURCES]
    lib = str(tmp_path / "liblattisense_amd.so")
    cmd = ["g++", "-shared", "-o", lib] + objs
    for d in ([nb.torch_lib_dir()] if nb.torch_lib_dir() else []) + ["/opt/rocm/lib"]:
        cmd += ["-L" + d, "-Wl,-rpath," + d]
    subprocess.check_call(cmd + ["-lamdhip64", "-lpthread"])
    code = "import lattisense_amd._native as n; n.LIB_PATH=%r; n.lib(); print('FLAGS', n.build_flags())" % lib
    env = {k: v for k, v in os.environ.items() if k != "LSA_NATIVE_LIB"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=env)
    assert r.returncode != 0 and "not the product build" in r.stderr and "LSA_AB_SWITCH=hygiene" in r.stderr
    # selected explicitly (what the A/B tools do) it loads and says what it is
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=dict(env, LSA_NATIVE_LIB=lib))
    assert r.returncode == 0 and "LSA_AB_SWITCH=hygiene" in r.stdout, r.stderr
