"""The C++ types that cross the executor boundary (ExecutorFunc, ExecutionContext, DatumNode, ComputeNode, OperationType:
/root/reference/mega_ag_runners/mega_ag.h:40-177) and the C structs of the data ABI (abi/c_types.h:26-60, c_argument.h:26-46)
keep the reference's layout: every offset, size and enumerator value is compared with tests/golden/abi_offsets.json (derived
from the reference's own header by tools/gen_abi_offsets.py).  Where /root/reference is present (build container) the table
is re-derived and must agree, and this project's header built with -DLSA_WITH_NLOHMANN (attributes = nlohmann::json, the
in-tree configuration) must match INCLUDING the sizes that depend on the JSON type."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = json.load(open(os.path.join(ROOT, "tests", "golden", "abi_offsets.json")))["values"]
REF_LIB = "/root/reference/lib"


def _ours(tmp_path, extra):
    src = tmp_path / "probe.cpp"
    src.write_text('#include "%s"\n#include "%s"\n' % (os.path.join(ROOT, "lattisense_amd", "csrc", "task_graph.h"),
                                                       os.path.join(ROOT, "tests", "cpp", "abi_probe.inc")))
    exe = str(tmp_path / "probe")
    subprocess.check_call(["g++", "-std=c++17", "-Wno-invalid-offsetof"] + extra + [str(src), "-o", exe])
    out = subprocess.check_output([exe], text=True)
    return {l.split()[0]: int(l.split()[1]) for l in out.splitlines()}


def test_stand_alone_build_matches_the_reference_layout(tmp_path):
    ours = _ours(tmp_path, [])
    assert set(ours) == set(TABLE)
    for k, v in TABLE.items():
        if k.startswith("ATTR."):
            continue   # sizes behind `attributes` (the last member): equal only in the in-tree configuration below
        assert ours[k] == v, (k, ours[k], v)
    # private enumerators stay outside the reference's value range
    assert max(v for k, v in TABLE.items() if k.startswith("OperationType.")) == TABLE["OperationType.STORE_FROM_BACKEND"] < 1000


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_LIB, "nlohmann", "json.hpp")), reason="reference tree not present")
def test_in_tree_configuration_is_layout_identical(tmp_path):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_abi_offsets
    assert gen_abi_offsets.reference_table() == TABLE            # the committed table IS the reference's layout
    ours = _ours(tmp_path, ["-DLSA_WITH_NLOHMANN", "-I" + REF_LIB])
    assert ours == TABLE                                         # every offset AND sizeof, attributes included
    # ... and the loader that fills nlohmann-typed attributes from the parsed task file compiles in that configuration
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-DLSA_WITH_NLOHMANN", "-I" + REF_LIB, "-I/opt/rocm/include",
                           "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "lattisense_amd", "csrc", "task_graph.cpp")])
