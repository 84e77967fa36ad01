"""Build container only: compiles the layout probe (tests/cpp/abi_probe.inc) against the REFERENCE's own type definitions
-- the type section of /root/reference/mega_ag_runners/mega_ag.h (the whole header cannot be included: it pulls the cgo
header of the absent Lattigo library, fhe_lib_v2.h:41), its vendored nlohmann/json.hpp, c_argument.h and abi/c_types.h --
and writes the offsets / sizes / enumerator values to tests/golden/abi_offsets.json.  tests/test_abi_layout.py holds this
project's structs to that table on every machine, and re-derives the table from the reference where it is present."""
import json
import os
import subprocess
import sys
import tempfile

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_table():
    src = open(os.path.join(REF, "mega_ag_runners", "mega_ag.h")).read().split("\n")
    a = next(i for i, l in enumerate(src) if l.startswith("using NodeIndex"))
    b = next(i for i, l in enumerate(src) if l.startswith("struct MegaAG"))
    with tempfile.TemporaryDirectory() as d:
        # the extracted lines stay in the temporary directory; only numbers leave it
        with open(os.path.join(d, "ref_types.h"), "w") as f:
            f.write("#include <any>\n#include <cstdint>\n#include <functional>\n#include <optional>\n#include <string>\n"
                    "#include <unordered_map>\n#include <vector>\n#include \"nlohmann/json.hpp\"\n#include \"c_argument.h\"\n"
                    "#include \"c_types.h\"\n" + "\n".join(src[a:b]) + "\n")
        with open(os.path.join(d, "probe.cpp"), "w") as f:
            f.write('#include "ref_types.h"\n#include "%s"\n' % os.path.join(ROOT, "tests", "cpp", "abi_probe.inc"))
        exe = os.path.join(d, "probe")
        subprocess.check_call(["g++", "-std=c++17", "-Wno-invalid-offsetof", "-I" + os.path.join(REF, "lib"),
                               "-I" + os.path.join(REF, "mega_ag_runners"), "-I" + os.path.join(REF, "abi"),
                               os.path.join(d, "probe.cpp"), "-o", exe])
        out = subprocess.check_output([exe], text=True)
    return {l.split()[0]: int(l.split()[1]) for l in out.splitlines()}


if __name__ == "__main__":
    t = reference_table()
    p = os.path.join(ROOT, "tests", "golden", "abi_offsets.json")
    json.dump({"source": "reference mega_ag.h type section + abi/c_types.h + c_argument.h, g++ -std=c++17, LP64, _GLIBCXX_USE_CXX11_ABI=1",
               "values": t}, open(p, "w"), indent=1, sort_keys=True)
    print(len(t), "entries ->", p)
