"""CPU-only: the task-graph loader (lattisense_amd/csrc/task_graph.cpp) under AddressSanitizer + UBSan.

1. every one of the 1252 graphs of the reference's GPU suite (tests/golden/ref_gpu_suite.tar.gz) loads with no report;
2. damaged graphs -- truncated files, dropped keys, wrong value kinds, dangling / negative / huge node indices, cycles --
   end in an exception (the C-ABI turns it into create_fhe_gpu_task() == NULL + lsa_last_error()), never in a sanitizer
   report or a crash.  The reference's loader (mega_ag_runners/mega_ag.cpp:125-657) throws on the same conditions it checks.
"""
import copy
import json
import os
import random
import subprocess

import pytest

from tests import ref_suite as rs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lattisense_amd", "csrc")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("san") / "graph_loader_san")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           os.path.join(ROOT, "tests", "cpp", "graph_loader_san.cpp"), os.path.join(CSRC, "task_graph.cpp"), "-o", exe])
    return exe


@pytest.fixture(scope="module")
def suite(tmp_path_factory):
    return rs.unpack(str(tmp_path_factory.mktemp("ref_suite")))


def run(exe, paths, tmp):
    lst = os.path.join(tmp, "list.txt")
    with open(lst, "w") as f:
        f.write("\n".join(paths) + "\n")
    out = subprocess.run([exe, lst], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert out.returncode == 0, (out.stdout[-2000:] + out.stderr[-4000:])
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-4000:]
    lines = out.stdout.strip().split("\n")
    assert lines[-1].startswith("DONE"), lines[-1]
    return lines[:-1]


def test_every_reference_graph_loads_clean(driver, suite, tmp_path):
    paths = [os.path.join(p, "mega_ag.json") for _, _, _, p in rs.tasks(suite)]
    res = run(driver, paths, str(tmp_path))
    assert len(res) == 1252
    bad = [(p, r) for p, r in zip(paths, res) if not r.startswith("ok ")]
    assert not bad, bad[:3]


def _mutations(g, rng):
    """damaged copies of one graph, each with a label"""
    first_c = sorted(g["compute"], key=int)[0]
    first_d = sorted(g["data"], key=int)[0]
    out = []

    def m(label, fn):
        h = copy.deepcopy(g)
        fn(h)
        out.append((label, json.dumps(h)))

    for key in ("inputs", "outputs", "data", "compute", "parameter", "algorithm"):
        m("drop_" + key, lambda h, key=key: h.pop(key))
    m("algo_unknown", lambda h: h.__setitem__("algorithm", "TFHE"))
    m("algo_number", lambda h: h.__setitem__("algorithm", 7))
    m("inputs_string", lambda h: h.__setitem__("inputs", "0,1"))
    m("inputs_dangling", lambda h: h["inputs"].append(10 ** 9))
    m("inputs_negative", lambda h: h["inputs"].append(-1))
    m("outputs_dangling", lambda h: h.__setitem__("outputs", [2 ** 31 - 1]))
    m("outputs_huge", lambda h: h.__setitem__("outputs", [2 ** 63]))
    m("outputs_float", lambda h: h.__setitem__("outputs", [1.5]))
    m("compute_in_dangling", lambda h: h["compute"][first_c]["inputs"].append(123456789))
    m("compute_in_negative", lambda h: h["compute"][first_c].__setitem__("inputs", [-5]))
    m("compute_out_dangling", lambda h: h["compute"][first_c].__setitem__("outputs", [987654321]))
    m("compute_no_outputs", lambda h: h["compute"][first_c].__setitem__("outputs", []))
    m("compute_no_inputs", lambda h: h["compute"][first_c].__setitem__("inputs", []))
    m("compute_type_unknown", lambda h: h["compute"][first_c].__setitem__("type", "frobnicate"))
    m("compute_type_number", lambda h: h["compute"][first_c].__setitem__("type", 3))
    m("compute_is_list", lambda h: h.__setitem__("compute", list(h["compute"].values())))
    m("compute_key_not_number", lambda h: h["compute"].__setitem__("abc", h["compute"][first_c]))
    m("compute_self_loop", lambda h: h["compute"][first_c].__setitem__("inputs", list(h["compute"][first_c]["outputs"])))
    m("datum_type_unknown", lambda h: h["data"][first_d].__setitem__("type", "ct9"))
    m("datum_level_string", lambda h: h["data"][first_d].__setitem__("level", "three"))
    m("datum_level_negative", lambda h: h["data"][first_d].__setitem__("level", -7))
    m("datum_level_huge", lambda h: h["data"][first_d].__setitem__("level", 2 ** 40))
    m("datum_drop_type", lambda h: h["data"][first_d].pop("type"))
    m("datum_drop_id", lambda h: h["data"][first_d].pop("id", None))
    m("datum_is_null", lambda h: h["data"].__setitem__(first_d, None))
    m("data_key_not_number", lambda h: h["data"].__setitem__("x1", h["data"][first_d]))
    m("two_producers", lambda h: [c.__setitem__("outputs", list(h["compute"][first_c]["outputs"])) for c in h["compute"].values()])
    m("custom_without_type", lambda h: (h["compute"][first_c].__setitem__("is_custom", True), h["compute"][first_c].pop("type")))
    m("custom_flag_string", lambda h: h["compute"][first_c].__setitem__("is_custom", "yes"))
    # operand / result relations: what the runtime sizes its slabs and launches from (task_graph.cpp, validate_structure)
    def fhe_nodes(h):
        return [c for _, c in sorted(h["compute"].items(), key=lambda kv: int(kv[0])) if not c.get("is_custom")]

    def first_of(h, *types):
        for c in fhe_nodes(h):
            if c["type"] in types:
                return c
        return None

    def out_of(h, c):
        return h["data"][str(c["outputs"][0])]

    def in_of(h, c, i=0):
        return h["data"][str(c["inputs"][i])]

    any_op = first_of(g, "add", "sub", "neg", "mult", "relin", "rotate_col", "rotate_row", "cmp_sum", "cmpac_sum")
    if any_op:
        ty = any_op["type"]
        m("result_level_lower_than_operands", lambda h: out_of(h, first_of(h, ty)).__setitem__("level", max(0, in_of(h, first_of(h, ty))["level"] - 1)
                                                                                                if in_of(h, first_of(h, ty))["level"] > 0 else in_of(h, first_of(h, ty))["level"] + 1))
        m("result_degree_wrong", lambda h: out_of(h, first_of(h, ty)).__setitem__("degree", (out_of(h, first_of(h, ty))["degree"] + 1) % 3))
    two_ct = next((c for c in fhe_nodes(g) if c["type"] in ("add", "sub", "mult") and len(c["inputs"]) == 2 and g["data"][str(c["inputs"][1])]["type"] == "ct"
                   and g["data"][str(c["inputs"][1])]["level"] > 0 and str(c["inputs"][1]) != str(c["inputs"][0])), None)
    if two_ct:
        idx = str(two_ct["inputs"][1])
        m("second_operand_lower_level", lambda h: h["data"][idx].__setitem__("level", h["data"][idx]["level"] - 1))
    rs_node = first_of(g, "rescale")
    if rs_node:
        m("rescale_keeps_level", lambda h: out_of(h, first_of(h, "rescale")).__setitem__("level", in_of(h, first_of(h, "rescale"))["level"]))
    keyed = first_of(g, "relin", "rotate_col", "rotate_row")
    if keyed and in_of(g, keyed)["level"] > 0:
        kt = keyed["type"]
        m("key_below_ciphertext_level", lambda h: in_of(h, first_of(h, kt), 1).__setitem__("level", in_of(h, first_of(h, kt))["level"] - 1))
    if g["parameter"].get("max_level") is not None and len(g["parameter"].get("q", [])) > 1:
        m("level_beyond_max_level", lambda h: (h["parameter"].__setitem__("max_level", 0), None))
    # a cycle between the first two compute nodes
    if len(g["compute"]) >= 2:
        a, b = sorted(g["compute"], key=int)[:2]

        def cyc(h):
            h["compute"][a]["inputs"] = list(h["compute"][b]["outputs"])
            h["compute"][b]["inputs"] = list(h["compute"][a]["outputs"])
        m("cycle", cyc)
    text = json.dumps(g)
    for k in range(12):
        cut = rng.randrange(1, len(text))
        out.append(("truncate_%d" % cut, text[:cut]))
    for k in range(12):
        pos = rng.randrange(len(text))
        out.append(("garble_%d" % pos, text[:pos] + rng.choice('{}[]",:x-9e') + text[pos + 1:]))
    out.append(("empty", ""))
    out.append(("deep_nesting", "[" * 100000))
    out.append(("not_object", "[1,2,3]"))
    out.append(("bad_escape", '{"a": "\\u12"}'))
    out.append(("big_number", '{"inputs": [1e999999]}'))
    return out


def test_damaged_graphs_are_refused_not_crashed_on(driver, suite, tmp_path):
    rng = random.Random(20260)
    picks = {}
    for ptag, name, lv, path in rs.tasks(suite):
        base = name.split("/")[0]
        if base in ("CKKS_4_cmc_relin_rescale", "BFV_custom_compute_in_middle", "CKKS_4_cmpac", "BFV_4_advanced_rotate_col",
                    "CKKS_4_rotate_col", "BFV_braid") and base not in picks:
            picks[base] = path
    assert len(picks) >= 4, sorted(picks)
    paths, labels = [], []
    for base, path in sorted(picks.items()):
        g = json.load(open(os.path.join(path, "mega_ag.json")))
        for label, text in _mutations(g, rng):
            d = tmp_path / ("%s__%s" % (base, label))
            d.mkdir()
            (d / "mega_ag.json").write_text(text)
            paths.append(str(d / "mega_ag.json"))
            labels.append(base + ":" + label)
    paths.append(str(tmp_path / "does_not_exist" / "mega_ag.json"))
    labels.append("missing_file")
    res = run(driver, paths, str(tmp_path))
    assert len(res) == len(paths)
    verdict = dict(zip(labels, res))
    # structural damage must be refused; single-character garbling may by luck still be a valid graph
    must_fail = [l for l in labels if not l.split(":")[-1].startswith(("garble_",))]
    accepted = [l for l in must_fail if verdict[l].startswith("ok ")]
    assert not accepted, sorted({a.split(":")[1] for a in accepted})
    assert verdict["missing_file"].startswith("err Cannot open MegaAG file")
