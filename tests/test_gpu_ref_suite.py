"""GPU parity over the reference's OWN GPU test graphs (tests/golden/ref_gpu_suite.tar.gz, generated from
/root/reference/unittests/test_gpu_bfv.py and test_gpu_ckks.py by tools/gen_ref_suite.sh): every graph shape of
unittests/test_gpu_bfv.cpp:36-1303 and test_gpu_ckks.cpp:50-760 at EVERY level of every parameter set the reference's
conftest enumerates (default N=16384 and N=8192 sets, the custom chains with ONE special prime, the 2048-slot variant) runs
through run_fhe_gpu_task with inputs of the declared shapes, and every output is compared bit for bit with the CPU oracle
walked over the same graph (1226 graphs, 6887 compute nodes, 83 s; tools/ref_suite_full.py prints the per-set table).  The custom-node graphs (custom_cmpac, custom_compute_at_start / in_middle / at_end, test_gpu_bfv.cpp:1087-1303) run
with executors bound through bind_gpu_task_custom_executors and are also checked at message level, as the reference does."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from tests import ref_suite as rs
from tests.gpu_util import need_gpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def suite(tmp_path_factory):
    return rs.unpack(str(tmp_path_factory.mktemp("ref_suite")))


def _run_tag(suite, tag, levels="all"):
    need_gpu()
    per_name = {}
    for ptag, name, lv, path in rs.tasks(suite, tag):
        per_name.setdefault(name, []).append((lv, path))
    ran = 0
    for name, items in sorted(per_name.items()):
        lvls = sorted({lv for lv, _ in items})
        keep = set(lvls) if levels == "all" else {lvls[0], lvls[-1]}
        for lv, path in items:
            if lv not in keep:
                continue
            g = rs.load(path)
            if rs.is_custom(g) or rs.has_type(g, "bootstrap"):
                continue   # custom nodes: below; bootstrapping: tests/test_gpu_bootstrap.py
            rs.run_and_compare(path, seed=ran)   # every output
            ran += 1
    return ran


def test_bfv_default_n16384_every_shape_every_level(suite):
    assert _run_tag(suite, "bfv_param_default_n16384_t10001") >= 120


def test_ckks_default_n16384_every_shape_every_level(suite):
    assert _run_tag(suite, "ckks_param_default_n16384") >= 550


@pytest.mark.parametrize("tag", ["bfv_param_custom_n8192_t10001", "bfv_param_default_n8192_t10001", "ckks_param_custom_n8192",
                                 "ckks_param_default_n8192", "ckks_param_default_n16384_slots2048"])
def test_other_parameter_sets_every_shape_every_level(suite, tag):
    assert _run_tag(suite, tag) >= 49


# ------------------------------------------------------------------------------------------------ custom executors
@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    """tests/cpp/custom_exec_shim.cpp: wraps a C callback into the std::function executors bind_gpu_task_custom_executors takes"""
    need_gpu()
    from lattisense_amd import build
    libdir = os.path.dirname(build.LIB)
    so = str(tmp_path_factory.mktemp("shim") / "libcustom_exec_shim.so")
    tl = build.torch_lib_dir()
    cmd = ["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "custom_exec_shim.cpp"), "-o", so, "-L" + libdir, "-llattisense_amd",
           "-Wl,-rpath," + libdir]
    for r in ([tl] if tl else []) + ["/opt/rocm/lib"]:
        cmd += ["-L" + r, "-Wl,-rpath," + r]
    subprocess.check_call(cmd)
    L = ctypes.CDLL(so)
    L.lsa_test_bind_custom.restype = ctypes.c_int
    return L


CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int,
                      ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p)


class _Custom:
    """the test's custom executors (the roles of the lambdas at test_gpu_bfv.cpp:1104-1131, 1228-1236): value-level functions
    for the oracle walk, and the handle-level callback the runtime calls from its CPU-pool threads"""

    def __init__(self, o, client):
        self.o, self.c, self.keep, self.calls = o, client, [], []

    # value level
    def encode_ringt(self, msg):
        return self.c.bfv_encode(msg)[None]

    def encode(self, msg, level):
        return self.o.bfv_scale_up(level, self.c.bfv_encode(msg))

    def custom_add(self, ct):
        return rs._limbwise(self.o, "add", ct, ct)

    def ops(self):
        return {"encode_ringt": lambda node, ins: self.encode_ringt(ins[0]),
                "encode": lambda node, ins: self.encode(ins[0], node["attributes"]["level"]),
                "custom_add": lambda node, ins: self.custom_add(ins[0])}

    # handle level
    def callback(self):
        from lattisense_amd.task import Ciphertext, CustomData, HostCiphertext, Plaintext

        def cb(ty, node_id, attr_level, ins, n_in, out, _user):
            try:
                ty = ty.decode()
                self.calls.append(ty)
                if ty in ("encode_ringt", "encode"):
                    v = ctypes.cast(ins[0], ctypes.POINTER(CustomData._View)).contents
                    msg = np.ctypeslib.as_array(v.data, shape=(v.n,)).copy()
                    obj = Plaintext(self.encode_ringt(msg) if ty == "encode_ringt" else self.encode(msg, attr_level))
                else:
                    h = ctypes.cast(ins[0], ctypes.POINTER(HostCiphertext)).contents
                    ct = np.ctypeslib.as_array(h.data, shape=(h.degree + 1, h.level + 1, h.n)).copy()
                    obj = Ciphertext(self.custom_add(ct))
                self.keep.append(obj)       # the handles stay alive until the test ends
                out[0] = ctypes.addressof(obj.h)
                return 0
            except Exception as e:          # never let a Python exception cross the C boundary
                print("custom executor failed:", repr(e))
                return 1
        return CB(cb)


def _bind(shim, task, types, cb):
    arr = (ctypes.c_char_p * len(types))(*[t.encode() for t in types])
    assert shim.lsa_test_bind_custom(ctypes.c_void_p(task.h), arr, len(types), cb, None) == 0


def _run_custom(suite, shim, name, lv, message_level):
    from lattisense_amd.task import CustomData, FheTaskGpu
    from oracle.client import Client, galois_element_for_col_rotation
    path = os.path.join(suite, "bfv_param_default_n16384_t10001", name, "level_%d" % lv)
    g = rs.load(path)
    o = rs.oracle_for(g)
    P, data = g["parameter"], g["data"]
    n, t_mod = P["n"], P["t"]
    c = Client(o, seed=lv)
    rng = np.random.default_rng(100 + lv)
    vals, keys = rs.random_inputs(g, o, rng)
    custom, msgs = {}, {}
    for idx in g["inputs"]:
        d = data[str(idx)]
        if d["type"] in ("ct",) and message_level:
            msgs[idx] = rng.integers(0, t_mod, size=n, dtype=np.uint64)
            vals[idx] = c.bfv_encrypt(msgs[idx], d["level"])
        elif d["type"] not in ("ct", "ct3", "pt", "pt_ringt", "rlk", "glk", "swk"):
            msgs[idx] = rng.integers(0, t_mod, size=n, dtype=np.uint64)
            vals[idx] = msgs[idx]
            custom[idx] = CustomData(msgs[idx])
    if message_level:   # decryptable keys
        for idx in list(keys):
            d = data[str(idx)]
            keys[idx] = ((c.gen_relin_key(d["level"]) if d["type"] == "rlk" else c.gen_galois_key(d["galois_element"], d["level"])), d["level"])
    ex = _Custom(o, c)
    cb = ex.callback()
    ins, outs, out_cts = rs.arguments(g, vals, keys, custom)
    task = FheTaskGpu(path)
    _bind(shim, task, ["encode_ringt", "encode", "custom_add"], cb)
    try:
        task.run(ins, outs)
    finally:
        task.close()
    want = rs.interpret(g, o, vals, keys, custom_ops=ex.ops())
    for idx, ct in zip(g["outputs"], out_cts):
        assert np.array_equal(ct.data, want[idx]), (name, lv)
    n_custom = sum(1 for cn in g["compute"].values() if cn.get("is_custom"))
    assert len(ex.calls) == n_custom            # every custom node ran exactly once, through the bound executors
    return g, c, msgs, out_cts, t_mod


def test_custom_nodes_at_start(suite, shim):
    """custom_cmpac / custom_compute_at_start: 7 messages -> encode_ringt -> ct x pt_ringt, summed, + encode(message 8)"""
    for name in ("BFV_custom_cmpac", "BFV_custom_compute_at_start"):
        for lv in (1, 3, 5):
            g, c, msgs, outs, t = _run_custom(suite, shim, name, lv, message_level=(lv == 1))
            if lv == 1:     # z = sum_i x_i * y_i + y_7 mod t   (test_gpu_bfv.cpp:1138-1145)
                data = g["data"]
                xs = [i for i in g["inputs"] if data[str(i)]["type"] == "ct"]
                ys = [i for i in g["inputs"] if data[str(i)]["type"] not in ("ct", "rlk", "glk")]
                exp = np.zeros(len(msgs[xs[0]]), dtype=np.uint64)
                for xi, yi in zip(xs, ys[:7]):
                    exp = (exp + msgs[xi] * msgs[yi]) % np.uint64(t)
                exp = (exp + msgs[ys[7]]) % np.uint64(t)
                assert np.array_equal(c.bfv_decrypt(outs[0].data), exp)


def test_custom_node_at_end(suite, shim):
    """custom_compute_at_end: relin(mult(x, y)) on the device, stored, imported into an intermediate handle, custom_add on the
    host, returned as the task output: z_i = 2 x_i y_i (test_gpu_bfv.cpp:1215-1252)"""
    for lv in (1, 4):
        g, c, msgs, outs, t = _run_custom(suite, shim, "BFV_custom_compute_at_end", lv, message_level=(lv == 1))
        if lv == 1:
            data = g["data"]
            cts = [i for i in g["inputs"] if data[str(i)]["type"] == "ct"]
            xs, ys = cts[: len(cts) // 2], cts[len(cts) // 2:]
            for k, out in enumerate(outs):
                exp = (msgs[xs[k]] * msgs[ys[k]] * np.uint64(2)) % np.uint64(t)
                assert np.array_equal(c.bfv_decrypt(out.data), exp)


def test_custom_node_in_the_middle(suite, shim):
    """custom_compute_in_middle: device (mult, relin) -> host (custom_add) -> device (rotate_cols by -990 = NAF 32 + 2 - 1024,
    add tree): STORE / IMPORT / custom / EXPORT / LOAD in the middle of the graph (test_gpu_bfv.cpp:1254-1303)"""
    for lv in (1, 5):
        g, c, msgs, outs, t = _run_custom(suite, shim, "BFV_custom_compute_in_middle", lv, message_level=(lv == 1))
        if lv == 1:
            data = g["data"]
            cts = [i for i in g["inputs"] if data[str(i)]["type"] == "ct"]
            xs, ys = cts[: len(cts) // 2], cts[len(cts) // 2:]
            n = len(msgs[xs[0]])
            h = n // 2
            exp = np.zeros(n, dtype=np.uint64)
            for xi, yi in zip(xs, ys):
                d = (msgs[xi] * msgs[yi] * np.uint64(2)) % np.uint64(t)
                exp = (exp + np.concatenate([np.roll(d[:h], 990), np.roll(d[h:], 990)])) % np.uint64(t)
            assert np.array_equal(c.bfv_decrypt(outs[0].data), exp)


def test_unbound_custom_executor_fails_loudly(suite):
    need_gpu()
    from lattisense_amd._native import LsaError
    from lattisense_amd.task import CustomData, FheTaskGpu
    path = os.path.join(suite, "bfv_param_default_n16384_t10001", "BFV_custom_cmpac", "level_1")
    g = rs.load(path)
    o = rs.oracle_for(g)
    rng = np.random.default_rng(5)
    vals, keys = rs.random_inputs(g, o, rng)
    custom = {i: CustomData(rng.integers(0, 65537, size=g["parameter"]["n"], dtype=np.uint64)) for i in g["inputs"]
              if g["data"][str(i)]["type"] not in ("ct", "rlk", "glk")}
    ins, outs, _ = rs.arguments(g, vals, keys, custom)
    t = FheTaskGpu(path)
    with pytest.raises(LsaError, match="no executor bound"):
        t.run(ins, outs)
    t.close()


# ------------------------------------------------------------------------------------------------ pools
def test_same_task_after_a_pool_flush(suite):
    """one task handle, gpu_device=0, run - flush every pooled device / pinned buffer - run again: the (device, lane)-keyed
    pools re-allocate on the owning device and the results are the same bits (reference: one task object on any device,
    README.md:195-202; the keying itself is unit-tested on the CPU, tests/test_buf_pool.py)"""
    need_gpu()
    from lattisense_amd.task import FheTaskGpu
    path = os.path.join(suite, "ckks_param_default_n16384", "CKKS_4_cmc_relin_rescale", "level_5")
    g = rs.load(path)
    o = rs.oracle_for(g)
    vals, keys = rs.random_inputs(g, o, np.random.default_rng(9))
    want = rs.interpret(g, o, vals, keys)
    t = FheTaskGpu(path)
    for round_ in range(3):
        ins, outs, out_cts = rs.arguments(g, vals, keys)
        t.run(ins, outs, gpu_device=0)
        for idx, ct in zip(g["outputs"], out_cts):
            assert np.array_equal(ct.data, want[idx]), round_
        if round_ == 0:
            t.trim_pools()
    t.close()
