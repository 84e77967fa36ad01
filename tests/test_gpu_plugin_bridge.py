"""A plug-in-shaped caller through the C entry point its real callers use: bind_gpu_task_abi_bridge_executors with export /
import executors that live in ANOTHER shared object (tests/cpp/plugin_bridge_shim.cpp), built the way the SEAL and Lattigo
plug-ins build theirs -- malloc'd C structs with one malloc per limb behind std::shared_ptr<CCiphertext | CPlaintext |
CRelinKey | CGaloisKey> with freeing deleters, one Galois element per exported CGaloisKey, file-static per-run state, the
std::function objects destroyed right after bind; the import side any_casts the backend's std::shared_ptr<CCiphertext> across
the library boundary (reference: plug-in/SEAL/acc/abi_bridge_executors.h:70-179, plug-in/SEAL/acc/c_struct_import_export.h:93-142,
plug-in/lattigo/acc/abi_bridge_executors.cc:76-189, abi/c_structs.c:23-98).

Graphs: the reference's own GPU test graphs (tests/golden/ref_gpu_suite.tar.gz), both key digit shapes -- the default sets'
hybrid digits of k = 2 special primes and the custom N = 8192 sets' ONE special prime (level+1 single-prime digits, the SEAL
shape).  Every output is compared bit for bit with the oracle walk of the same graph (tests/ref_suite.py)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from tests import ref_suite as rs
from tests.gpu_util import need_gpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
c_u64p = ctypes.POINTER(ctypes.c_uint64)


@pytest.fixture(scope="module")
def suite(tmp_path_factory):
    return rs.unpack(str(tmp_path_factory.mktemp("ref_suite_plugin")))


@pytest.fixture(scope="module")
def plg(tmp_path_factory):
    need_gpu()
    from lattisense_amd import build
    libdir = os.path.dirname(build.LIB)
    so = str(tmp_path_factory.mktemp("plugin") / "libplugin_bridge_shim.so")
    tl = build.torch_lib_dir()
    cmd = ["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "plugin_bridge_shim.cpp"), "-o", so, "-L" + libdir, "-llattisense_amd",
           "-Wl,-rpath," + libdir]
    for r in ([tl] if tl else []) + ["/opt/rocm/lib"]:
        cmd += ["-L" + r, "-Wl,-rpath," + r]
    subprocess.check_call(cmd)
    from lattisense_amd.task import CArgument, _task_lib
    _task_lib()
    L = ctypes.CDLL(so)
    L.plg_ct_new.restype = ctypes.c_void_p
    L.plg_ct_new.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_u64p]
    L.plg_ct_read.argtypes = [ctypes.c_void_p, c_u64p]
    L.plg_ct_lie.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.plg_ct_free.argtypes = [ctypes.c_void_p]
    L.plg_pt_new.restype = ctypes.c_void_p
    L.plg_pt_new.argtypes = [ctypes.c_int, ctypes.c_int, c_u64p]
    L.plg_pt_free.argtypes = [ctypes.c_void_p]
    L.plg_ksk_new.restype = ctypes.c_void_p
    L.plg_ksk_new.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_u64p]
    L.plg_ksk_free.argtypes = [ctypes.c_void_p]
    L.plg_glk_new.restype = ctypes.c_void_p
    L.plg_glk_add.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_u64p]
    L.plg_glk_free.argtypes = [ctypes.c_void_p]
    L.plg_bind.argtypes = [ctypes.c_void_p]
    L.plg_run.restype = ctypes.c_int
    L.plg_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(CArgument), ctypes.c_uint64, ctypes.POINTER(CArgument),
                          ctypes.c_uint64, ctypes.c_int]
    L.plg_counters.argtypes = [ctypes.POINTER(ctypes.c_long)] * 4
    return L


def _counters(L):
    v = [ctypes.c_long() for _ in range(4)]
    L.plg_counters(*[ctypes.byref(x) for x in v])
    return dict(zip(("exports", "imports", "struct_frees", "limb_mallocs"), [x.value for x in v]))


def _ptr(a):
    return a.ctypes.data_as(c_u64p)


class PluginRun:
    """the plug-in's FheTaskGpu: create the task, bind ITS executors, run with handles to ITS objects"""

    def __init__(self, L, path):
        from lattisense_amd.task import _task_lib
        self.L, self.T = L, _task_lib()
        self.g = rs.load(path)
        self.h = self.T.create_fhe_gpu_task(str(path).encode())
        assert self.h, self.T.lsa_last_error().decode()
        L.plg_bind(self.h)
        self.free = []

    def close(self):
        for fn, h in self.free:
            fn(h)
        self.free = []
        if self.h:
            self.T.release_fhe_gpu_task(self.h)
            self.h = None

    def run(self, vals, keys, lie=None):
        """returns (rc, {output datum index: ndarray}); lie = (input datum index, level, degree, drop_limbs) for a negative test"""
        from lattisense_amd.task import (CArgument, TYPE_CIPHERTEXT, TYPE_GALOIS_KEY, TYPE_PLAINTEXT, TYPE_RELIN_KEY)
        L, g = self.L, self.g
        P, data = g["parameter"], g["data"]
        n, np_ = P["n"], len(P["p"])
        keep, cin = [], []

        def arg(name, ty, handle, level):
            arr = (ctypes.c_void_p * 1)(handle)
            idb = name.encode()
            keep.extend([arr, idb])
            return CArgument(idb, ty, ctypes.cast(arr, ctypes.c_void_p), level, 1)

        glk = None
        for idx in g["inputs"]:
            d = data[str(idx)]
            ty = d["type"]
            if ty in ("ct", "ct3"):
                v = np.ascontiguousarray(vals[idx])
                h = L.plg_ct_new(d["level"], d["degree"], n, _ptr(v))
                self.free.append((L.plg_ct_free, h))
                if lie and lie[0] == idx:
                    L.plg_ct_lie(h, lie[1], lie[2], lie[3])
                cin.append(arg(d["id"], TYPE_CIPHERTEXT, h, d["level"]))
            elif ty in ("pt", "pt_ringt"):
                v = np.ascontiguousarray(vals[idx])
                h = L.plg_pt_new(v.shape[0], n, _ptr(v))
                self.free.append((L.plg_pt_free, h))
                cin.append(arg(d["id"], TYPE_PLAINTEXT, h, v.shape[0] - 1))
            elif ty == "rlk":
                k, lvl = keys[idx]
                h = L.plg_ksk_new(lvl, np_, n, _ptr(np.ascontiguousarray(k)))
                self.free.append((L.plg_ksk_free, h))
                cin.append(arg("rlk_ntt", TYPE_RELIN_KEY, h, lvl))
            elif ty == "glk":
                if glk is None:       # ONE GaloisKeys object behind every Galois datum (cpu_task_utils.h:300-316)
                    glk = L.plg_glk_new()
                    self.free.append((L.plg_glk_free, glk))
                    cin.append(arg("glk_ntt", TYPE_GALOIS_KEY, glk, d["level"]))
                k, lvl = keys[idx]
                L.plg_glk_add(glk, d["galois_element"], lvl, np_, n, _ptr(np.ascontiguousarray(k)))
            else:
                raise NotImplementedError(ty)
        outs, cout = {}, []
        for idx in g["outputs"]:
            d = data[str(idx)]
            h = L.plg_ct_new(d["level"], d["degree"], n, None)
            self.free.append((L.plg_ct_free, h))
            outs[idx] = (h, (d["degree"] + 1, d["level"] + 1, n))
            cout.append(arg(d["id"], TYPE_CIPHERTEXT, h, d["level"]))
        a_in = (CArgument * len(cin))(*cin)
        a_out = (CArgument * len(cout))(*cout)
        rc = L.plg_run(self.h, n, a_in, len(cin), a_out, len(cout), 0)
        got = {}
        if rc == 0:
            for idx, (h, shape) in outs.items():
                buf = np.empty(shape, dtype=np.uint64)
                L.plg_ct_read(h, _ptr(buf))
                got[idx] = buf
        return rc, got


def _task(suite, tag, name, level):
    for ptag, nm, lv, path in rs.tasks(suite, tag):
        if nm == name and lv == level:
            return path
    raise AssertionError("no task %s/%s level %d in the reference suite" % (tag, name, level))


CASES = [
    # (parameter set, graph, level)                                        digit shape / what it exercises
    ("ckks_param_default_n16384", "CKKS_4_cmc_relin_rescale", 5),        # hybrid digits, k = 2 special primes; relin key
    ("ckks_param_custom_n8192", "CKKS_4_cmc_relin_rescale", 3),          # ONE special prime: level+1 single-prime digits (SEAL shape)
    ("ckks_param_custom_n8192", "CKKS_4_rotate_col/steps_1_to_8", 4),                  # Galois keys, one element per exported struct, k = 1
    ("ckks_param_default_n16384", "CKKS_4_rotate_col/steps_1_to_8", 3),                # Galois keys, hybrid digits
    ("bfv_param_default_n16384_t10001", "BFV_cmpac", 1),                  # ring-t plaintexts, multiply-accumulate
    ("ckks_param_default_n16384", "CKKS_cmpac_ringt", 3),
    ("bfv_param_custom_n8192_t10001", "BFV_4_cmc", 2),                    # degree-2 (ct3) outputs
    ("bfv_param_custom_n8192_t10001", "BFV_4_cmc_relin", 2),              # BFV relinearisation with single-prime digits
]


@pytest.mark.parametrize("tag,name,level", CASES)
def test_reference_graphs_through_foreign_bridge_executors(plg, suite, tag, name, level):
    path = _task(suite, tag, name, level)
    g = rs.load(path)
    o = rs.oracle_for(g)
    vals, keys = rs.random_inputs(g, o, np.random.default_rng(11))
    before = _counters(plg)
    run = PluginRun(plg, path)
    try:
        rc, got = run.run(vals, keys)
        assert rc == 0, run.T.lsa_last_error().decode()
    finally:
        run.close()
    want = rs.interpret(g, o, vals, keys, targets=g["outputs"])
    for idx in g["outputs"]:
        assert np.array_equal(got[idx], want[idx]), "output %s differs from the oracle" % g["data"][str(idx)]["id"]
    after = _counters(plg)
    assert after["exports"] - before["exports"] == len(g["inputs"])          # every input datum went through the plug-in's exporter
    assert after["imports"] - before["imports"] == len(g["outputs"])         # ... every output through its importer
    assert after["limb_mallocs"] > before["limb_mallocs"]                     # per-limb malloc'd structs, as abi/c_structs.c builds them
    assert after["struct_frees"] > before["struct_frees"]                     # and the backend released them through the plug-in's deleters


@pytest.mark.parametrize("lie,what", [((7, -1, 0), "level"), ((-1, 2, 0), "degree"), ((-1, -1, 1), "limb count")])
def test_a_struct_that_disagrees_with_the_task_is_refused_with_the_datum_named(plg, suite, lie, what):
    """LOAD sizes nothing from a caller's struct it has not checked: wrong level / degree / limb count -> LSA_ERR_ARG naming the input"""
    from lattisense_amd._native import lib
    path = _task(suite, "ckks_param_default_n16384", "CKKS_4_cmc_relin_rescale", 5)
    g = rs.load(path)
    o = rs.oracle_for(g)
    vals, keys = rs.random_inputs(g, o, np.random.default_rng(5))
    first_ct = next(i for i in g["inputs"] if g["data"][str(i)]["type"] == "ct")
    run = PluginRun(plg, path)
    try:
        rc, _ = run.run(vals, keys, lie=(first_ct,) + lie)
        msg = lib().lsa_last_error().decode()
    finally:
        run.close()
    assert rc == 1, (rc, msg)                                   # LSA_ERR_ARG
    assert g["data"][str(first_ct)]["id"] in msg and "datum %d" % first_ct in msg, msg
    # the handle is still usable afterwards: same task directory, honest structs
    run = PluginRun(plg, path)
    try:
        rc, got = run.run(vals, keys)
        assert rc == 0
    finally:
        run.close()
