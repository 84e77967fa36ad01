"""Host-side mirror of the reference SDK's GPU task class (lattisense::FheTaskGpu, cxx_sdk_v2/cxx_fhe_task.h:132-148,
cxx_fhe_task_gpu.cpp:30-117) on top of the task-layer C-ABI (include/lattisense_task.h).

The reference SDK feeds Lattigo `Handle`s; this front-end's objects are plain host limb buffers (NumPy arrays), the
analogue of plug-in/SEAL/acc for callers without a host crypto library.  Names, argument meaning and error behaviour
follow the reference: FheTaskGpu(project_path), run(args, progress_cb=None, gpu_device=0) -> elapsed nanoseconds;
a missing task file or an unsupported operation raises at construction.
"""
import ctypes
import time

import numpy as np

from ._native import LsaError, lib

c_u64p = ctypes.POINTER(ctypes.c_uint64)

TYPE_PLAINTEXT, TYPE_CIPHERTEXT, TYPE_RELIN_KEY, TYPE_GALOIS_KEY, TYPE_SWITCH_KEY, TYPE_CUSTOM = range(6)


class CArgument(ctypes.Structure):
    _fields_ = [("id", ctypes.c_char_p), ("type", ctypes.c_int), ("data", ctypes.c_void_p), ("level", ctypes.c_int),
                ("size", ctypes.c_int)]


class HostCiphertext(ctypes.Structure):
    _fields_ = [("level", ctypes.c_int), ("degree", ctypes.c_int), ("n", ctypes.c_int), ("data", c_u64p)]


class HostPlaintext(ctypes.Structure):
    _fields_ = [("level", ctypes.c_int), ("n", ctypes.c_int), ("data", c_u64p)]


class HostKsKey(ctypes.Structure):
    _fields_ = [("level", ctypes.c_int), ("n_special", ctypes.c_int), ("n", ctypes.c_int), ("data", c_u64p)]


class HostGaloisKey(ctypes.Structure):
    _fields_ = [("n_keys", ctypes.c_int), ("galois_elements", c_u64p), ("keys", ctypes.POINTER(HostKsKey))]


PROGRESS_CB = ctypes.CFUNCTYPE(None, ctypes.c_int, ctypes.c_int, ctypes.c_void_p)

_task_sigs_done = False


def _task_lib():
    global _task_sigs_done
    L = lib()
    if not _task_sigs_done:
        L.create_fhe_gpu_task.restype = ctypes.c_void_p
        L.create_fhe_gpu_task.argtypes = [ctypes.c_char_p]
        L.release_fhe_gpu_task.restype = None
        L.release_fhe_gpu_task.argtypes = [ctypes.c_void_p]
        L.run_fhe_gpu_task.restype = ctypes.c_int
        L.run_fhe_gpu_task.argtypes = [ctypes.c_void_p, ctypes.POINTER(CArgument), ctypes.c_uint64,
                                       ctypes.POINTER(CArgument), ctypes.c_uint64, PROGRESS_CB, ctypes.c_void_p,
                                       ctypes.c_int]
        L.lsa_frontend_bind.restype = ctypes.c_int
        L.lsa_frontend_bind.argtypes = [ctypes.c_void_p]
        L.lsa_task_trim_pools.restype = ctypes.c_int
        L.lsa_task_trim_pools.argtypes = [ctypes.c_void_p]
        L.lsa_task_set_devices.restype = ctypes.c_int
        L.lsa_task_set_devices.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
        L.lsa_task_last_run_shards.restype = ctypes.c_int
        L.lsa_task_last_run_shards.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int)] * 3
        L.lsa_host_register.restype = ctypes.c_int
        L.lsa_host_register.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.lsa_host_alloc.restype = ctypes.c_int
        L.lsa_host_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
        L.lsa_host_free.restype = ctypes.c_int
        L.lsa_host_free.argtypes = [ctypes.c_void_p]
        L.lsa_host_unregister.restype = ctypes.c_int
        L.lsa_host_unregister.argtypes = [ctypes.c_void_p]
        L.lsa_task_last_run_direct.restype = ctypes.c_int
        L.lsa_task_last_run_direct.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int)] * 2
        L.lsa_task_drop_keys.restype = ctypes.c_int
        L.lsa_task_drop_keys.argtypes = [ctypes.c_void_p]
        L.lsa_task_last_run_keys.restype = ctypes.c_int
        L.lsa_task_last_run_keys.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int)] * 2
        L.lsa_task_counts.restype = ctypes.c_int
        L.lsa_task_counts.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int)] * 4
        L.lsa_task_last_run_stats.restype = ctypes.c_int
        L.lsa_task_last_run_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                              ctypes.POINTER(ctypes.c_double)]
        _task_sigs_done = True
    return L


def register_host(array):
    """pins a NumPy array's buffer in place (lsa_host_register): ciphertexts / plaintexts whose limbs lie inside it are copied
    to and from the device without staging.  Keep the array alive until unregister_host(array)."""
    L = _task_lib()
    rc = L.lsa_host_register(ctypes.c_void_p(array.ctypes.data), array.nbytes)
    if rc:
        raise LsaError(rc, L.lsa_last_error().decode())


def alloc_host(shape):
    """a uint64 NumPy array in pinned memory allocated by the library (lsa_host_alloc): ciphertexts built on it are copied to and
    from the device without staging, at full PCIe rate.  Release with free_host(array) once nothing uses it."""
    L = _task_lib()
    n = int(np.prod(shape))
    p = ctypes.c_void_p()
    rc = L.lsa_host_alloc(n * 8, ctypes.byref(p))
    if rc:
        raise LsaError(rc, L.lsa_last_error().decode())
    buf = (ctypes.c_uint64 * n).from_address(p.value)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(shape)
    return a


def free_host(array):
    L = _task_lib()
    rc = L.lsa_host_free(ctypes.c_void_p(array.ctypes.data))
    if rc:
        raise LsaError(rc, L.lsa_last_error().decode())


def unregister_host(array):
    L = _task_lib()
    rc = L.lsa_host_unregister(ctypes.c_void_p(array.ctypes.data))
    if rc:
        raise LsaError(rc, L.lsa_last_error().decode())


class Ciphertext:
    """Host ciphertext: array [degree+1][level+1][N] of uint64 residues (BFV: coefficient domain, CKKS: NTT domain)."""

    def __init__(self, data):
        self.data = np.ascontiguousarray(data, dtype=np.uint64)
        assert self.data.ndim == 3
        self.h = HostCiphertext(self.data.shape[1] - 1, self.data.shape[0] - 1, self.data.shape[2],
                                self.data.ctypes.data_as(c_u64p))

    @classmethod
    def empty(cls, degree, level, n):
        return cls(np.zeros((degree + 1, level + 1, n), dtype=np.uint64))


class Plaintext:
    def __init__(self, data):
        self.data = np.ascontiguousarray(data, dtype=np.uint64)
        assert self.data.ndim == 2
        self.h = HostPlaintext(self.data.shape[0] - 1, self.data.shape[1], self.data.ctypes.data_as(c_u64p))


class KeySwitchKey:
    """compact [beta][2][level+1+np][N], NTT domain, non-Montgomery (plug-in/lattigo/acc/c_struct_import_export.go:41-135)"""

    def __init__(self, data, level, n_special):
        self.data = np.ascontiguousarray(data, dtype=np.uint64)
        assert self.data.ndim == 4 and self.data.shape[2] == level + 1 + n_special
        self.level, self.n_special = level, n_special
        self.h = HostKsKey(level, n_special, self.data.shape[3], self.data.ctypes.data_as(c_u64p))


class GaloisKey:
    def __init__(self, keys):
        """keys: {galois_element: KeySwitchKey}"""
        self.keys = dict(keys)
        n = len(self.keys)
        self._elts = (ctypes.c_uint64 * max(n, 1))(*self.keys.keys())
        self._arr = (HostKsKey * max(n, 1))(*[k.h for k in self.keys.values()])
        self.h = HostGaloisKey(n, ctypes.cast(self._elts, c_u64p), ctypes.cast(self._arr, ctypes.POINTER(HostKsKey)))


class CustomData:
    """Opaque caller data of a custom node input (CustomData, cxx_argument.h:72,96-99): never reaches the device; the custom
    executors the caller binds receive the handle (here: the address of `.h`, a {n, data} view of a uint64 vector)."""

    class _View(ctypes.Structure):
        _fields_ = [("n", ctypes.c_int), ("data", c_u64p)]

    def __init__(self, values):
        self.data = np.ascontiguousarray(values, dtype=np.uint64)
        self.h = CustomData._View(self.data.size, self.data.ctypes.data_as(c_u64p))


class Argument:
    """One task argument: id + list of host objects (CxxVectorArgument, cxx_argument.h:108-133)."""

    def __init__(self, arg_id, objects):
        self.id = arg_id
        self.objects = list(objects)

    def _type(self):
        o = self.objects[0]
        return {Ciphertext: TYPE_CIPHERTEXT, Plaintext: TYPE_PLAINTEXT, KeySwitchKey: TYPE_RELIN_KEY,
                GaloisKey: TYPE_GALOIS_KEY, CustomData: TYPE_CUSTOM}[type(o)]

    def to_c(self, keep):
        n = len(self.objects)
        arr = (ctypes.c_void_p * n)(*[ctypes.addressof(o.h) for o in self.objects])
        keep.append(arr)
        idb = self.id.encode()
        keep.append(idb)
        lvl = getattr(self.objects[0].h, "level", 0)
        return CArgument(idb, self._type(), ctypes.cast(arr, ctypes.c_void_p), lvl, n)


class FheTaskGpu:
    def __init__(self, project_path):
        L = _task_lib()
        self.h = L.create_fhe_gpu_task(str(project_path).encode())
        if not self.h:
            raise RuntimeError(L.lsa_last_error().decode())
        rc = L.lsa_frontend_bind(self.h)
        if rc:
            raise LsaError(rc, L.lsa_last_error().decode())

    def close(self):
        if self.h:
            _task_lib().release_fhe_gpu_task(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def trim_pools(self):
        """frees the task's pooled device / pinned buffers on every device (lsa_task_trim_pools)"""
        L = _task_lib()
        rc = L.lsa_task_trim_pools(self.h)
        if rc:
            raise LsaError(rc, L.lsa_last_error().decode())

    def set_devices(self, device_ids):
        """spread the runs of this task over these devices (an index may repeat: two logical shards on one device); [] clears
        the list (lsa_task_set_devices, include/lattisense_task.h)"""
        L = _task_lib()
        ids = list(device_ids)
        arr = (ctypes.c_int * max(len(ids), 1))(*ids)
        rc = L.lsa_task_set_devices(self.h, arr, len(ids))
        if rc:
            raise LsaError(rc, L.lsa_last_error().decode())

    def last_run_shards(self):
        v = [ctypes.c_int() for _ in range(3)]
        _task_lib().lsa_task_last_run_shards(self.h, *[ctypes.byref(x) for x in v])
        return dict(zip(("shards", "chunks", "key_peer_copies"), [x.value for x in v]))

    def drop_keys(self):
        """frees the evaluation keys kept on the device(s) between runs (lsa_task_drop_keys); the next run uploads them again"""
        L = _task_lib()
        rc = L.lsa_task_drop_keys(self.h)
        if rc:
            raise LsaError(rc, L.lsa_last_error().decode())

    def last_run_direct(self):
        a, b = ctypes.c_int(), ctypes.c_int()
        _task_lib().lsa_task_last_run_direct(self.h, ctypes.byref(a), ctypes.byref(b))
        return {"loads": a.value, "stores": b.value}

    def last_run_keys(self):
        a, b = ctypes.c_int(), ctypes.c_int()
        _task_lib().lsa_task_last_run_keys(self.h, ctypes.byref(a), ctypes.byref(b))
        return {"uploaded": a.value, "reused": b.value}

    def counts(self):
        v = [ctypes.c_int() for _ in range(4)]
        _task_lib().lsa_task_counts(self.h, *[ctypes.byref(x) for x in v])
        return dict(zip(("data", "compute", "inputs", "outputs"), [x.value for x in v]))

    def last_run_stats(self):
        a, b, ms = ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
        _task_lib().lsa_task_last_run_stats(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(ms))
        return {"gpu_nodes": a.value, "gpu_batches": b.value, "run_ms": ms.value}

    def run(self, inputs, outputs, progress_cb=None, gpu_device=0):
        """inputs: [Argument...] in task order, evaluation keys appended last as the SDK does (ids rlk_ntt / glk_ntt,
        cxx_argument.h:178-260); outputs: [Argument...] of pre-allocated ciphertexts.  Returns elapsed nanoseconds."""
        L = _task_lib()
        keep = []
        cin = (CArgument * len(inputs))(*[a.to_c(keep) for a in inputs])
        cout = (CArgument * len(outputs))(*[a.to_c(keep) for a in outputs])
        cb = PROGRESS_CB(lambda done, total, _u: progress_cb(done, total)) if progress_cb else PROGRESS_CB()
        t0 = time.perf_counter_ns()
        rc = L.run_fhe_gpu_task(self.h, cin, len(inputs), cout, len(outputs), cb, None, gpu_device)
        dt = time.perf_counter_ns() - t0
        if rc:
            raise LsaError(rc, L.lsa_last_error().decode())
        return dt
