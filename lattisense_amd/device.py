"""Host-side mirror of the operator surface the reference's GPU executors use (heongpu::HEContext +
HEArithmeticOperator as called from mega_ag_runners/gpu/mega_ag_executors_gpu.cu:71-426), over the C-ABI.

Buffers are device-resident u64 limb arrays laid out [batch][poly][limb][N]; NumPy is used only to move test data in
and out.  No arithmetic happens in Python.
"""
import ctypes

import numpy as np

from . import _native
from ._native import check, lib

ALGO_BFV, ALGO_CKKS = 0, 1


class DeviceBuffer:
    def __init__(self, ctx, nwords):
        self.ctx = ctx
        self.nwords = int(nwords)
        p = ctypes.c_void_p()
        check(lib().lsa_malloc(ctx.h, ctypes.byref(p), self.nwords * 8))
        self.ptr = p.value

    def free(self):
        if self.ptr:
            check(lib().lsa_free(self.ctx.h, self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceContext:
    """One parameter set on one GPU (replaces init_gpu_context, gpu_wrapper.cu:53-138; tables are cached)."""

    def __init__(self, algo, n, q, p, t=0, device=0):
        self.algo, self.n, self.q, self.p, self.t = algo, int(n), list(q), list(p), int(t)
        qa = (ctypes.c_uint64 * len(q))(*q)
        pa = (ctypes.c_uint64 * max(1, len(p)))(*p)
        h = ctypes.c_void_p()
        check(lib().lsa_context_create(algo, self.n, qa, len(q), pa, len(p), self.t, device, ctypes.byref(h)))
        self.h = h
        self.stream = None  # default (null) stream unless the caller sets one
        cnt = ctypes.c_int()
        out = (ctypes.c_uint64 * 256)()
        check(lib().lsa_context_moduli(self.h, out, 256, ctypes.byref(cnt)))
        self.moduli = [int(out[i]) for i in range(cnt.value)]

    def close(self):
        if self.h:
            check(lib().lsa_context_destroy(self.h))
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- memory
    def alloc(self, nwords):
        return DeviceBuffer(self, nwords)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        buf = self.alloc(arr.size)
        check(lib().lsa_memcpy_h2d(self.h, buf.ptr, arr.ctypes.data, arr.nbytes, self.stream))
        self.sync()
        return buf

    def download(self, buf, shape):
        out = np.empty(shape, dtype=np.uint64)
        assert out.size <= buf.nwords
        check(lib().lsa_memcpy_d2h(self.h, out.ctypes.data, buf.ptr, out.nbytes, self.stream))
        return out

    def sync(self):
        check(lib().lsa_stream_synchronize(self.h, self.stream))

    def set_tile_batch(self, tb):
        check(lib().lsa_set_tile_batch(self.h, tb))

    def set_fp64_ntt(self, enable):
        check(lib().lsa_set_fp64_ntt(self.h, int(enable)))

    # ---- keys
    def upload_key(self, compact, key_level):
        compact = np.ascontiguousarray(compact, dtype=np.uint64)
        assert compact.nbytes == lib().lsa_key_bytes(self.h, key_level), "key shape does not match its level"
        k = ctypes.c_void_p()
        check(lib().lsa_key_upload(self.h, compact.ctypes.data, key_level, self.stream, ctypes.byref(k)))
        return k

    def adopt_key(self, dev_ptr, key_level):
        k = ctypes.c_void_p()
        check(lib().lsa_key_adopt_device(self.h, dev_ptr, key_level, self.stream, ctypes.byref(k)))
        return k

    def key_bytes(self, key_level):
        return lib().lsa_key_bytes(self.h, key_level)

    def destroy_key(self, k):
        check(lib().lsa_key_destroy(self.h, k))

    # ---- operators (all batched; strides in u64 elements)
    def ntt(self, buf, batch, rows, mod_of, inverse=False, batch_stride=None):
        mo = (ctypes.c_int * len(mod_of))(*mod_of)
        bs = rows * self.n if batch_stride is None else batch_stride
        check(lib().lsa_ntt(self.h, buf.ptr, batch, bs, rows, mo, len(mod_of), int(inverse), self.stream))

    def addsub(self, op, level, polys, a, b, batch):
        L = level + 1
        out = self.alloc(batch * polys * L * self.n)
        s = polys * L * self.n
        check(lib().lsa_poly_addsub(self.h, op, level, polys, a.ptr, b.ptr if b is not None else None, out.ptr,
                                    batch, s, s, s, self.stream))
        return out

    def ckks_mult(self, level, a, b, batch):
        L = level + 1
        out = self.alloc(batch * 3 * L * self.n)
        check(lib().lsa_ckks_mult(self.h, level, a.ptr, b.ptr, out.ptr, batch, 2 * L * self.n, 2 * L * self.n,
                                  3 * L * self.n, self.stream))
        return out

    def ckks_relin(self, level, d3, rlk, batch):
        L = level + 1
        out = self.alloc(batch * 2 * L * self.n)
        check(lib().lsa_ckks_relin(self.h, level, d3.ptr, rlk, out.ptr, batch, 3 * L * self.n, 2 * L * self.n,
                                   self.stream))
        return out

    def ckks_rescale(self, level, polys, ct, batch):
        L = level + 1
        out = self.alloc(batch * polys * level * self.n)
        check(lib().lsa_ckks_rescale(self.h, level, polys, ct.ptr, out.ptr, batch, polys * L * self.n,
                                     polys * level * self.n, self.stream))
        return out

    def ckks_rotate_many(self, level, ct, keys, batch):
        """keys: {galois element: key handle}; returns {element: device buffer}, one decomposition for all (hoisted)"""
        L = level + 1
        outs = {g: self.alloc(batch * 2 * L * self.n) for g in keys}
        els = (ctypes.c_uint64 * len(keys))(*keys.keys())
        hk = (ctypes.c_void_p * len(keys))(*[k.value for k in keys.values()])
        po = (ctypes.c_void_p * len(keys))(*[o.ptr for o in outs.values()])
        check(lib().lsa_ckks_rotate_many(self.h, level, ct.ptr, len(keys), els, hk, po, batch, 2 * L * self.n, 2 * L * self.n,
                                         self.stream))
        return outs

    def ckks_rotate(self, level, ct, g, glk, batch):
        L = level + 1
        out = self.alloc(batch * 2 * L * self.n)
        check(lib().lsa_ckks_rotate(self.h, level, ct.ptr, g, glk, out.ptr, batch, 2 * L * self.n, 2 * L * self.n,
                                    self.stream))
        return out

    def drop_level(self, level, polys, ct, batch):
        L = level + 1
        out = self.alloc(batch * polys * level * self.n)
        check(lib().lsa_drop_level(self.h, level, polys, ct.ptr, out.ptr, batch, polys * L * self.n,
                                   polys * level * self.n, self.stream))
        return out

    def ckks_mult_relin_rescale(self, level, a, b, rlk, batch, out=None):
        L = level + 1
        if out is None:
            out = self.alloc(batch * 2 * level * self.n)
        check(lib().lsa_ckks_mult_relin_rescale(self.h, level, a.ptr, b.ptr, rlk, out.ptr, batch, 2 * L * self.n,
                                                2 * L * self.n, 2 * level * self.n, self.stream))
        return out

    def bfv_mult(self, level, a, b, batch):
        L = level + 1
        out = self.alloc(batch * 3 * L * self.n)
        check(lib().lsa_bfv_mult(self.h, level, a.ptr, b.ptr, out.ptr, batch, 2 * L * self.n, 2 * L * self.n,
                                 3 * L * self.n, self.stream))
        return out

    def bfv_relin(self, level, d3, rlk, batch):
        L = level + 1
        out = self.alloc(batch * 2 * L * self.n)
        check(lib().lsa_bfv_relin(self.h, level, d3.ptr, rlk, out.ptr, batch, 3 * L * self.n, 2 * L * self.n,
                                  self.stream))
        return out

    def bfv_rotate(self, level, ct, g, glk, batch):
        L = level + 1
        out = self.alloc(batch * 2 * L * self.n)
        check(lib().lsa_bfv_rotate(self.h, level, ct.ptr, g, glk, out.ptr, batch, 2 * L * self.n, 2 * L * self.n,
                                   self.stream))
        return out

    def bfv_rescale(self, level, polys, ct, batch):
        L = level + 1
        out = self.alloc(batch * polys * level * self.n)
        check(lib().lsa_bfv_rescale(self.h, level, polys, ct.ptr, out.ptr, batch, polys * L * self.n,
                                    polys * level * self.n, self.stream))
        return out

    def bfv_mult_relin(self, level, a, b, rlk, batch, out=None):
        L = level + 1
        if out is None:
            out = self.alloc(batch * 2 * L * self.n)
        check(lib().lsa_bfv_mult_relin(self.h, level, a.ptr, b.ptr, rlk, out.ptr, batch, 2 * L * self.n,
                                       2 * L * self.n, 2 * L * self.n, self.stream))
        return out


class BootstrapPlan:
    """Device-side CKKS bootstrapping plan (include/lattisense_amd.h: lsa_bootstrap_*)."""

    def __init__(self, ctx, cts_depth=4, stc_depth=3, k=16, double_angle=3, message_ratio=256.0, in_scale=2.0 ** 40,
                 out_scale=0.0, log_slots=0, sine_deg=30, arcsine_deg=0):
        self.ctx = ctx
        h = ctypes.c_void_p()
        check(lib().lsa_bootstrap_create_ex(ctx.h, cts_depth, stc_depth, k, double_angle, message_ratio, in_scale, out_scale,
                                            log_slots, sine_deg, arcsine_deg, ctx.stream, ctypes.byref(h)))
        self.h = h
        lv, sc, ng, nm, nc, sp = ctypes.c_int(), ctypes.c_double(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib().lsa_bootstrap_info(self.h, ctypes.byref(lv), ctypes.byref(sc), ctypes.byref(ng), ctypes.byref(nm),
                                       ctypes.byref(nc), ctypes.byref(sp)))
        self.out_level, self.out_scale, self.n_matrices = lv.value, sc.value, nm.value
        self.sparse, self.n_cts = bool(sp.value), nc.value
        g = (ctypes.c_uint64 * ng.value)()
        check(lib().lsa_bootstrap_galois_elements(self.h, g, ng.value))
        self.galois_elements = [int(x) for x in g]
        # double hoisting (the default): baby-step / giant-step matrices carry plaintext rows for the special primes too
        self.double_hoist = False
        for i in range(self.n_matrices):
            lvl, rows = ctypes.c_int(), ctypes.c_int()
            check(lib().lsa_bootstrap_matrix_info(self.h, i, ctypes.byref(lvl), None, None, None, 0))
            check(lib().lsa_bootstrap_plaintext_rows(self.h, i, ctypes.byref(rows)))
            self.double_hoist = self.double_hoist or rows.value > lvl.value + 1

    def close(self):
        if self.h:
            lib().lsa_bootstrap_destroy(self.h)
            self.h = None

    def chebyshev(self):
        return self.evalmod_constants()[0]

    def evalmod_constants(self):
        """(Chebyshev coefficients of the cosine interpolant, monomial coefficients of the arcsine correction or None)"""
        nc, na = ctypes.c_int(), ctypes.c_int()
        check(lib().lsa_bootstrap_evalmod_constants(self.h, ctypes.byref(nc), None, ctypes.byref(na), None))
        c = (ctypes.c_double * nc.value)()
        a = (ctypes.c_double * max(na.value, 1))()
        check(lib().lsa_bootstrap_evalmod_constants(self.h, None, c, None, a))
        return np.array(c[:], dtype=np.float64), (np.array(a[: na.value], dtype=np.float64) if na.value else None)

    def matrix(self, index):
        """(level, n1 (0: no baby-step/giant-step), diagonal indices, {k: plaintext [rows][N]}); rows = level+1, or level+1+k
        (residues at the special primes too) for a baby-step / giant-step matrix of a double-hoisting plan"""
        lv, n1, nd = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib().lsa_bootstrap_matrix_info(self.h, index, ctypes.byref(lv), ctypes.byref(n1), ctypes.byref(nd), None, 0))
        ks = (ctypes.c_int * nd.value)()
        check(lib().lsa_bootstrap_matrix_info(self.h, index, None, None, None, ks, nd.value))
        plains = {}
        rows = ctypes.c_int()
        check(lib().lsa_bootstrap_plaintext_rows(self.h, index, ctypes.byref(rows)))
        for i, k in enumerate(ks):
            pt = np.empty((rows.value, self.ctx.n), dtype=np.uint64)
            check(lib().lsa_bootstrap_plaintext_ext(self.h, index, i, pt.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), pt.size))
            plains[int(k)] = pt
        return lv.value, n1.value, [int(k) for k in ks], plains

    def run(self, in_buf, batch, rlk, glk, swk_dts=None, swk_std=None):
        """in_buf: device [batch][2][1][N]; glk: {galois element: key handle}; returns device [batch][2][out_level+1][N]"""
        n = self.ctx.n
        out = self.ctx.alloc(batch * 2 * (self.out_level + 1) * n)
        elts = (ctypes.c_uint64 * len(glk))(*glk.keys())
        keys = (ctypes.c_void_p * len(glk))(*[k.value for k in glk.values()])
        check(lib().lsa_ckks_bootstrap(self.ctx.h, self.h, in_buf.ptr, out.ptr, batch, 2 * n, 2 * (self.out_level + 1) * n, rlk,
                                       len(glk), elts, keys, swk_dts, swk_std, self.ctx.stream))
        return out

    def oracle_plains(self):
        """the plan's encoded diagonals keyed as oracle/ckks_bootstrap.py expects them"""
        keys = [("cts", i) for i in range(self.n_cts)] + ([("p1",), ("p2",)] if self.sparse else [])
        keys += [("stc", i) for i in range(self.n_matrices - len(keys))]
        return {k: self.matrix(i)[3] for i, k in enumerate(keys)}

    def oracle_levels(self):
        """level of every matrix, keyed like oracle_plains()"""
        keys = [("cts", i) for i in range(self.n_cts)] + ([("p1",), ("p2",)] if self.sparse else [])
        keys += [("stc", i) for i in range(self.n_matrices - len(keys))]
        out = {}
        for i, k in enumerate(keys):
            lv = ctypes.c_int()
            check(lib().lsa_bootstrap_matrix_info(self.h, i, ctypes.byref(lv), None, None, None, 0))
            out[k] = lv.value
        return out
