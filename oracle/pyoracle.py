"""ctypes binding of the CPU oracle (oracle/ls_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under lattisense_amd/ may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libls_oracle.so")

u64p = ctypes.POINTER(ctypes.c_uint64)


def build(force=False):
    src = os.path.join(_HERE, "ls_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def build_fast(out_dir):
    """the timing build of the same source (-O3 -march=native) compiled on THIS machine into out_dir; returns its path.
    (A Barrett modmul in place of the 128-bit remainder was tried for this build and dropped: on the hosts measured the
    hardware 128/64 division pipelines better -- 6 ns against 17 ns per NTT butterfly.)"""
    out = os.path.join(out_dir, "libls_oracle_fast.so")
    subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-ffp-contract=off", "-fno-fast-math",
                           "-shared", "-o", out, os.path.join(_HERE, "ls_oracle.c")])
    return out


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        # LS_ORACLE_LIB: bench.py's cpu_baseline leg points this at the -O3 -march=native timing build of the same source
        L = ctypes.CDLL(os.environ.get("LS_ORACLE_LIB") or _LIB_PATH)
        L.ora_ctx_new.restype = ctypes.c_void_p
        L.ora_ctx_new.argtypes = [ctypes.c_int, u64p, ctypes.c_int, u64p, ctypes.c_int, ctypes.c_uint64]
        L.ora_ctx_free.argtypes = [ctypes.c_void_p]
        L.ora_is_prime.restype = ctypes.c_int
        L.ora_is_prime.argtypes = [ctypes.c_uint64]
        L.ora_primitive_root.restype = ctypes.c_uint64
        L.ora_primitive_root.argtypes = [ctypes.c_uint64]
        L.ora_psi.restype = ctypes.c_uint64
        L.ora_psi.argtypes = [ctypes.c_uint64, ctypes.c_int]
        L.ora_bfv_aux_count.restype = ctypes.c_int
        L.ora_bfv_aux_count.argtypes = [u64p, ctypes.c_int, ctypes.c_int]
        L.ora_gen_aux_primes.restype = ctypes.c_int
        L.ora_gen_aux_primes.argtypes = [ctypes.c_int, ctypes.c_int, u64p, ctypes.c_int, u64p]
        vp, i, u = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64
        for name, args in {
            "ora_ntt": [vp, i, u64p],
            "ora_intt": [vp, i, u64p],
            "ora_vec_add": [vp, i, u64p, u64p, u64p],
            "ora_vec_sub": [vp, i, u64p, u64p, u64p],
            "ora_vec_neg": [vp, i, u64p, u64p],
            "ora_vec_mul": [vp, i, u64p, u64p, u64p],
            "ora_automorph_ntt": [vp, u, u64p, u64p],
            "ora_automorph_coeff": [vp, i, u, u64p, u64p],
            "ora_keyswitch": [vp, i, u64p, u64p, i, u64p, u64p],
            "ora_gadget_product": [vp, i, u64p, u64p, i, u64p, u64p],
            "ora_moddown": [vp, i, u64p, u64p],
            "ora_ckks_mult": [vp, i, u64p, u64p, u64p],
            "ora_ckks_relin": [vp, i, u64p, u64p, i, u64p],
            "ora_ckks_rescale": [vp, i, u64p, i, u64p],
            "ora_ckks_rotate": [vp, i, u64p, u, u64p, i, u64p],
            "ora_ckks_mult_relin_rescale": [vp, i, u64p, u64p, u64p, i, u64p],
            "ora_bfv_mult": [vp, i, u64p, u64p, u64p],
            "ora_bfv_relin": [vp, i, u64p, u64p, i, u64p],
            "ora_bfv_rotate": [vp, i, u64p, u, u64p, i, u64p],
            "ora_bfv_rescale": [vp, i, u64p, i, u64p],
            "ora_bfv_mult_relin": [vp, i, u64p, u64p, u64p, i, u64p],
            "ora_ckks_plain_ringt": [vp, i, i, i, u64p, u64p, u64p],
            "ora_bfv_plain_ringt": [vp, i, i, i, u64p, u64p, u64p],
            "ora_bfv_scale_up": [vp, i, u64p, u64p],
        }.items():
            getattr(L, name).argtypes = args
            getattr(L, name).restype = None
        L.ora_baseconv.argtypes = [vp, ctypes.POINTER(i), i, ctypes.POINTER(i), i,
                                   ctypes.POINTER(u64p), ctypes.POINTER(u64p), i]
        L.ora_baseconv.restype = None
        _lib = L
    return _lib


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def _arr(vals):
    return (ctypes.c_uint64 * len(vals))(*[int(v) for v in vals])


class Oracle:
    """One parameter set: N, Q chain, special primes P, optional BFV plaintext modulus t."""

    def __init__(self, n, q, p, t=0):
        self.n, self.q, self.p, self.t = int(n), [int(x) for x in q], [int(x) for x in p], int(t)
        self.nq, self.np_ = len(q), len(p)
        L = lib()
        self.h = L.ora_ctx_new(self.n, _arr(q), len(q), _arr(p), len(p), self.t)
        if not self.h:
            raise RuntimeError("ora_ctx_new failed")
        self.aux = []
        if t:
            cnt = L.ora_bfv_aux_count(_arr(q), len(q), self.n.bit_length() - 1)
            out = (ctypes.c_uint64 * cnt)()
            avoid = self.q + self.p
            L.ora_gen_aux_primes(self.n, cnt, _arr(avoid), len(avoid), out)
            self.aux = [int(x) for x in out]
        self.mod = self.q + self.p + self.aux

    def __del__(self):
        try:
            lib().ora_ctx_free(self.h)
        except Exception:
            pass

    def mod_index_p(self, i):
        return self.nq + i

    # --- single limb
    def ntt(self, mi, a):
        r = np.ascontiguousarray(a, dtype=np.uint64).copy()
        lib().ora_ntt(self.h, mi, _p(r))
        return r

    def intt(self, mi, a):
        r = np.ascontiguousarray(a, dtype=np.uint64).copy()
        lib().ora_intt(self.h, mi, _p(r))
        return r

    def vec(self, op, mi, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        r = np.empty_like(a)
        if op == "neg":
            lib().ora_vec_neg(self.h, mi, _p(a), _p(r))
        else:
            b = np.ascontiguousarray(b, dtype=np.uint64)
            getattr(lib(), "ora_vec_" + op)(self.h, mi, _p(a), _p(b), _p(r))
        return r

    def baseconv(self, sidx, didx, src, centered):
        src = np.ascontiguousarray(src, dtype=np.uint64)
        ns, nd = len(sidx), len(didx)
        dst = np.zeros((nd, self.n), dtype=np.uint64)
        sp = (u64p * ns)(*[_p(src[i]) for i in range(ns)])
        dp = (u64p * nd)(*[_p(dst[j]) for j in range(nd)])
        lib().ora_baseconv(self.h, (ctypes.c_int * ns)(*sidx), ns, (ctypes.c_int * nd)(*didx), nd, sp, dp,
                           int(centered))
        return dst

    def automorph_ntt(self, g, a):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        r = np.empty_like(a)
        lib().ora_automorph_ntt(self.h, g, _p(a), _p(r))
        return r

    def automorph_coeff(self, mi, g, a):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        r = np.empty_like(a)
        lib().ora_automorph_coeff(self.h, mi, g, _p(a), _p(r))
        return r

    # --- ciphertext-level ops.  ct arrays: [poly][L][N] uint64, keys: [beta][2][klvl+1+np][N]
    def keyswitch(self, lvl, cx, key, klvl):
        L = lvl + 1
        cx = np.ascontiguousarray(cx, dtype=np.uint64)
        o0 = np.empty((L, self.n), dtype=np.uint64)
        o1 = np.empty((L, self.n), dtype=np.uint64)
        lib().ora_keyswitch(self.h, lvl, _p(cx), _p(key), klvl, _p(o0), _p(o1))
        return o0, o1

    def gadget_product(self, lvl, cx, key, klvl):
        """the key switch without its division by P: [2][lvl+1+np][N] over Q_lvl u P, NTT domain"""
        cx = np.ascontiguousarray(cx, dtype=np.uint64)
        acc = np.empty((2, lvl + 1 + len(self.p), self.n), dtype=np.uint64)
        lib().ora_gadget_product(self.h, lvl, _p(cx), _p(key), klvl, _p(acc[0]), _p(acc[1]))
        return acc

    def moddown(self, lvl, ext):
        """[polys][lvl+1+np][N] over Q_lvl u P (NTT domain) -> [polys][lvl+1][N]: rounded division by P"""
        ext = np.array(ext, dtype=np.uint64, order="C")   # a copy: the P rows are transformed in place
        out = np.empty((ext.shape[0], lvl + 1, self.n), dtype=np.uint64)
        for h in range(ext.shape[0]):
            lib().ora_moddown(self.h, lvl, _p(ext[h]), _p(out[h]))
        return out

    def ckks_mult(self, lvl, a, b):
        d = np.empty((3, lvl + 1, self.n), dtype=np.uint64)
        lib().ora_ckks_mult(self.h, lvl, _p(a), _p(b), _p(d))
        return d

    def ckks_relin(self, lvl, d3, rlk, klvl):
        o = np.empty((2, lvl + 1, self.n), dtype=np.uint64)
        lib().ora_ckks_relin(self.h, lvl, _p(d3), _p(rlk), klvl, _p(o))
        return o

    def ckks_rescale(self, lvl, ct):
        npoly = ct.shape[0]
        o = np.empty((npoly, lvl, self.n), dtype=np.uint64)
        lib().ora_ckks_rescale(self.h, lvl, _p(ct), npoly, _p(o))
        return o

    def ckks_rotate(self, lvl, ct, g, glk, klvl):
        o = np.empty((2, lvl + 1, self.n), dtype=np.uint64)
        lib().ora_ckks_rotate(self.h, lvl, _p(ct), g, _p(glk), klvl, _p(o))
        return o

    def ckks_mult_relin_rescale(self, lvl, a, b, rlk, klvl):
        o = np.empty((2, lvl, self.n), dtype=np.uint64)
        lib().ora_ckks_mult_relin_rescale(self.h, lvl, _p(a), _p(b), _p(rlk), klvl, _p(o))
        return o

    def bfv_mult(self, lvl, a, b):
        d = np.empty((3, lvl + 1, self.n), dtype=np.uint64)
        lib().ora_bfv_mult(self.h, lvl, _p(a), _p(b), _p(d))
        return d

    def bfv_relin(self, lvl, d3, rlk, klvl):
        o = np.empty((2, lvl + 1, self.n), dtype=np.uint64)
        lib().ora_bfv_relin(self.h, lvl, _p(d3), _p(rlk), klvl, _p(o))
        return o

    def bfv_rotate(self, lvl, ct, g, glk, klvl):
        o = np.empty((2, lvl + 1, self.n), dtype=np.uint64)
        lib().ora_bfv_rotate(self.h, lvl, _p(ct), g, _p(glk), klvl, _p(o))
        return o

    def bfv_rescale(self, lvl, ct):
        npoly = ct.shape[0]
        o = np.empty((npoly, lvl, self.n), dtype=np.uint64)
        lib().ora_bfv_rescale(self.h, lvl, _p(ct), npoly, _p(o))
        return o

    def plain_ringt(self, op, lvl, ct, pt):
        """op 0 add, 1 sub, 2 mul with a ring-t plaintext limb (CKKS: mod q_0, BFV: mod t)"""
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        pt = np.ascontiguousarray(pt, dtype=np.uint64)
        o = np.empty_like(ct)
        fn = lib().ora_bfv_plain_ringt if self.t else lib().ora_ckks_plain_ringt
        fn(self.h, op, lvl, ct.shape[0], _p(ct), _p(pt), _p(o))
        return o

    def bfv_scale_up(self, lvl, pt):
        """ring-t message limb (mod t) -> the full BFV plaintext of level `lvl` ([lvl+1][N], coefficient domain, scaled by Q/t
        with rounding: Lattigo scaleUp) that ct + pt adds to c0"""
        pt = np.ascontiguousarray(pt, dtype=np.uint64)
        o = np.empty((lvl + 1, self.n), dtype=np.uint64)
        lib().ora_bfv_scale_up(self.h, lvl, _p(pt), _p(o))
        return o

    def bfv_mult_relin(self, lvl, a, b, rlk, klvl):
        o = np.empty((2, lvl + 1, self.n), dtype=np.uint64)
        lib().ora_bfv_mult_relin(self.h, lvl, _p(a), _p(b), _p(rlk), klvl, _p(o))
        return o
