"""ctypes binding of liblattisense_amd.so (the C-ABI of include/lattisense_amd.h).

Host-side plumbing only: every compute call goes to the HIP library; there is no Python/NumPy fallback —
if the library or a GPU is missing the calls raise.
"""
import ctypes
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LSA_NATIVE_LIB: an A/B build of the SAME library (lattisense_amd.build.build_variant) for measurements; never a fallback
LIB_PATH = os.environ.get("LSA_NATIVE_LIB") or os.path.join(_HERE, "liblattisense_amd.so")

_lib = None

c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_ll = ctypes.c_longlong
c_vp = ctypes.c_void_p
c_int = ctypes.c_int


class LsaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("lattisense_amd error %d: %s" % (code, msg))
        self.code = code


def _preload_hip_runtime():
    """Load the HIP runtime torch bundles (if torch is installed) before our library, so that a process that also
    imports torch (RCCL, device tensors) ends up with exactly one HIP runtime."""
    spec = importlib.util.find_spec("torch")
    if spec is not None and spec.origin:
        p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(p):
            ctypes.CDLL(p, mode=ctypes.RTLD_GLOBAL)


SIGNATURES = {
    # name: (restype, argtypes)
    "lsa_last_error": (ctypes.c_char_p, []),
    "lsa_version": (ctypes.c_char_p, []),
    "lsa_build_flags": (ctypes.c_char_p, []),
    "lsa_context_create": (c_int, [c_int, c_int, c_u64p, c_int, c_u64p, c_int, ctypes.c_uint64, c_int,
                                   ctypes.POINTER(c_vp)]),
    "lsa_context_destroy": (c_int, [c_vp]),
    "lsa_context_moduli": (c_int, [c_vp, c_u64p, c_int, ctypes.POINTER(c_int)]),
    "lsa_malloc": (c_int, [c_vp, ctypes.POINTER(c_vp), ctypes.c_size_t]),
    "lsa_free": (c_int, [c_vp, c_vp]),
    "lsa_memcpy_h2d": (c_int, [c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "lsa_memcpy_d2h": (c_int, [c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "lsa_memcpy_d2d": (c_int, [c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "lsa_stream_create": (c_int, [c_vp, ctypes.POINTER(c_vp)]),
    "lsa_stream_destroy": (c_int, [c_vp, c_vp]),
    "lsa_stream_synchronize": (c_int, [c_vp, c_vp]),
    "lsa_event_create": (c_int, [c_vp, ctypes.POINTER(c_vp)]),
    "lsa_event_record": (c_int, [c_vp, c_vp, c_vp]),
    "lsa_event_elapsed_ms": (c_int, [c_vp, c_vp, c_vp, ctypes.POINTER(ctypes.c_float)]),
    "lsa_event_destroy": (c_int, [c_vp, c_vp]),
    "lsa_key_upload": (c_int, [c_vp, c_vp, c_int, c_vp, ctypes.POINTER(c_vp)]),
    "lsa_key_adopt_device": (c_int, [c_vp, c_vp, c_int, c_vp, ctypes.POINTER(c_vp)]),
    "lsa_key_destroy": (c_int, [c_vp, c_vp]),
    "lsa_key_bytes": (ctypes.c_size_t, [c_vp, c_int]),
    "lsa_ntt": (c_int, [c_vp, c_vp, c_int, c_ll, c_int, ctypes.POINTER(c_int), c_int, c_int, c_vp]),
    "lsa_poly_addsub": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_ll, c_ll, c_ll, c_vp]),
    "lsa_ckks_mult": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_ll, c_ll, c_ll, c_vp]),
    "lsa_ckks_relin": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_ll, c_ll, c_vp]),
    "lsa_ckks_rescale": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_ll, c_ll, c_vp]),
    "lsa_ckks_rotate": (c_int, [c_vp, c_int, c_vp, ctypes.c_uint64, c_vp, c_vp, c_int, c_ll, c_ll, c_vp]),
    "lsa_drop_level": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_ll, c_ll, c_vp]),
    "lsa_ckks_mult_relin_rescale": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_ll, c_ll, c_ll, c_vp]),
    "lsa_bfv_mult": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_ll, c_ll, c_ll, c_vp]),
    "lsa_bfv_relin": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_ll, c_ll, c_vp]),
    "lsa_bfv_rotate": (c_int, [c_vp, c_int, c_vp, ctypes.c_uint64, c_vp, c_vp, c_int, c_ll, c_ll, c_vp]),
    "lsa_bfv_rescale": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_ll, c_ll, c_vp]),
    "lsa_bfv_mult_relin": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_ll, c_ll, c_ll, c_vp]),
    "lsa_set_tile_batch": (c_int, [c_vp, c_int]),
    "lsa_set_fp64_ntt": (c_int, [c_vp, c_int]),
    "lsa_set_dual_stream": (c_int, [c_vp, c_int]),
    "lsa_set_fuse_tails": (c_int, [c_vp, c_int]),
    "lsa_set_ntt_chunk_mib": (c_int, [c_vp, c_int]),
    "lsa_debug_set_ntt_stamps": (c_int, [c_vp, c_vp]),
    "lsa_ckks_rotate_many": (c_int, [c_vp, c_int, c_vp, c_int, c_u64p, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), c_int,
                                     ctypes.c_longlong, ctypes.c_longlong, c_vp]),
    "lsa_bootstrap_create": (c_int, [c_vp, c_int, c_int, c_int, c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_int,
                                     c_vp, ctypes.POINTER(c_vp)]),
    "lsa_bootstrap_create_ex": (c_int, [c_vp, c_int, c_int, c_int, c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_int,
                                        c_int, c_int, c_vp, ctypes.POINTER(c_vp)]),
    "lsa_bootstrap_evalmod_constants": (c_int, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int),
                                                ctypes.POINTER(ctypes.c_double)]),
    "lsa_bootstrap_destroy": (None, [c_vp]),
    "lsa_bootstrap_info": (c_int, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int),
                                   ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "lsa_bootstrap_galois_elements": (c_int, [c_vp, c_u64p, c_int]),
    "lsa_bootstrap_chebyshev": (c_int, [c_vp, ctypes.POINTER(ctypes.c_double)]),
    "lsa_bootstrap_matrix_info": (c_int, [c_vp, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                          ctypes.POINTER(c_int), c_int]),
    "lsa_bootstrap_plaintext": (c_int, [c_vp, c_int, c_int, c_u64p]),
    "lsa_bootstrap_plaintext_rows": (c_int, [c_vp, c_int, ctypes.POINTER(c_int)]),
    "lsa_bootstrap_plaintext_ext": (c_int, [c_vp, c_int, c_int, c_u64p, ctypes.c_longlong]),
    "lsa_ckks_bootstrap": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, ctypes.c_longlong, ctypes.c_longlong, c_vp, c_int, c_u64p,
                                   ctypes.POINTER(c_vp), c_vp, c_vp, c_vp]),
    "lsa_profile_begin": (c_int, [c_vp, c_int]),
    "lsa_profile_end": (c_int, [c_vp]),
    "lsa_profile_read": (c_int, [c_vp, c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                 ctypes.POINTER(c_ll), ctypes.POINTER(c_ll)]),
    "lsa_profile_read_primary": (c_int, [c_vp, c_int, ctypes.POINTER(ctypes.c_double)]),
    "lsa_probe_copy": (c_int, [c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "lsa_probe_mulhi": (c_int, [c_vp, c_vp, ctypes.c_size_t, c_int, c_vp]),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("liblattisense_amd.so is not built: run `python -m lattisense_amd.build` "
                               "(or __graft_entry__.build()); there is no fallback path")
        _preload_hip_runtime()
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here == the .so does not export what the header declares
            fn.restype = res
            fn.argtypes = args
        flags = L.lsa_build_flags().decode()
        if flags and not os.environ.get("LSA_NATIVE_LIB"):
            # a library at the product's path must be the product: A/B and diagnostic builds (some compute wrong results on
            # purpose) live under variants/ and are only ever selected explicitly through LSA_NATIVE_LIB
            raise RuntimeError("%s was built with switches [%s]: not the product build; rebuild with "
                               "`python -m lattisense_amd.build --force`" % (LIB_PATH, flags))
        _lib = L
    return _lib


def build_flags():
    """The LSA_* switches of the loaded library ("" = product build)."""
    return lib().lsa_build_flags().decode()


def check(rc):
    if rc != 0:
        raise LsaError(rc, lib().lsa_last_error().decode())
