"""ctypes binding of libcovgram.so — one prototype per symbol of include/covgram.h.

The library is the product; this module only loads it and turns status codes into exceptions.
There is deliberately NO fallback: if the shared object is missing, or no gfx950 device is
visible, every compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# COVGRAM_LIB: another build of the SAME library (A/B of two builds in tools/); there is no other implementation to point it at
LIB_PATH = os.environ.get("COVGRAM_LIB") or os.path.normpath(os.path.join(_HERE, "..", "lib", "libcovgram.so"))

# enums (include/covgram.h)
EQ, EXP, RQ, GAMMAEXP, CAUCHY, IMQ, MATERNP, DOT, EXPDOT, MATERN, ASINDOT = range(11)
CONSTANT, COMPOSITE = 100, 101
COMPOSITE_MAX_TERMS, COMPOSITE_MAX_FACTORS = 8, 8
ISOTROPIC, DOTPRODUCT = 1, 2
F32, F64 = 0, 1
HOST, DEVICE = 0, 1
OK, EINVAL, EUNSUPPORTED, EHIP, ENODEVICE, ENOMEM = 0, -1, -2, -3, -4, -5
MATERNP_MAX_P = 8
COMM_ID_BYTES = 128
ABI_VERSION = 113   # COVGRAM_VERSION of the include/covgram.h these prototypes mirror


class covgram_kernel(C.Structure):
    _fields_ = [
        ("family", C.c_int32),
        ("trait", C.c_int32),
        ("p", C.c_int32),
        ("power", C.c_int32),
        ("param", C.c_double),
        ("lengthscale", C.c_double),
        ("scale", C.c_double),
    ]


class covgram_kernel_composite(C.Structure):
    """Sum of products of same-trait profiles; pass `kref(c)` (== &c.head) wherever a covgram_kernel* is expected."""
    _fields_ = [
        ("head", covgram_kernel),
        ("nterms", C.c_int32),
        ("nfactors", C.c_int32 * COMPOSITE_MAX_TERMS),
        ("factors", covgram_kernel * COMPOSITE_MAX_FACTORS),
    ]


def kref(spec):
    """`const covgram_kernel*` for a simple spec or a composite (whose head is its first member)."""
    return C.cast(C.pointer(spec), C.POINTER(covgram_kernel))


class CovgramError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libcovgram status {status}: {message}")
        self.status = status


class DimensionMismatch(CovgramError, ValueError):
    """Julia's DimensionMismatch (src/util.jl:41, :9) — raised for COVGRAM_EINVAL."""


class UnsupportedKernel(CovgramError, NotImplementedError):
    """Kernel / dimension outside the compiled device set (COVGRAM_EUNSUPPORTED)."""


class NoDevice(CovgramError):
    """No gfx950 device: the product path fails loudly (COVGRAM_ENODEVICE)."""


_P = C.c_void_p
_I64 = C.c_int64
_I32 = C.c_int32
_D = C.c_double
_KP = C.POINTER(covgram_kernel)

# name -> (restype, argtypes): exactly the declarations of include/covgram.h
PROTOTYPES = {
    "covgram_version": (C.c_int, []),
    "covgram_sizeof_kernel": (C.c_int, []),
    "covgram_sizeof_composite": (C.c_int, []),
    "covgram_last_error": (C.c_char_p, []),
    "covgram_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "covgram_ctx_create": (C.c_int, [C.POINTER(_P), C.c_int, _P]),
    "covgram_ctx_destroy": (C.c_int, [_P]),
    "covgram_ctx_set_stream": (C.c_int, [_P, _P]),
    "covgram_ctx_get_stream": (C.c_int, [_P, C.POINTER(_P)]),
    "covgram_ctx_set_option": (C.c_int, [_P, C.c_char_p, _I64]),
    "covgram_ctx_get_info": (C.c_int, [_P, C.c_char_p, C.POINTER(_I64)]),
    "covgram_sync": (C.c_int, [_P]),
    "covgram_ctx_kernel_time": (C.c_int, [_P, C.POINTER(_D), C.POINTER(_I64), _I32]),
    "covgram_points_create": (C.c_int, [_P, C.POINTER(_P), _P, _I64, _I32, _I32, _I32]),
    "covgram_points_slice": (C.c_int, [_P, _I64, _I64, C.POINTER(_P)]),
    "covgram_points_destroy": (C.c_int, [_P]),
    "covgram_points_info": (C.c_int, [_P, C.POINTER(_I64), C.POINTER(_I32), C.POINTER(_I32)]),
    "covgram_mvm": (C.c_int, [_P, _KP, _P, _P, _P, _I64, _P, _I64, _I32, _D, _D, _I32]),
    "covgram_matrix": (C.c_int, [_P, _KP, _P, _P, _P, _I64, _I32]),
    "covgram_grad_mvm": (C.c_int, [_P, _KP, _P, _P, _P, _I64, _P, _I64, _I32, _D, _D, _I32]),
    "covgram_valgrad_mvm": (C.c_int, [_P, _KP, _P, _P, _P, _I64, _P, _I64, _I32, _D, _D, _I32]),
    "covgram_mvm_sym_supported": (C.c_int, [_P, _KP, _P, _I32, C.POINTER(C.c_int32)]),
    "covgram_mvm_sym_partial": (C.c_int, [_P, _KP, _P, _P, _P, _I32, _I32]),
    "covgram_comm_unique_id": (C.c_int, [_P, _I64]),
    "covgram_comm_create": (C.c_int, [_P, _P, _I32, _I32]),
    "covgram_comm_destroy": (C.c_int, [_P]),
    "covgram_comm_info": (C.c_int, [_P, C.POINTER(_I32), C.POINTER(_I32)]),
    "covgram_comm_all_gather": (C.c_int, [_P, _P, _P, _I64, _I32]),
    "covgram_comm_all_reduce_sum": (C.c_int, [_P, _P, _I64, _I32]),
    "covgram_mvm_sharded": (C.c_int, [_P, _KP, _P, _P, _P, _P, _D, _D]),
    "covgram_mvm_sym_allreduce": (C.c_int, [_P, _KP, _P, _P, _P, _D, _D]),
    "covgram_toeplitz_create": (C.c_int, [_P, C.POINTER(_P), _P, _P, _I64, _I64, _I32, _I32, _I32]),
    "covgram_toeplitz_mvm": (C.c_int, [_P, _P, _P, _D, _D, _I32]),
    "covgram_toeplitz_destroy": (C.c_int, [_P]),
    "covgram_cg_step": (C.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _P]),
    "covgram_cg_step_shifted": (C.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _P, _P]),
    "covgram_toeplitz_durbin": (C.c_int, [_P, _P, _I64, _P, _I32, _I32]),
    "covgram_toeplitz_levinson": (C.c_int, [_P, _P, _P, _I64, _P, _I32, _I32]),
    "covgram_toeplitz_trench": (C.c_int, [_P, _P, _I64, _P, _I64, _I32, _I32]),
    "covgram_kron_mvm": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64), _I32, _I32,
                                   _P, _I64, _P, _I64, _I32, _D, _D, _I32]),
    "covgram_lowrank_mvm": (C.c_int, [_P, _P, _I64, _P, _I64, _I64, _I64, _I64, _I32, _P, _I64, _P, _I64, _I32, _D, _D, _I32]),
    "covgram_debug_kernel_params": (C.c_int, [_KP, _I32, _I32, C.POINTER(_D)]),
}

_lib = None


def lib():
    """Load libcovgram.so (once).  Missing library = hard error with the build recipe."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `make -C covariancefunctions.jl_amd -j8` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(l, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        got = l.covgram_version()
        if got != ABI_VERSION:   # e.g. COVGRAM_LIB pointing at another build: its argument lists differ (covgram.h history)
            raise RuntimeError(f"{LIB_PATH} is ABI version {got}, this binding is {ABI_VERSION}: rebuild the library")
        _lib = l
    return _lib


def check(status: int):
    if status == OK:
        return
    msg = lib().covgram_last_error().decode("utf-8", "replace")
    if status == EINVAL:
        raise DimensionMismatch(status, msg)
    if status == EUNSUPPORTED:
        raise UnsupportedKernel(status, msg)
    if status == ENODEVICE:
        raise NoDevice(status, msg)
    raise CovgramError(status, msg)
