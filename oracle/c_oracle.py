"""ctypes access to the C oracle (oracle/covgram_oracle.c).  TEST INFRASTRUCTURE ONLY — see that file's header."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class oracle_kernel(C.Structure):
    _fields_ = [("family", C.c_int32), ("trait", C.c_int32), ("p", C.c_int32), ("power", C.c_int32),
                ("param", C.c_double), ("lengthscale", C.c_double), ("scale", C.c_double)]


def build(march: str = "x86-64-v3", out: str | None = None) -> str:
    out = out or os.path.join(HERE, "build")
    subprocess.run(["make", "-C", HERE, f"MARCH={march}", f"OUT={out}"], check=True, capture_output=True)
    return out


def _load(name: str, out: str | None = None):
    path = os.path.join(out or os.path.join(HERE, "build"), name)
    if not os.path.exists(path):
        build(out=out)
    return C.CDLL(path)


def to_c(k) -> oracle_kernel:
    """covgram_oracle.Kernel -> C struct."""
    return oracle_kernel(k.family, k.trait, k.p, k.power, float(k.param), float(k.lengthscale), float(k.scale))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def mvm(k, X, Y, a, y=None, alpha=1.0, beta=0.0, lib=None):
    lib = lib or _load("libcovgram_oracle.so")
    X = np.ascontiguousarray(X); Y = np.ascontiguousarray(Y)
    if X.ndim == 1: X = X[:, None]
    if Y.ndim == 1: Y = Y[:, None]
    dt = X.dtype
    fn = lib.oracle_mvm_f64 if dt == np.float64 else lib.oracle_mvm_f32
    a = np.asarray(a, dtype=dt)
    A = np.asfortranarray(a.reshape(a.shape[0], -1))
    n, m, d, nrhs = X.shape[0], Y.shape[0], X.shape[1], A.shape[1]
    out = np.zeros((n, nrhs), dtype=dt, order="F") if y is None else np.asfortranarray(np.asarray(y, dtype=dt).reshape(n, -1).copy())
    ck = to_c(k)
    fn.restype = C.c_int
    rc = fn(C.byref(ck), _p(X), C.c_int64(n), _p(Y), C.c_int64(m), C.c_int32(d), _p(A), C.c_int64(m), _p(out), C.c_int64(n),
            C.c_int32(nrhs), C.c_double(alpha), C.c_double(beta))
    assert rc == 0, rc
    return out[:, 0] if a.ndim == 1 else np.ascontiguousarray(out)


def grad_mvm(k, X, Y, a, y=None, alpha=1.0, beta=0.0, lib=None):
    lib = lib or _load("libcovgram_oracle.so")
    X = np.ascontiguousarray(X, dtype=np.float64); Y = np.ascontiguousarray(Y, dtype=np.float64)
    n, d = X.shape; m = Y.shape[0]
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = np.zeros(n * d) if y is None else np.array(y, dtype=np.float64).copy()
    ck = to_c(k)
    lib.oracle_grad_mvm_f64.restype = C.c_int
    rc = lib.oracle_grad_mvm_f64(C.byref(ck), _p(X), C.c_int64(n), _p(Y), C.c_int64(m), C.c_int32(d), _p(a), _p(out),
                                 C.c_double(alpha), C.c_double(beta))
    assert rc == 0, rc
    return out


def toeplitz_mvm(vc, vr, a, y=None, alpha=1.0, beta=0.0, lib=None):
    lib = lib or _load("libcovgram_oracle.so")
    vc = np.ascontiguousarray(vc, dtype=np.float64)
    n = vc.shape[0]
    vr_ = None if vr is None else np.ascontiguousarray(vr, dtype=np.float64)
    m = n if vr_ is None else vr_.shape[0]
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = np.zeros(n) if y is None else np.array(y, dtype=np.float64).copy()
    lib.oracle_toeplitz_mvm_f64.restype = C.c_int
    rc = lib.oracle_toeplitz_mvm_f64(_p(vc), None if vr_ is None else _p(vr_), C.c_int64(n), C.c_int64(m), _p(a), _p(out),
                                     C.c_double(alpha), C.c_double(beta))
    assert rc == 0, rc
    return out


def eq_rows(X, Y, a, row0, row1, lib):
    """Timed CPU-baseline kernel: rows [row0,row1) of the EQ MVM over all m columns."""
    X = np.ascontiguousarray(X); Y = np.ascontiguousarray(Y); a = np.ascontiguousarray(a)
    fn = lib.oracle_mvm_eq_f32 if X.dtype == np.float32 else lib.oracle_mvm_eq_f64
    out = np.zeros(row1 - row0, dtype=X.dtype)
    fn.restype = C.c_int
    fn(_p(X), C.c_int64(X.shape[0]), _p(Y), C.c_int64(Y.shape[0]), C.c_int32(X.shape[1]), _p(a), _p(out), C.c_int64(row0), C.c_int64(row1))
    return out


def num_threads(lib=None):
    lib = lib or _load("libcovgram_oracle.so")
    return lib.oracle_num_threads()
