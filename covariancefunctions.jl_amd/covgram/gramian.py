"""`gramian` / `Gramian` / `mul!` — the host-side mirror of src/gramian.jl for the device engine.

Layout conventions (the reference's):
  * a point set is a tensor of shape (n, d) (C-contiguous) — the same memory as Julia's d×n matrix whose
    columns are the points (src/gramian.jl:2,154-155); a 1-D tensor is n scalar points;
  * block vectors of multi-output Gramians are flat and point-major;
  * matrix right-hand sides have shape (m, p).
torch is used for device memory and streams only; every product below runs in libcovgram.so.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _ffi
from . import kernels as K

# ----------------------------------------------------------------------------------------------
# context handling: one covgram_ctx per device, bound to torch's current stream at call time
# ----------------------------------------------------------------------------------------------
_CTX = {}


class _Ctx:
    def __init__(self, device: torch.device):
        self.device = device
        self.handle = _ffi._P()
        self._stream = torch.cuda.current_stream(device).cuda_stream
        _ffi.check(_ffi.lib().covgram_ctx_create(C.byref(self.handle), device.index, _ffi._P(self._stream)))

    def bind_stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        if s != self._stream:
            _ffi.check(_ffi.lib().covgram_ctx_set_stream(self.handle, _ffi._P(s)))
            self._stream = s
        return self.handle

    def set_option(self, key: str, value: int):
        _ffi.check(_ffi.lib().covgram_ctx_set_option(self.handle, key.encode(), int(value)))


def kernel_time(device=None, reset=True):
    """(total_ms, launches) of the HIP-event-bracketed dominant kernels since the last reset
    (enable with set_option("time_kernels", 1))."""
    ctx = get_ctx(device)
    ms, cnt = C.c_double(0), C.c_int64(0)
    _ffi.check(_ffi.lib().covgram_ctx_kernel_time(ctx.handle, C.byref(ms), C.byref(cnt), 1 if reset else 0))
    return ms.value, cnt.value


def get_ctx(device=None) -> _Ctx:
    """The library context for `device` (default: torch's current CUDA/HIP device).  Raises if the
    shared library is missing or no gfx950 device is visible — there is no CPU fallback."""
    lib = _ffi.lib()
    if device is None or (isinstance(device, torch.device) and device.type != "cuda"):
        n = C.c_int(0)
        lib.covgram_device_count(C.byref(n))
        if n.value <= 0 or not torch.cuda.is_available():
            raise _ffi.NoDevice(_ffi.ENODEVICE, "no HIP device visible — covgram has no CPU fallback")
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if device.index not in _CTX:
        _CTX[device.index] = _Ctx(device)
    return _CTX[device.index]


def set_option(key: str, value: int, device=None):
    get_ctx(device).set_option(key, value)


def get_info(key: str, device=None) -> int:
    """covgram_ctx_get_info: e.g. "last_dense_path" (1 lane-per-row direct differences, 2 matrix-core EQ, 3 wide rows)."""
    v = C.c_int64(0)
    _ffi.check(_ffi.lib().covgram_ctx_get_info(get_ctx(device).handle, key.encode(), C.byref(v)))
    return v.value


def _dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return _ffi.F32
    if dt == torch.float64:
        return _ffi.F64
    raise TypeError(f"covgram supports float32/float64 points, got {dt}")


# ----------------------------------------------------------------------------------------------
# inputs: ranges, grids, point tensors
# ----------------------------------------------------------------------------------------------
class StepRangeLen:
    """Julia's `range(start, stop, length)` / StepRangeLen: the trigger for Toeplitz structure
    (src/gramian.jl:167-189)."""

    def __init__(self, start, step, length, dtype=torch.float64):
        self.start, self.step, self.length, self.dtype = float(start), float(step), int(length), dtype

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self.tensor("cpu")[i]
        if i < 0:
            i += self.length
        return self.start + self.step * i

    def __add__(self, c):   # x .+ c keeps the step (test/gramian.jl:169)
        return StepRangeLen(self.start + float(c), self.step, self.length, self.dtype)

    def tensor(self, device):
        idx = torch.arange(self.length, dtype=torch.float64, device=device)
        return (self.start + self.step * idx).to(self.dtype)


def srange(start, stop, length, dtype=torch.float64) -> StepRangeLen:
    """range(start, stop, length)."""
    step = (float(stop) - float(start)) / (length - 1) if length > 1 else 0.0
    return StepRangeLen(start, step, length, dtype)


class LazyGrid:
    """Lazy Cartesian grid (src/lazy_grid.jl:3-38); default enumeration: FIRST axis varies fastest."""

    def __init__(self, *args):
        if len(args) == 2 and isinstance(args[1], int) and not isinstance(args[0], (int, float)):
            args = tuple([args[0]] * args[1])                 # LazyGrid(x, d) = Fill(x, d)  (lazy_grid.jl:11)
        self.args = tuple(args)

    def __len__(self):
        return int(np.prod([len(a) for a in self.args]))

    def ndims(self):
        return len(self.args)

    def points(self, device, dtype=torch.float64, rowmajor=False):
        axes = [(a.tensor(device) if isinstance(a, StepRangeLen) else torch.as_tensor(a, device=device)).to(dtype).reshape(-1)
                for a in self.args]
        mesh = torch.meshgrid(*axes, indexing="ij")
        if rowmajor:       # lazy_grid.jl:40-58
            cols = [g.reshape(-1) for g in mesh]
        else:              # lazy_grid.jl:20-38: first axis fastest == Fortran-order flattening
            cols = [g.permute(*reversed(range(g.dim()))).reshape(-1) for g in mesh]
        return torch.stack(cols, dim=1).contiguous()


def _as_points(x, device=None, dtype=None) -> torch.Tensor:
    if isinstance(x, StepRangeLen):
        dev = device if device is not None else get_ctx().device
        t = x.tensor(dev)
    elif isinstance(x, LazyGrid):
        dev = device if device is not None else get_ctx().device
        t = x.points(dev)
    elif isinstance(x, torch.Tensor):
        t = x
    else:
        t = torch.as_tensor(np.asarray(x))
    if t.dtype not in (torch.float32, torch.float64):
        t = t.to(torch.float64)
    if dtype is not None:
        t = t.to(dtype)
    if t.dim() == 1:
        t = t.reshape(-1, 1)
    if t.dim() != 2:
        raise _ffi.DimensionMismatch(_ffi.EINVAL, f"points must be (n,) or (n, d), got shape {tuple(t.shape)}")
    if t.device.type != "cuda":
        t = t.to(device if device is not None else get_ctx().device)
    return t.contiguous()


class _Points:
    """Device-resident point set + its covgram_points handle (borrowing the tensor's memory).

    The C ABI's contract for borrowed memory is "unchanged while the handle lives" (the library caches the point set's max norm
    and, on the matrix-core path, its packed fragments).  A lazy Gramian in the reference, however, follows in-place changes of
    its points, so this wrapper watches torch's in-place version counter and re-creates the handle when the tensor was written
    to since the handle was made."""

    def __init__(self, t: torch.Tensor):
        self.t = t
        self.ctx = get_ctx(t.device)
        self._h = _ffi._P()
        self._create()

    def _create(self):
        t = self.t
        _ffi.check(_ffi.lib().covgram_points_create(self.ctx.handle, C.byref(self._h), _ffi._P(t.data_ptr()), t.shape[0],
                                                    t.shape[1], _dtype_code(t.dtype), _ffi.DEVICE))
        self._ver = t._version

    @property
    def handle(self):
        if self.t._version != self._ver:                      # points were modified in place: norms / fragments are stale
            _ffi.lib().covgram_points_destroy(self._h)
            self._h = _ffi._P()
            self._create()
        return self._h

    def __del__(self):
        try:
            if self._h:
                _ffi.lib().covgram_points_destroy(self._h)
                self._h = None
        except Exception:
            pass


def _vec_arg(a: torch.Tensor, n: int, dtype, device, what: str) -> torch.Tensor:
    if not isinstance(a, torch.Tensor):
        a = torch.as_tensor(np.asarray(a))
    if a.shape[0] != n:
        raise _ffi.DimensionMismatch(_ffi.EINVAL, f"DimensionMismatch: {what} has length {a.shape[0]}, expected {n}")
    return a.to(device=device, dtype=dtype)


# ----------------------------------------------------------------------------------------------
# lazy operators
# ----------------------------------------------------------------------------------------------
class LazyOperator:
    """Common surface of every lazily represented matrix: shape, `@`, `mul_` (= mul!), `to_dense()`."""

    shape = (0, 0)
    dtype = torch.float64
    device = None

    def size(self, i=None):
        return self.shape if i is None else self.shape[i]

    def mul_(self, y, a, alpha=1.0, beta=0.0):
        raise NotImplementedError

    def __matmul__(self, a):
        a = _vec_arg(a, self.shape[1], self.dtype, self.device, "a")
        y = torch.empty((self.shape[0],) + tuple(a.shape[1:]), dtype=self.dtype, device=self.device)
        return self.mul_(y, a, 1.0, 0.0)                      # src/gramian.jl:66-75: zeros + mul!

    def to_dense(self):
        eye = torch.eye(self.shape[1], dtype=self.dtype, device=self.device)
        return self @ eye

    def __add__(self, D):
        return LazyMatrixSum(self, D)                         # src/gramian.jl:55-60

    __radd__ = __add__


def mul_(y, A: LazyOperator, a, alpha=1.0, beta=0.0):
    """LinearAlgebra.mul!(y, A, a, α, β): y ← α A a + β y, returns y."""
    return A.mul_(y, a, alpha, beta)


class Gramian(LazyOperator):
    """Lazy kernel matrix holding (k, x, y) (src/gramian.jl:10-21); O(1) construction, O(n m d) `mul!`."""

    def __init__(self, k, x, y=None):
        self.k = k
        self.x = _as_points(x)
        if y is None or y is x:
            self.y = self.x
        else:
            self.y = _as_points(y, device=self.x.device, dtype=self.x.dtype)
        if self.x.shape[1] != self.y.shape[1]:
            raise _ffi.DimensionMismatch(_ffi.EINVAL, f"inputs have to have the same length: {self.x.shape[1]}, {self.y.shape[1]}")
        self.dtype = self.x.dtype                              # gramian_eltype: T follows the data (src/gramian.jl:30-33)
        self.device = self.x.device
        self.shape = (self.x.shape[0], self.y.shape[0])
        self._px = _Points(self.x)
        self._py = self._px if self.y is self.x else _Points(self.y)

    # -- structure ------------------------------------------------------------------------------
    @property
    def T(self):
        return Gramian(self.k, self.y, self.x)                 # src/gramian.jl:116-117

    adjoint = T

    def issymmetric(self):
        return self.x is self.y or (self.x.shape == self.y.shape and bool(torch.equal(self.x, self.y)))   # :132

    def isposdef(self):
        return isinstance(self.k, (K.MercerKernel, K.MultiKernel)) and self.issymmetric()               # :135-137

    def _spec(self):
        # lowered once per Gramian: like the reference's Gramian{T, K, ...}, which holds an immutable kernel by value
        sp = getattr(self, "_spec_cached", None)
        if sp is None:
            sp = self._spec_cached = K.require_device_spec(self.k)
        return sp

    # -- products ---------------------------------------------------------------------------------
    def mul_(self, y, a, alpha=1.0, beta=0.0):
        """src/gramian.jl:78-99.  β == 0 ⇒ y's previous contents (NaN included) are ignored."""
        n, m = self.shape
        a = _vec_arg(a, m, self.dtype, self.device, "a")
        if y.shape[0] != n or y.dtype != self.dtype or tuple(y.shape[1:]) != tuple(a.shape[1:]):
            raise _ffi.DimensionMismatch(_ffi.EINVAL, f"DimensionMismatch: y has shape {tuple(y.shape)}, expected ({n}, ...) of {self.dtype}")
        spec = self._spec()
        ctx = self._px.ctx.bind_stream()
        lib = _ffi.lib()
        if a.dim() == 1:
            a_c = a.contiguous()
            y_c = y if y.is_contiguous() else y.contiguous()
            _ffi.check(lib.covgram_mvm(ctx, _ffi.kref(spec), self._px.handle, self._py.handle, _ffi._P(a_c.data_ptr()), max(m, 1),
                                       _ffi._P(y_c.data_ptr()), max(n, 1), 1, float(alpha), float(beta), _ffi.DEVICE))
            if y_c is not y:
                y.copy_(y_c)
            return y
        p = a.shape[1]
        a_cm = a.t().contiguous()                              # (p, m) row-major == m×p column-major
        direct = y.t().is_contiguous()
        y_cm = y.t() if direct else y.t().contiguous()
        _ffi.check(lib.covgram_mvm(ctx, _ffi.kref(spec), self._px.handle, self._py.handle, _ffi._P(a_cm.data_ptr()), max(m, 1),
                                   _ffi._P(y_cm.data_ptr()), max(n, 1), p, float(alpha), float(beta), _ffi.DEVICE))
        if not direct:
            y.copy_(y_cm.t())
        return y

    # -- multi-GPU symmetric form (include/covgram.h: covgram_mvm_sym_partial) ---------------------
    def sym_partial_supported(self, world: int = 1) -> bool:
        """True when rank r of `world` can take its cyclic panels / row blocks of the upper triangle (one point set on both sides: the
        symmetric matrix-core kernels in fp32, the direct-difference symmetric kernels in fp64 and for what the matrix cores refuse);
        depends on the kernel, the points and the world size only, so every rank answers alike — and sym_partial_ raises
        UnsupportedKernel exactly when this says False."""
        if self._px is not self._py and not (self._px.t.data_ptr() == self._py.t.data_ptr() and self.shape[0] == self.shape[1]):
            return False
        try:
            spec = self._spec()
        except Exception:
            return False
        ok = C.c_int32(0)
        _ffi.check(_ffi.lib().covgram_mvm_sym_supported(self._px.ctx.bind_stream(), _ffi.kref(spec), self._px.handle, int(world), C.byref(ok)))
        return bool(ok.value)

    def sym_partial_(self, y, a, rank: int, world: int):
        """y <- the part of G a that comes from the upper-triangle tiles of the panels p ≡ rank (mod world) and their mirror
        images; the partials of all ranks sum to G a.  Vectors only."""
        n, m = self.shape
        a = _vec_arg(a, m, self.dtype, self.device, "a")
        if a.dim() != 1 or y.shape != (n,) or y.dtype != self.dtype or not y.is_contiguous():
            raise _ffi.DimensionMismatch(_ffi.EINVAL, "sym_partial_: contiguous vectors of length n expected")
        a_c = a.contiguous()
        _ffi.check(_ffi.lib().covgram_mvm_sym_partial(self._px.ctx.bind_stream(), _ffi.kref(self._spec()), self._px.handle,
                                                      _ffi._P(a_c.data_ptr()), _ffi._P(y.data_ptr()), int(rank), int(world)))
        return y

    def to_dense(self):
        """Matrix(G) (src/gramian.jl:102-114) on the device."""
        n, m = self.shape
        buf = torch.empty((m, n), dtype=self.dtype, device=self.device)   # column-major n×m
        if n * m:
            spec = self._spec()
            ctx = self._px.ctx.bind_stream()
            _ffi.check(_ffi.lib().covgram_matrix(ctx, _ffi.kref(spec), self._px.handle, self._py.handle, _ffi._P(buf.data_ptr()), n, _ffi.DEVICE))
        return buf.t()

    def __getitem__(self, ij):
        """G[i, j] = k(x[i], y[j]) and sub-block indexing (src/gramian.jl:37-52), evaluated on the device."""
        i, j = ij
        xi = self.x[i] if not isinstance(i, int) else self.x[i:i + 1]
        yj = self.y[j] if not isinstance(j, int) else self.y[j:j + 1]
        sub = Gramian(self.k, xi.reshape(-1, self.x.shape[1]), yj.reshape(-1, self.y.shape[1])).to_dense()
        if isinstance(i, int) and isinstance(j, int):
            return sub[0, 0]
        if isinstance(i, int):
            return sub[0]
        if isinstance(j, int):
            return sub[:, 0]
        return sub


class BlockGramian(LazyOperator):
    """Gramian of a GradientKernel: the lazy (n d)×(m d) BlockFactorization of src/gramian.jl:120-123 whose
    `mul!` is blockmul! (src/gramian.jl:241-257) with the O(d) block product of src/gradient.jl:86-115."""

    def __init__(self, g, x, y=None):
        self.g = g
        self.value = isinstance(g, K.ValueGradientKernel)      # blocks of d+1: src/gradient.jl:400-474
        self.inner = Gramian(g.k, x, y)
        n, m = self.inner.shape
        d = self.inner.x.shape[1]
        self.d = d
        self.block = d + 1 if self.value else d
        self.shape = (n * self.block, m * self.block)
        self.dtype, self.device = self.inner.dtype, self.inner.device

    def issymmetric(self):
        return self.inner.issymmetric()

    def mul_(self, y, a, alpha=1.0, beta=0.0):
        spec = K.require_device_spec(self.g.k)   # GenericInput gradient kernels have no device path
        a = _vec_arg(a, self.shape[1], self.dtype, self.device, "a")
        if y.shape[0] != self.shape[0] or y.dtype != self.dtype or y.dim() != a.dim() or (a.dim() == 2 and y.shape[1] != a.shape[1]):
            raise _ffi.DimensionMismatch(_ffi.EINVAL, f"DimensionMismatch: y has shape {tuple(y.shape)}")
        # matrix right-hand sides (block solvers; src/gramian.jl:241-257 takes vectors of matrices): ONE library call, the columns
        # as column-major storage (a torch matrix is row-major: its transpose's storage); the library shares the pair evaluations
        # between two columns at a time where its kernel holds two accumulators
        nrhs = 1 if a.dim() == 1 else a.shape[1]
        if a.dim() == 1:
            a_c = a.contiguous()
            y_c = y if y.is_contiguous() else y.contiguous()
        else:
            a_c = a.t().contiguous()
            y_c = y.t().contiguous() if beta != 0.0 else torch.empty((nrhs, self.shape[0]), dtype=self.dtype, device=self.device)
        ctx = self.inner._px.ctx.bind_stream()
        fn = _ffi.lib().covgram_valgrad_mvm if self.value else _ffi.lib().covgram_grad_mvm
        _ffi.check(fn(ctx, _ffi.kref(spec), self.inner._px.handle, self.inner._py.handle, _ffi._P(a_c.data_ptr()), self.shape[1],
                      _ffi._P(y_c.data_ptr()), self.shape[0], nrhs, float(alpha), float(beta), _ffi.DEVICE))
        if a.dim() == 2:
            y.copy_(y_c.t())
        elif y_c is not y:
            y.copy_(y_c)
        return y


class _ToeplitzBase(LazyOperator):
    def __init__(self, vc: torch.Tensor, vr: Optional[torch.Tensor], circulant: bool):
        self.vc = vc.contiguous()
        self.vr = None if vr is None else vr.contiguous()
        self.circulant = circulant
        n = vc.shape[0]
        m = n if vr is None else vr.shape[0]
        self.shape = (n, m)
        self.dtype, self.device = vc.dtype, vc.device
        self._ctx = get_ctx(vc.device)
        self.handle = _ffi._P()
        _ffi.check(_ffi.lib().covgram_toeplitz_create(self._ctx.bind_stream(), C.byref(self.handle), _ffi._P(self.vc.data_ptr()),
                                                      _ffi._P(self.vr.data_ptr()) if self.vr is not None else None, n, m,
                                                      _dtype_code(self.dtype), _ffi.DEVICE, 1 if circulant else 0))

    def __del__(self):
        try:
            if self.handle:
                _ffi.lib().covgram_toeplitz_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def mul_(self, y, a, alpha=1.0, beta=0.0):
        a = _vec_arg(a, self.shape[1], self.dtype, self.device, "a")
        if a.dim() != 1:
            for c in range(a.shape[1]):
                yc = y[:, c].contiguous()
                self.mul_(yc, a[:, c].contiguous(), alpha, beta)
                y[:, c] = yc
            return y
        if y.shape[0] != self.shape[0] or y.dtype != self.dtype:
            raise _ffi.DimensionMismatch(_ffi.EINVAL, f"DimensionMismatch: y has shape {tuple(y.shape)}")
        a_c = a.contiguous()
        y_c = y if y.is_contiguous() else y.contiguous()
        self._ctx.bind_stream()
        _ffi.check(_ffi.lib().covgram_toeplitz_mvm(self.handle, _ffi._P(a_c.data_ptr()), _ffi._P(y_c.data_ptr()), float(alpha), float(beta), _ffi.DEVICE))
        if y_c is not y:
            y.copy_(y_c)
        return y

    def to_dense(self):
        n, m = self.shape
        i = torch.arange(n, device=self.device)[:, None]
        j = torch.arange(m, device=self.device)[None, :]
        if self.circulant:
            return self.vc[(i - j) % n]
        vr = self.vc if self.vr is None else self.vr
        return torch.where(i >= j, self.vc[(i - j).clamp(min=0)], vr[(j - i).clamp(min=0)])


class SymmetricToeplitz(_ToeplitzBase):
    def __init__(self, vc):
        super().__init__(vc, None, False)


class Toeplitz(_ToeplitzBase):
    def __init__(self, vc, vr):
        super().__init__(vc, vr, False)


class Circulant(_ToeplitzBase):
    def __init__(self, vc):
        super().__init__(vc, None, True)


class KroneckerProduct(LazyOperator):
    """kronecker(F_1, ..., F_q) (KroneckerProducts 1.1.1): standard order, F_1 = slowest index.  Factors are
    lazy Gramians or dense matrices; lazy factors are instantiated once on the device (they are the small
    per-axis matrices, cf. README.md:205-210) and the MVM runs as mode products in libcovgram."""

    def __init__(self, *factors):
        self.factors = list(factors)
        rows = [f.shape[0] for f in self.factors]
        cols = [f.shape[1] for f in self.factors]
        self.shape = (int(np.prod(rows)), int(np.prod(cols)))
        f0 = self.factors[0]
        self.dtype = f0.dtype
        self.device = f0.device
        self._dense = None

    def _dense_factors(self):
        if self._dense is None:
            out = []
            for f in self.factors:
                D = f.to_dense() if isinstance(f, LazyOperator) else f
                out.append(D.to(self.dtype).t().contiguous())    # (cols, rows) row-major == column-major rows×cols
            self._dense = out
        return self._dense

    def mul_(self, y, a, alpha=1.0, beta=0.0):
        a = _vec_arg(a, self.shape[1], self.dtype, self.device, "a")
        if y.shape[0] != self.shape[0] or y.dtype != self.dtype or y.dim() != a.dim() or (a.dim() == 2 and y.shape[1] != a.shape[1]):
            raise _ffi.DimensionMismatch(_ffi.EINVAL, f"DimensionMismatch: y has shape {tuple(y.shape)}")
        fs = self._dense_factors()
        q = len(fs)
        ptrs = (_ffi._P * q)(*[_ffi._P(f.data_ptr()) for f in fs])
        rows = (C.c_int64 * q)(*[f.shape[1] for f in fs])
        cols = (C.c_int64 * q)(*[f.shape[0] for f in fs])
        lds = (C.c_int64 * q)(*[f.shape[1] for f in fs])
        # right-hand sides: column-major n x p (a torch matrix is row-major: its transpose's storage)
        nrhs = 1 if a.dim() == 1 else a.shape[1]
        a_c = a.contiguous() if a.dim() == 1 else a.t().contiguous()
        if a.dim() == 1:
            y_c = y if y.is_contiguous() else y.contiguous()
        else:
            y_c = y.t().contiguous() if beta != 0.0 else torch.empty((nrhs, self.shape[0]), dtype=self.dtype, device=self.device)
        ctx = get_ctx(self.device).bind_stream()
        _ffi.check(_ffi.lib().covgram_kron_mvm(ctx, ptrs, rows, cols, lds, q, _dtype_code(self.dtype), _ffi._P(a_c.data_ptr()), self.shape[1],
                                               _ffi._P(y_c.data_ptr()), self.shape[0], nrhs, float(alpha), float(beta), _ffi.DEVICE))
        if a.dim() == 2:
            y.copy_(y_c.t())
        elif y_c is not y:
            y.copy_(y_c)
        return y

    def to_dense(self):
        out = None
        for f in self.factors:
            D = f.to_dense() if isinstance(f, LazyOperator) else f
            D = D.contiguous()
            out = D if out is None else torch.kron(out, D)
        return out


def kronecker(*factors):
    if len(factors) == 1 and isinstance(factors[0], SeparableGramian):
        return factors[0].kronecker()
    return KroneckerProduct(*factors)


class SeparableGramian(LazyOperator):
    """Gramian of a SeparableKernel: mul! = kronecker(G) = gramian(k.k, x, y) ⊗ B (src/separable.jl:33-42)."""

    def __init__(self, s: K.SeparableKernel, x, y=None):
        self.s = s
        self.inner = Gramian(s.k, x, y)
        self.dtype, self.device = self.inner.dtype, self.inner.device
        self.B = torch.as_tensor(s.B, dtype=self.dtype, device=self.device)
        n, m = self.inner.shape
        self.shape = (n * self.B.shape[0], m * self.B.shape[1])
        self._kron = None

    def kronecker(self):
        if self._kron is None:
            self._kron = KroneckerProduct(self.inner, self.B)
        return self._kron

    def mul_(self, y, a, alpha=1.0, beta=0.0):
        return self.kronecker().mul_(y, a, alpha, beta)

    def to_dense(self):
        return self.kronecker().to_dense()


class LazyMatrixProduct(LazyOperator):
    """LazyMatrixProduct(U, V') — the low-rank Gramian of a FiniteBasis kernel (src/mercer.jl:61-70);
    mul! applies the factors right to left (src/lazy_linear_algebra.jl:78-85): y = α U (Vᵀ a) + β y."""

    def __init__(self, U: torch.Tensor, V: torch.Tensor):
        # U: (n, r), V: (m, r) in torch indexing; stored column-major for the library
        self.U, self.V = U, V
        self.shape = (U.shape[0], V.shape[0])
        self.dtype, self.device = U.dtype, U.device
        self._Ucm = U.t().contiguous()
        self._Vcm = self._Ucm if V is U else V.t().contiguous()

    def mul_(self, y, a, alpha=1.0, beta=0.0):
        a = _vec_arg(a, self.shape[1], self.dtype, self.device, "a")
        n, m = self.shape
        r = self.U.shape[1]
        nrhs = 1 if a.dim() == 1 else a.shape[1]
        if y.shape[0] != n or y.dtype != self.dtype or (a.dim() == 2 and tuple(y.shape) != (n, nrhs)):
            raise _ffi.DimensionMismatch(_ffi.EINVAL, f"DimensionMismatch: y has shape {tuple(y.shape)}")
        # matrix right-hand sides go down in ONE call (column-major m x p / n x p): from 8 columns on both products run on the
        # matrix cores (csrc/lowrank.hip)
        a_c = a.contiguous() if a.dim() == 1 else a.t().contiguous()
        if a.dim() == 1:
            y_c = y if y.is_contiguous() else y.contiguous()
        else:
            y_c = y.t().contiguous() if beta != 0 else torch.empty((nrhs, n), dtype=self.dtype, device=self.device)
        ctx = get_ctx(self.device).bind_stream()
        _ffi.check(_ffi.lib().covgram_lowrank_mvm(ctx, _ffi._P(self._Ucm.data_ptr()), n, _ffi._P(self._Vcm.data_ptr()), m, n, m, r,
                                                  _dtype_code(self.dtype), _ffi._P(a_c.data_ptr()), m, _ffi._P(y_c.data_ptr()), n, nrhs,
                                                  float(alpha), float(beta), _ffi.DEVICE))
        if a.dim() == 2:
            y.copy_(y_c.t())
        elif y_c is not y:
            y.copy_(y_c)
        return y

    def to_dense(self):
        return self.U @ self.V.t()


def _by_columns(op, y, a, alpha, beta):
    """Matrix right-hand side for operators whose mul_ is written for vectors: one column at a time."""
    for c in range(a.shape[1]):
        yc = y[:, c].contiguous()
        op.mul_(yc, a[:, c].contiguous(), alpha, beta)
        y[:, c] = yc
    return y


class ScaledOperator(LazyOperator):
    """Diagonal(dx) · K · Diagonal(dy), lazy — gramian(::VerticalRescaling) (src/transformation.jl:165-171 builds exactly this
    LazyMatrixProduct(Dx, K, Dy)).  The two diagonal scalings are O(n) vector ops around K's own MVM."""

    def __init__(self, dx: torch.Tensor, K: LazyOperator, dy: torch.Tensor):
        self.dx, self.K, self.dy = dx, K, dy
        self.shape, self.dtype, self.device = K.shape, K.dtype, K.device

    def mul_(self, y, a, alpha=1.0, beta=0.0):
        a = _vec_arg(a, self.shape[1], self.dtype, self.device, "a")
        sa = self.dy * a if a.dim() == 1 else self.dy[:, None] * a
        t = self.K @ sa
        t = self.dx * t if t.dim() == 1 else self.dx[:, None] * t
        return y.copy_(alpha * t) if beta == 0 else y.mul_(beta).add_(t, alpha=alpha)

    def to_dense(self):
        return self.dx[:, None] * self.K.to_dense() * self.dy[None, :]


class LinearMapBlockGramian(LazyOperator):
    """Gradient Gramian of k(Ux, Uy): block (i, j) = Uᵀ B_ij(Ux, Uy) U (chain rule), applied as  b_i = Uᵀ Σ_j B_ij (U a_j):
    two small (d′ × d) maps around the device block MVM on the transformed points.  U: (d′, d) matrix or a length-d diagonal."""

    def __init__(self, U: torch.Tensor, inner: "BlockGramian", d: int):
        self.U, self.inner, self.d = U, inner, d
        n, m = inner.inner.shape
        self.shape = (n * d, m * d)
        self.dtype, self.device = inner.dtype, inner.device

    def mul_(self, y, a, alpha=1.0, beta=0.0):
        if torch.is_tensor(a) and a.dim() == 2:
            return _by_columns(self, y, a, alpha, beta)
        n, m = self.inner.inner.shape
        A = _vec_arg(a, self.shape[1], self.dtype, self.device, "a").reshape(m, self.d)
        UA = (A * self.U if self.U.dim() == 1 else A @ self.U.t()).contiguous()
        t = (self.inner @ UA.reshape(-1)).reshape(n, -1)
        b = (t * self.U if self.U.dim() == 1 else t @ self.U).reshape(-1)
        return y.copy_(alpha * b) if beta == 0 else y.mul_(beta).add_(b, alpha=alpha)


class CosineBlockGramian(LazyOperator):
    """Gradient Gramian of the Cosine kernel (src/gradient.jl:129-136): block (i, j) = −k₂ c cᵀ with k₂ = −4π² cos(2π c·(x_i − y_j)),
    i.e. (rank-2 scalar Gramian) ⊗ c cᵀ:  b_i = 4π² c Σ_j cos(u_i − v_j) (c·a_j)."""

    def __init__(self, c: torch.Tensor, scalar: "LazyMatrixProduct"):
        self.c, self.scalar = c, scalar
        d = c.shape[0]
        self.d = d
        self.shape = (scalar.shape[0] * d, scalar.shape[1] * d)
        self.dtype, self.device = scalar.dtype, scalar.device

    def mul_(self, y, a, alpha=1.0, beta=0.0):
        if torch.is_tensor(a) and a.dim() == 2:
            return _by_columns(self, y, a, alpha, beta)
        n, m = self.scalar.shape
        A = _vec_arg(a, self.shape[1], self.dtype, self.device, "a").reshape(m, self.d)
        t = self.scalar @ (A @ self.c).contiguous()
        b = ((4 * math.pi ** 2) * t[:, None] * self.c[None, :]).reshape(-1)
        return y.copy_(alpha * b) if beta == 0 else y.mul_(beta).add_(b, alpha=alpha)


class PointJacobianBlockGramian(LazyOperator):
    """Gradient Gramian of k(u(x), u(y)) for a pointwise warp u: block (i, j) = J_iᵀ B̂_ij J_j.  Here u is the NeuralNetwork
    normalisation x ↦ x̂ = [x, √σ] / ρ, ρ = sqrt(1 + |x|² + σ):  J a = [a; 0]/ρ − x̂ (x·a)/ρ²,  Jᵀ v = v[:d]/ρ − x (x̂·v)/ρ²."""

    def __init__(self, x: torch.Tensor, y: torch.Tensor, xh: torch.Tensor, yh: torch.Tensor, rx: torch.Tensor, ry: torch.Tensor,
                 inner: "BlockGramian"):
        self.x, self.y, self.xh, self.yh, self.rx, self.ry, self.inner = x, y, xh, yh, rx, ry, inner
        n, m, d = x.shape[0], y.shape[0], x.shape[1]
        self.d = d
        self.shape = (n * d, m * d)
        self.dtype, self.device = inner.dtype, inner.device

    def mul_(self, yv, a, alpha=1.0, beta=0.0):
        if torch.is_tensor(a) and a.dim() == 2:
            return _by_columns(self, yv, a, alpha, beta)
        n, m, d = self.x.shape[0], self.y.shape[0], self.d
        A = _vec_arg(a, self.shape[1], self.dtype, self.device, "a").reshape(m, d)
        ya = (self.y * A).sum(dim=1)
        JA = torch.cat([A, torch.zeros(m, 1, dtype=self.dtype, device=self.device)], dim=1) / self.ry[:, None] \
            - self.yh * (ya / self.ry ** 2)[:, None]
        V = (self.inner @ JA.reshape(-1).contiguous()).reshape(n, d + 1)
        xv = (self.xh * V).sum(dim=1)
        b = (V[:, :d] / self.rx[:, None] - self.x * (xv / self.rx ** 2)[:, None]).reshape(-1)
        return yv.copy_(alpha * b) if beta == 0 else yv.mul_(beta).add_(b, alpha=alpha)


class LazyMatrixSum(LazyOperator):
    """D + G kept lazy (src/gramian.jl:55-60, src/lazy_linear_algebra.jl:91-133): mul! accumulates the
    terms with β = 1 after the first."""

    def __init__(self, *args):
        self.args = []
        for a in args:
            self.args.extend(a.args if isinstance(a, LazyMatrixSum) else [a])
        lazy = [a for a in self.args if (hasattr(a, "mul_") and not torch.is_tensor(a))][0]
        self.shape, self.dtype, self.device = lazy.shape, lazy.dtype, lazy.device

    def mul_(self, y, a, alpha=1.0, beta=0.0):
        a = _vec_arg(a, self.shape[1], self.dtype, self.device, "a")
        first = True
        for A in self.args:
            b = beta if first else 1.0
            if (hasattr(A, "mul_") and not torch.is_tensor(A)):
                A.mul_(y, a, alpha, b)
            else:   # Diagonal given as a 1-D tensor, or a dense matrix
                A = torch.as_tensor(A, dtype=self.dtype, device=self.device)
                if A.dim() == 1 and b != 0:                   # the noise term of G + sigma^2 I inside a Krylov loop: ONE launch
                    if b != 1:
                        y.mul_(b)
                    y.addcmul_(A if a.dim() == 1 else A[:, None], a, value=alpha)
                    first = False
                    continue
                t = (A * a if a.dim() == 1 else A[:, None] * a) if A.dim() == 1 else A @ a
                if b == 0:
                    y.copy_(alpha * t)
                else:
                    y.mul_(b).add_(t, alpha=alpha)
            first = False
        return y


class Fill(LazyOperator):
    """gramian(k::Constant, x, y) = Fill(k.c, n, m) (src/stationary.jl:34)."""

    def __init__(self, c, n, m, dtype, device):
        self.c, self.shape, self.dtype, self.device = float(c), (n, m), dtype, device

    def mul_(self, y, a, alpha=1.0, beta=0.0):
        a = _vec_arg(a, self.shape[1], self.dtype, self.device, "a")
        s = a.sum(dim=0, keepdim=True) * (alpha * self.c)
        if beta == 0:
            y.copy_(s.expand_as(y) if a.dim() > 1 else s.expand(y.shape[0]))
        else:
            y.mul_(beta).add_(s.expand_as(y) if a.dim() > 1 else s.expand(y.shape[0]))
        return y

    def to_dense(self):
        return torch.full(self.shape, self.c, dtype=self.dtype, device=self.device)


# ----------------------------------------------------------------------------------------------
# the smart pseudo-constructor (src/gramian.jl:139-189 and the specialisations listed in SURVEY §3.1)
# ----------------------------------------------------------------------------------------------
def _first_column(k, x: torch.Tensor, y0: torch.Tensor) -> torch.Tensor:
    """k.(x, y0) for n points x against ONE point y0 — n kernel evaluations on the device."""
    return Gramian(k, x, y0.reshape(1, -1)).to_dense()[:, 0].contiguous()


def _transform_points(k, p: torch.Tensor) -> torch.Tensor:
    """u(x) for every point (one O(n d d′) pass on the device): ScaledInputKernel / ARD, Warped, Periodic."""
    if isinstance(k, K.ScaledInputKernel):
        U = torch.as_tensor(k.U, dtype=p.dtype, device=p.device)
        return (p * U if U.dim() == 1 else p @ U.t()).contiguous()
    if isinstance(k, K.Warped):
        if callable(k.u):
            return torch.as_tensor(k.u(p), dtype=p.dtype, device=p.device).reshape(p.shape[0], -1).contiguous()
        U = torch.as_tensor(k.u, dtype=p.dtype, device=p.device)
        return (p @ U.t()).contiguous()
    if isinstance(k, K.Periodic):
        if p.shape[1] != 1:
            raise _ffi.DimensionMismatch(_ffi.EINVAL, "Periodic: input has to be one-dimensional (src/transformation.jl:53)")
        ang = (2 * math.pi) * p[:, 0]
        return torch.stack([torch.cos(ang), torch.sin(ang)], dim=1).contiguous()
    raise TypeError(type(k))


def _cosine_lowrank(k, px: torch.Tensor, py: torch.Tensor, same: bool):
    c = torch.as_tensor(k.c, dtype=px.dtype, device=px.device)
    if c.numel() == 1 and px.shape[1] != 1:
        c = c.expand(px.shape[1])
    if c.shape[0] != px.shape[1]:
        raise _ffi.DimensionMismatch(_ffi.EINVAL, f"Cosine: length(c) = {c.shape[0]} ≠ d = {px.shape[1]}")
    u = (2 * math.pi) * (px @ c)
    U = torch.stack([torch.cos(u), torch.sin(u)], dim=1)
    if same:
        return LazyMatrixProduct(U, U)
    v = (2 * math.pi) * (py @ c)
    return LazyMatrixProduct(U, torch.stack([torch.cos(v), torch.sin(v)], dim=1))


def gramian(k, x=None, y=None, trait: Optional[K.InputTrait] = None):
    """gramian(k, x[, y][, trait]) — picks the representation exactly as the reference does."""
    if x is None:
        raise TypeError("gramian(k, x[, y])")
    if not isinstance(k, (K.AbstractKernel,)) and not callable(k):   # gramian(x, y) = Gramian(Dot(), x, y)  (:150-151)
        return Gramian(K.Dot(), k, x)
    periodic = isinstance(y, K.PeriodicInput) or isinstance(trait, K.PeriodicInput)
    if isinstance(y, K.InputTrait):
        trait, y = y, None
    same = y is None or y is x

    if isinstance(k, K.Constant):                               # src/stationary.jl:34
        px = _as_points(x)
        py = px if same else _as_points(y, device=px.device, dtype=px.dtype)
        return Fill(k.c, px.shape[0], py.shape[0], px.dtype, px.device)

    if isinstance(k, K.FiniteBasis):                            # src/mercer.jl:61-70
        px = _as_points(x)
        py = px if same else _as_points(y, device=px.device, dtype=px.dtype)
        r = len(k.basis)
        if px.shape[0] > r and py.shape[0] > r:
            def basis(p):
                arg = p[:, 0] if p.shape[1] == 1 else p
                return torch.stack([torch.as_tensor(b(arg), dtype=p.dtype, device=p.device).reshape(-1) for b in k.basis], dim=1)
            U = basis(px)
            V = U if same else basis(py)
            return LazyMatrixProduct(U, V)
        raise _ffi.UnsupportedKernel(_ffi.EUNSUPPORTED, "FiniteBasis with fewer points than basis functions is a GenericInput Gramian (src/mercer.jl:68)")

    # ---- input / output transformations: the points are transformed ONCE, the hot path runs on the result -------------------
    if isinstance(k, K.VerticalRescaling):                     # src/transformation.jl:165-171
        px = _as_points(x)
        py = px if same else _as_points(y, device=px.device, dtype=px.dtype)
        fx = torch.as_tensor(k.f(px), dtype=px.dtype, device=px.device).reshape(-1)
        fy = fx if same else torch.as_tensor(k.f(py), dtype=px.dtype, device=px.device).reshape(-1)
        return ScaledOperator(fx, gramian(k.k, px, None if same else py), fy)
    if isinstance(k, (K.ScaledInputKernel, K.Warped, K.Periodic)):   # src/transformation.jl:82-90, 114-118, 54-65
        px = _as_points(x)
        py = px if same else _as_points(y, device=px.device, dtype=px.dtype)
        tx = _transform_points(k, px)
        return gramian(k.k, tx, None if same else _transform_points(k, py))
    if isinstance(k, K.NeuralNetwork) or (isinstance(k, K.GradientKernel) and isinstance(k.k, K.NeuralNetwork)):
        nn = k if isinstance(k, K.NeuralNetwork) else k.k          # src/mercer.jl:82-85 on normalised augmented inputs
        px = _as_points(x)
        py = px if same else _as_points(y, device=px.device, dtype=px.dtype)
        def warp(p):
            rho = torch.sqrt(1 + (p * p).sum(dim=1) + nn.sigma)
            z = torch.cat([p, torch.full((p.shape[0], 1), math.sqrt(nn.sigma), dtype=p.dtype, device=p.device)], dim=1)
            return (z / rho[:, None]).contiguous(), rho
        xh, rx = warp(px)
        yh, ry = (xh, rx) if same else warp(py)
        if isinstance(k, K.NeuralNetwork):
            return Gramian(K.AsinDot(), xh, None if same else yh)
        return PointJacobianBlockGramian(px, py, xh, yh, rx, ry, BlockGramian(K.GradientKernel(K.AsinDot()), xh, None if same else yh))
    if isinstance(k, K.CosineKernel):                           # rank 2: cos(u_i − v_j) = cos u_i cos v_j + sin u_i sin v_j
        px = _as_points(x)
        py = px if same else _as_points(y, device=px.device, dtype=px.dtype)
        return _cosine_lowrank(k, px, py, same)
    if isinstance(k, K.GradientKernel) and isinstance(k.k, K.ScaledInputKernel):
        px = _as_points(x)
        py = px if same else _as_points(y, device=px.device, dtype=px.dtype)
        U = torch.as_tensor(k.k.U, dtype=px.dtype, device=px.device)
        tx = _transform_points(k.k, px)
        inner = BlockGramian(K.GradientKernel(k.k.k), tx, None if same else _transform_points(k.k, py))
        return LinearMapBlockGramian(U, inner, px.shape[1])
    if isinstance(k, K.GradientKernel) and isinstance(k.k, K.CosineKernel):
        px = _as_points(x)
        py = px if same else _as_points(y, device=px.device, dtype=px.dtype)
        return CosineBlockGramian(torch.as_tensor(k.k.c, dtype=px.dtype, device=px.device), _cosine_lowrank(k.k, px, py, same))

    if isinstance(k, (K.GradientKernel, K.ValueGradientKernel)):   # src/gramian.jl:120-123
        return BlockGramian(k, x, None if same else y)
    if isinstance(k, K.SeparableKernel):
        return SeparableGramian(k, x, None if same else y)

    if isinstance(k, K.SeparableProduct) and isinstance(x, LazyGrid):   # src/algebra.jl:91-95
        gy = x if same else y
        if not isinstance(gy, LazyGrid) or len(gy.args) != len(x.args):
            raise _ffi.DimensionMismatch(_ffi.EINVAL, f"length(X.args) = {len(x.args)} ≠ length(Y.args)")
        if len(k.args) != len(x.args):
            raise _ffi.DimensionMismatch(_ffi.EINVAL, f"SeparableProduct needs d = {len(x.args)} kernels but has r = {len(k.args)}")
        return KroneckerProduct(*[gramian(ki, xi, None if yi is xi else yi) for ki, xi, yi in zip(k.args, x.args, gy.args)])

    if isinstance(x, StepRangeLen) and (same or isinstance(y, StepRangeLen)):   # src/gramian.jl:167-189
        tr = trait if trait is not None else K.input_trait(k)
        dev = get_ctx().device
        if periodic and isinstance(k, K.StationaryKernel):      # :186-189
            xt = _as_points(x, dev)
            return Circulant(_first_column(k, xt, xt[0]))
        if isinstance(tr, (K.IsotropicInput, K.StationaryInput)) and K.device_spec(k) is not None:
            xt = _as_points(x, dev)
            if same:
                return SymmetricToeplitz(_first_column(k, xt, xt[0]))          # k.(x[1], x)
            if x.step == y.step and len(x) >= 1:
                yt = _as_points(y, dev, xt.dtype)
                vc = _first_column(k, xt, yt[0])                                  # k.(x, y[1])
                vr = _first_column(k, yt, xt[0])                                  # k.(x[1], y)
                return Toeplitz(vc, vr)
        return Gramian(k, x, None if same else y)               # different step / GenericInput: plain Gramian

    return Gramian(k, x, None if same else y)
