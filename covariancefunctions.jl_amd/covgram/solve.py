"""Krylov callers of the hot path: conjugate gradients for `G \\ b` and `(G + σ²I) \\ b`.

The reference solves lazy (block) Gramians with IterativeSolvers.cg! (src/gramian.jl:229-238,
src/lazy_linear_algebra.jl:135-144).  Here every MVM is a device kernel of libcovgram and all vectors stay resident on
the GPU.  Without a preconditioner the O(n) vector updates and the two dot products of an iteration are ONE library call
(covgram_cg_step_shifted: one launch for vectors that fit a workgroup's registers, three above, scalars on the device; the
diagonal term of G + σ²I and ‖r‖ ride along) — as torch ops they were eleven small launches, 45 us next to a 40 us MVM at
n = 16384; with a preconditioner they stay torch ops on the same stream.  One host synchronisation per
iteration (the convergence test), none inside the MVM.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _ffi
from .gramian import LazyOperator, get_ctx, _dtype_code


NORM_SLOT = 2 + 512            # scal[NORM_SLOT] = |r| after a step (include/covgram.h: covgram_cg_step_shifted)


def _split_shift(A):
    """A = G + Diagonal(d) with one lazy term and one 1-D tensor -> (G, d); otherwise (A, None).  The library's CG step adds the
    diagonal term to Ap inside its first launch (covgram_cg_step_shifted), so the MVM of the iteration is the Gramian's alone."""
    from .gramian import LazyMatrixSum
    if isinstance(A, LazyMatrixSum) and len(A.args) == 2:
        lazy = [t for t in A.args if hasattr(t, "mul_") and not torch.is_tensor(t)]
        diag = [t for t in A.args if torch.is_tensor(t) and t.dim() == 1]
        if len(lazy) == 1 and len(diag) == 1:
            return lazy[0], diag[0]
    return A, None


def _fused_step(A, x, r, p, Ap):
    """The iteration's work through the library when everything is a contiguous CUDA vector of one supported dtype:
    returns (iterate, scal) or None; iterate() = the MVM Ap = G p + covgram_cg_step_shifted (alpha, x, r, rho', p, |r|: three
    launches).  scal[1] = |r|^2 and scal[NORM_SLOT] = |r| after every step."""
    if not (x.is_cuda and x.dim() == 1 and x.dtype in (torch.float32, torch.float64)):
        return None
    if not all(t.is_contiguous() and t.dtype == x.dtype and t.device == x.device for t in (r, p, Ap)):
        return None
    G, diag = _split_shift(A)
    if diag is not None:
        diag = diag.to(device=x.device, dtype=x.dtype).contiguous()
        if diag.shape[0] != x.shape[0]:
            G, diag = A, None
    lib = _ffi.lib()
    scal = torch.zeros(NORM_SLOT + 1, dtype=x.dtype, device=x.device)
    scal[1] = torch.dot(r, r)
    ctx = get_ctx(x.device)
    code, n = _dtype_code(x.dtype), x.shape[0]
    P = _ffi._P
    dptr = P(diag.data_ptr()) if diag is not None else None

    def iterate():
        G.mul_(Ap, p)                       # the hot path
        _ffi.check(lib.covgram_cg_step_shifted(ctx.bind_stream(), n, code, P(x.data_ptr()), P(r.data_ptr()), P(p.data_ptr()),
                                               P(Ap.data_ptr()), P(scal.data_ptr()), dptr))
    iterate.keep = (diag, G)                # (the captured graph holds raw pointers)
    return iterate, scal


def cg(A: LazyOperator, b: torch.Tensor, x0: Optional[torch.Tensor] = None, reltol: float = 1e-8, abstol: float = 0.0,
       maxiter: Optional[int] = None, precond=None, graph: bool = False, check_every: int = 8) -> Tuple[torch.Tensor, dict]:
    """Solve A x = b for a symmetric positive definite lazy operator A (cg!, IterativeSolvers 0.9.2 semantics:
    stops when ‖r‖ ≤ max(reltol·‖r₀‖, abstol)); `precond(r)` applies an SPD preconditioner M⁻¹ (Pl = M in the reference's
    keyword).  Returns (x, {"iterations", "residual_norm", "converged"}).  All scalars of the recurrence stay on the device;
    the only host synchronisation per iteration is the convergence test.

    graph=True (small, launch-bound systems): `check_every` iterations — the MVM's kernels and the vector updates — are captured
    once into a HIP graph and replayed; the residual is read back between replays, so the solve may run up to
    check_every − 1 iterations past the tolerance (they only refine x)."""
    if graph:
        return _cg_graph(A, b, x0, reltol, abstol, maxiter, precond, max(1, int(check_every)))
    n = A.shape[0]
    if A.shape[0] != A.shape[1] or b.shape[0] != n:
        raise ValueError("cg: A must be square and match b")
    b = b.to(device=A.device, dtype=A.dtype)
    x = torch.zeros_like(b) if x0 is None else x0.to(device=A.device, dtype=A.dtype).clone()
    r = b.clone()
    Ap = torch.empty_like(b)
    if x0 is not None:
        A.mul_(Ap, x)
        r -= Ap
    z = precond(r) if precond is not None else r
    p = z.clone()
    rz = torch.dot(r, z)
    r0 = float(torch.linalg.vector_norm(r))
    tol = max(reltol * r0, abstol)
    maxiter = n if maxiter is None else maxiter
    it, res = 0, r0
    fused = _fused_step(A, x, r, p, Ap) if precond is None else None
    if fused is not None:
        iterate, scal = fused
        while it < maxiter and res > tol:
            iterate()                       # Ap = A p; alpha, x, r, rho', p
            res = float(scal[NORM_SLOT])    # the convergence test: the iteration's one synchronisation
            it += 1
        return x, {"iterations": it, "residual_norm": res, "converged": res <= tol}
    while it < maxiter and res > tol:
        A.mul_(Ap, p)                       # the hot path
        alpha = rz / torch.dot(p, Ap)       # 0-dim device tensors: no synchronisation
        x.addcmul_(p, alpha)
        r.addcmul_(Ap, -alpha)
        z = precond(r) if precond is not None else r
        rz_new = torch.dot(r, z)
        p.mul_(rz_new / rz).add_(z)
        rz = rz_new
        res = float(torch.linalg.vector_norm(r))
        it += 1
    return x, {"iterations": it, "residual_norm": res, "converged": res <= tol}


def _cg_graph(A, b, x0, reltol, abstol, maxiter, precond, check_every):
    """cg with the iteration body as a replayed HIP graph (torch.cuda.CUDAGraph on ROCm = hipGraph).  The body is warmed up
    once eagerly (one real iteration: libcovgram sizes its workspaces and fragment caches outside the capture), then captured;
    every tensor the body touches is persistent, scalars are 0-dim device tensors."""
    n = A.shape[0]
    if A.shape[0] != A.shape[1] or b.shape[0] != n:
        raise ValueError("cg: A must be square and match b")
    b = b.to(device=A.device, dtype=A.dtype)
    x = torch.zeros_like(b) if x0 is None else x0.to(device=A.device, dtype=A.dtype).clone()
    r = b.clone()
    Ap = torch.empty_like(b)
    if x0 is not None:
        A.mul_(Ap, x)
        r -= Ap
    z = precond(r) if precond is not None else r
    p = z.clone()
    rz = torch.dot(r, z).clone()
    res = torch.linalg.vector_norm(r).clone()
    r0 = float(res)
    tol = max(reltol * r0, abstol)
    maxiter = n if maxiter is None else maxiter
    it = 0
    if not (r0 > tol) or maxiter <= 0:
        return x, {"iterations": 0, "residual_norm": r0, "converged": r0 <= tol}

    fused = _fused_step(A, x, r, p, Ap) if precond is None else None

    def body():
        if fused is not None:
            fused[0]()                      # |r| is left in scal[NORM_SLOT]
            return
        A.mul_(Ap, p)
        alpha = rz / torch.dot(p, Ap)
        x.addcmul_(p, alpha)
        r.addcmul_(Ap, -alpha)
        zz = precond(r) if precond is not None else r
        rz_new = torch.dot(r, zz)
        p.mul_(rz_new / rz).add_(zz)
        rz.copy_(rz_new)
        res.copy_(torch.linalg.vector_norm(r))

    side = torch.cuda.Stream(device=A.device)
    side.wait_stream(torch.cuda.current_stream(A.device))
    with torch.cuda.stream(side):
        body(); it += 1                                    # eager warm-up = iteration 1 (on the capture stream)
    torch.cuda.current_stream(A.device).wait_stream(side)
    # ONE graph holds check_every iterations: a replay costs the host tens of microseconds whatever it holds (n = 8192 fp32: 62 us per
    # replayed single iteration against 25 us of kernels), and the residual is only looked at between replays anyway
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(check_every):
            body()
    if fused is not None:
        res = fused[1][NORM_SLOT]
    resf = float(res)
    while it + check_every <= maxiter and resf > tol:
        g.replay()
        it += check_every
        resf = float(res)
    while it < maxiter and resf > tol:      # fewer than check_every iterations left of maxiter: eagerly
        body()
        it += 1
        resf = float(res)
    return x, {"iterations": it, "residual_norm": resf, "converged": resf <= tol, "graph": True}


def solve(A: LazyOperator, b: torch.Tensor, **kw) -> torch.Tensor:
    """`A \\ b` for lazy Gramians (src/gramian.jl:229-238)."""
    return cg(A, b, **kw)[0]


def toeplitz_solve(T, b: torch.Tensor, reltol: float = 1e-12, maxiter: Optional[int] = None):
    """x = T \\ b for a symmetric positive definite Toeplitz operator.

    The reference's direct solvers (`levinson`, src/toeplitz.jl:77-111) are O(n²) recurrences of n−1 dependent steps — no
    parallelism to give a GPU.  The device answer is preconditioned CG over the O(n log n) FFT MVM of `covgram_toeplitz_mvm`
    with T. Chan's optimal circulant preconditioner c_k = ((n−k) t_k + k t_{n−k}) / n, applied by FFT (torch.fft → rocFFT);
    it is positive definite whenever T is.  Returns (x, info)."""
    vc = T.vc
    n = vc.shape[0]
    if getattr(T, "vr", None) is not None or T.shape[0] != T.shape[1]:
        raise NotImplementedError("toeplitz_solve: symmetric Toeplitz expected")
    k = torch.arange(n, device=vc.device, dtype=vc.dtype)
    c = ((n - k) * vc + k * torch.roll(vc.flip(0), 1)) / n          # t_{n-k} with t_n := t_0 at k = 0 (weight 0)
    lam = torch.fft.rfft(c)                                          # real spectrum of the symmetric circulant
    inv = 1.0 / lam.real
    def precond(r):
        return torch.fft.irfft(torch.fft.rfft(r) * inv, n)
    return cg(T, b, reltol=reltol, maxiter=maxiter if maxiter is not None else 10 * n, precond=precond)


# ---- the reference's direct Toeplitz solvers on the device (src/toeplitz.jl:12-111; csrc/toeplitz_direct.hip) ---------------
def _unit_diagonal(r_or_T, dtype=None):
    """(r, r0): r = first column without its leading entry, scaled to a unit diagonal; a SymmetricToeplitz operator T gives
    r = T.vc[1:] / T.vc[0] and r0 = T.vc[0] (src/toeplitz.jl:40-46,100-111 — the reference tests `r_0 == 1` where it means
    `r_0 != 1`; the normalisation is done right here, as in the oracle)."""
    from .gramian import get_ctx
    if hasattr(r_or_T, "vc"):
        if getattr(r_or_T, "vr", None) is not None:
            raise NotImplementedError("direct Toeplitz solvers: symmetric Toeplitz expected")
        vc = r_or_T.vc
        return (vc[1:] / vc[0]).contiguous(), float(vc[0])
    r = torch.as_tensor(r_or_T)
    if not r.is_cuda:
        r = r.to(get_ctx().device)
    if dtype is not None:
        r = r.to(dtype)
    return r.contiguous(), 1.0


def durbin(r: torch.Tensor) -> torch.Tensor:
    """durbin(r): y = K \\ (-r), K = SymmetricToeplitz([1, r[1:end-1]]) (src/toeplitz.jl:12-27)."""
    from . import _ffi
    from .gramian import get_ctx, _dtype_code
    r, _ = _unit_diagonal(r)
    y = torch.empty_like(r)
    ctx = get_ctx(r.device).bind_stream()
    _ffi.check(_ffi.lib().covgram_toeplitz_durbin(ctx, _ffi._P(r.data_ptr()), r.shape[0], _ffi._P(y.data_ptr()), _dtype_code(r.dtype), _ffi.DEVICE))
    return y


def levinson(r_or_T, b: torch.Tensor) -> torch.Tensor:
    """levinson(r, b) = SymmetricToeplitz([1; r]) \\ b, or levinson(T, b) = T \\ b (src/toeplitz.jl:75-111): the O(n²) chain of
    n - 1 dependent steps on one workgroup.  For large n `toeplitz_solve` (PCG over the FFT MVM) is the faster path."""
    from . import _ffi
    from .gramian import get_ctx, _dtype_code
    r, r0 = _unit_diagonal(r_or_T, b.dtype if torch.is_tensor(b) else None)
    b = torch.as_tensor(b).to(device=r.device, dtype=r.dtype).contiguous()
    n = b.shape[0]
    if r.shape[0] != n - 1:
        raise _ffi.DimensionMismatch(_ffi.EINVAL, f"DimensionMismatch: length(b) = {n} ≠ {r.shape[0] + 1} = length(r) + 1")
    x = torch.empty_like(b)
    ctx = get_ctx(r.device).bind_stream()
    _ffi.check(_ffi.lib().covgram_toeplitz_levinson(ctx, _ffi._P(r.data_ptr()), _ffi._P(b.data_ptr()), n, _ffi._P(x.data_ptr()), _dtype_code(r.dtype), _ffi.DEVICE))
    return x / r0 if r0 != 1.0 else x


def trench(r_or_T) -> torch.Tensor:
    """trench(r) = inv(SymmetricToeplitz([1; r])), or trench(T) = inv(T) (src/toeplitz.jl:29-71), as a full symmetric matrix."""
    from . import _ffi
    from .gramian import get_ctx, _dtype_code
    r, r0 = _unit_diagonal(r_or_T)
    n = r.shape[0] + 1
    Bt = torch.empty((n, n), dtype=r.dtype, device=r.device)       # column-major n x n == the transpose of a C-contiguous tensor
    ctx = get_ctx(r.device).bind_stream()
    _ffi.check(_ffi.lib().covgram_toeplitz_trench(ctx, _ffi._P(r.data_ptr()) if n > 1 else _ffi._P(Bt.data_ptr()), n, _ffi._P(Bt.data_ptr()), n,
                                                  _dtype_code(r.dtype), _ffi.DEVICE))
    B = Bt.t()
    return B / r0 if r0 != 1.0 else B
