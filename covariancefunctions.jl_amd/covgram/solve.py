"""Krylov callers of the hot path: conjugate gradients for `G \\ b` and `(G + σ²I) \\ b`.

The reference solves lazy (block) Gramians with IterativeSolvers.cg! (src/gramian.jl:229-238,
src/lazy_linear_algebra.jl:135-144).  Here every MVM is a device kernel of libcovgram and all vectors stay resident on
the GPU; the O(n) vector updates and the two dot products per iteration are torch ops on the same stream (plumbing).
One host synchronisation per iteration (the convergence test), none inside the MVM.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .gramian import LazyOperator


def cg(A: LazyOperator, b: torch.Tensor, x0: Optional[torch.Tensor] = None, reltol: float = 1e-8, abstol: float = 0.0,
       maxiter: Optional[int] = None, precond=None) -> Tuple[torch.Tensor, dict]:
    """Solve A x = b for a symmetric positive definite lazy operator A (cg!, IterativeSolvers 0.9.2 semantics:
    stops when ‖r‖ ≤ max(reltol·‖r₀‖, abstol)); `precond(r)` applies an SPD preconditioner M⁻¹ (Pl = M in the reference's
    keyword).  Returns (x, {"iterations", "residual_norm", "converged"}).  All scalars of the recurrence stay on the device;
    the only host synchronisation per iteration is the convergence test."""
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
