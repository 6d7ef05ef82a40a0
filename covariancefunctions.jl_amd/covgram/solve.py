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
       maxiter: Optional[int] = None) -> Tuple[torch.Tensor, dict]:
    """Solve A x = b for a symmetric positive definite lazy operator A (cg!, IterativeSolvers 0.9.2 semantics:
    stops when ‖r‖ ≤ max(reltol·‖r₀‖, abstol)).  Returns (x, {"iterations", "residual_norm", "converged"})."""
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
    p = r.clone()
    rs = torch.dot(r, r)
    r0 = float(rs.sqrt())
    tol = max(reltol * r0, abstol)
    maxiter = n if maxiter is None else maxiter
    it, res = 0, r0
    while it < maxiter and res > tol:
        A.mul_(Ap, p)                       # the hot path
        alpha = rs / torch.dot(p, Ap)
        x.add_(p, alpha=float(alpha))
        r.add_(Ap, alpha=-float(alpha))
        rs_new = torch.dot(r, r)
        p.mul_(float(rs_new / rs)).add_(r)
        rs = rs_new
        res = float(rs.sqrt())
        it += 1
    return x, {"iterations": it, "residual_norm": res, "converged": res <= tol}


def solve(A: LazyOperator, b: torch.Tensor, **kw) -> torch.Tensor:
    """`A \\ b` for lazy Gramians (src/gramian.jl:229-238)."""
    return cg(A, b, **kw)[0]
