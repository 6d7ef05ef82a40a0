"""Dense / low-rank factorizations of a lazy Gramian: `cholesky`, `factorize`.

The reference instantiates `Matrix(G)` and calls LAPACK (src/gramian.jl:192-213); its own comment asks for "a special
cholesky implementation to avoid instantiating G in the low rank case" (:191).  Here
  * `cholesky(G)`               = device `Matrix(G)` tile (covgram_matrix) handed to torch.linalg.cholesky (rocSOLVER);
  * `cholesky(G, pivoted=True)` = that special implementation: a LAZY diagonally pivoted Cholesky that evaluates only the
                                   diagonal and one Gramian column per step (n kernel evaluations on the device each), so a
                                   rank-r factor of an n x n Gramian costs O(n r) kernel evaluations and O(n r^2) flops and
                                   never forms the matrix;
  * `factorize(G)`              = pivoted Cholesky with tol = 1e-6 up to n = 2^14, else the lazy G itself (CG path).
Everything besides the kernel evaluations is torch plumbing on the same stream.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import kernels as K
from .gramian import Gramian, LazyOperator

DEFAULT_MAX_CHOLESKY_SIZE = 2 ** 14     # src/gramian.jl:201
DEFAULT_TOL = 1e-6                      # src/gramian.jl:202


class CholeskyFactor(LazyOperator):
    """Lower factor L (n x n) with G = L L'."""

    def __init__(self, L: torch.Tensor):
        self.L = L
        self.shape = (L.shape[0], L.shape[0])
        self.dtype, self.device = L.dtype, L.device

    def to_dense(self):
        return self.L @ self.L.T

    def solve(self, b: torch.Tensor) -> torch.Tensor:
        return torch.cholesky_solve(b.reshape(b.shape[0], -1), self.L).reshape(b.shape)

    def mul_(self, y, a, alpha=1.0, beta=0.0):
        t = self.L @ (self.L.T @ a)
        return y.mul_(beta).add_(t, alpha=alpha) if beta != 0 else y.copy_(alpha * t)


class PivotedCholesky(LazyOperator):
    """P' G P ≈ L L' of rank r: `L` is n x r in ORIGINAL row order (row piv[k] is the k-th pivot), `piv`, `rank`."""

    def __init__(self, L: torch.Tensor, piv: torch.Tensor, rank: int):
        self.L, self.piv, self.rank = L, piv, rank
        self.shape = (L.shape[0], L.shape[0])
        self.dtype, self.device = L.dtype, L.device

    def to_dense(self):
        return self.L @ self.L.T

    def mul_(self, y, a, alpha=1.0, beta=0.0):
        t = self.L @ (self.L.T @ a)
        return y.mul_(beta).add_(t, alpha=alpha) if beta != 0 else y.copy_(alpha * t)


def diagonal(G: Gramian) -> torch.Tensor:
    """diag(G) for a square Gramian with x ≡ y: k(x_i, x_i) evaluated on the device (n kernel evaluations)."""
    n = G.shape[0]
    tr = K.input_trait(G.k)
    if isinstance(tr, K.IsotropicInput):                      # phi(0) for every point
        v = Gramian(G.k, G.x[:1], G.x[:1]).to_dense()[0, 0]
        return v.expand(n).clone()
    if isinstance(tr, K.DotProductInput):                     # phi(|x_i|^2): a 1-d dot-product Gramian against the point 1
        z = (G.x * G.x).sum(dim=1, keepdim=True)
        one = torch.ones(1, 1, dtype=G.dtype, device=G.device)
        return Gramian(G.k, z, one).to_dense()[:, 0].contiguous()
    raise NotImplementedError("diagonal: GenericInput kernels have no device path")


def cholesky(G: LazyOperator, pivoted: bool = False, check: bool = True, tol: float = 0.0, max_rank: Optional[int] = None):
    """LinearAlgebra.cholesky(G::Gramian, Val(pivoted); check, tol) (src/gramian.jl:192-199)."""
    n, m = G.shape
    if n != m:
        raise ValueError("DimensionMismatch: matrix is not square")
    if not pivoted:
        L, info = torch.linalg.cholesky_ex(G.to_dense())
        if check and int(info) != 0:
            raise ValueError(f"PosDefException: matrix is not positive definite; Cholesky factorization failed at {int(info)}")
        return CholeskyFactor(L)
    if not isinstance(G, Gramian) or not G.issymmetric():
        raise NotImplementedError("pivoted cholesky: symmetric Gramian expected")
    max_rank = n if max_rank is None else min(max_rank, n)
    d = diagonal(G)
    live = torch.ones(n, dtype=torch.bool, device=G.device)
    L = torch.zeros((n, max_rank), dtype=G.dtype, device=G.device)
    piv, rank = [], 0
    neg_inf = torch.tensor(float("-inf"), dtype=G.dtype, device=G.device)
    for k in range(max_rank):
        dm = torch.where(live, d, neg_inf)
        p = int(torch.argmax(dm))                             # one host sync per pivot (the stopping test needs it anyway)
        dmax = float(dm[p])
        if not dmax > tol:                                    # LAPACK pstrf: stop at the first pivot <= tol
            break
        col = Gramian(G.k, G.x, G.x[p:p + 1]).to_dense()[:, 0]                   # G[:, p]: n kernel evaluations, on the device
        if k:
            col = col - L[:, :k] @ L[p, :k]
        L[:, k] = col / (dmax ** 0.5)
        d = d - L[:, k] ** 2
        live[p] = False
        piv.append(p)
        rank = k + 1
    rest = torch.nonzero(live).flatten().tolist()
    return PivotedCholesky(L[:, :rank].contiguous(), torch.tensor(piv + rest, device=G.device), rank)


def factorize(G: LazyOperator, max_cholesky_size: int = DEFAULT_MAX_CHOLESKY_SIZE, tol: float = DEFAULT_TOL):
    """LinearAlgebra.factorize(G::Gramian) (src/gramian.jl:205-213): pivoted Cholesky (detects low rank) up to
    max_cholesky_size, otherwise G stays lazy and solves go through CG."""
    n, m = G.shape
    if n != m:
        raise ValueError("DimensionMismatch: matrix is not square")
    if n <= max_cholesky_size and isinstance(G, Gramian) and G.issymmetric():
        return cholesky(G, pivoted=True, check=False, tol=tol)
    return G
