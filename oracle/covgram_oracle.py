"""CPU oracle for the lazy-Gramian MVM hot path of CovarianceFunctions.jl (reference v0.3.5).

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (`covariancefunctions.jl_amd/`)
imports this file; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may.  It is a plain numpy float64 (or float32 on request)
restatement of the reference algorithm; every function cites the reference
`file:line` (relative to /root/reference) it follows.

PARITY PIN STATUS ("parity unpinned" in the strict sense of the task statement):
the reference is pure Julia, Julia is not installed in the build container, and the
reference ships NO golden vectors (all of its tests use unseeded randn and assert
relations such as `G*a ≈ Matrix(G)*a`).  The oracle is therefore pinned by
  (1) the reference's own test *relations*, re-asserted in tests/test_oracle.py;
  (2) closed-form known answers (k(0)=1, MaternP(0)=exp(-r), EQ block = k(I - r r'));
  (3) an independent 50-digit mpmath evaluation of the same definitions
      (oracle/make_golden.py, committed together with the vectors it wrote).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from fractions import Fraction

import numpy as np

# ----------------------------------------------------------------------------------------
# kernel description (mirrors include/covgram.h :: covgram_kernel)
# ----------------------------------------------------------------------------------------
EQ, EXP, RQ, GAMMAEXP, CAUCHY, IMQ, MATERNP, DOT, EXPDOT, MATERN, ASINDOT = range(11)
FAMILY_NAMES = ["EQ", "EXP", "RQ", "GAMMAEXP", "CAUCHY", "IMQ", "MATERNP", "DOT", "EXPDOT", "MATERN", "ASINDOT"]
ISOTROPIC, DOTPRODUCT = 1, 2
CONSTANT = 100   # a factor of a Composite that is just its `scale` (stationary.jl:27-34)


@dataclass(frozen=True)
class Kernel:
    """family/trait/p/power/param/lengthscale/scale — same fields as the C struct."""
    family: int
    p: int = 0            # MaternP order (stationary.jl:117-121)
    power: int = 1        # Power exponent (algebra.jl:50-63); 1 = none
    param: float = 0.0    # RQ alpha (stationary.jl:45-53) | gammaExp gamma (:63-71) | IMQ c (:231-235) | Matern nu (:87-114)
    lengthscale: float = 1.0  # Lengthscale(k, l): s <- s / l^2 (transformation.jl:6-19)
    scale: float = 1.0    # Constant(c) * k (algebra.jl:23-25, stationary.jl:15-32)

    @property
    def trait(self) -> int:
        # properties.jl:39-43: IsotropicKernel -> IsotropicInput, Dot/ExponentialDot -> DotProductInput,
        # Power inherits the trait of its base kernel.
        return DOTPRODUCT if self.family in (DOT, EXPDOT, ASINDOT) else ISOTROPIC


@dataclass(frozen=True)
class Composite:
    """Sum (algebra.jl:27-47) of Products (algebra.jl:5-25) of kernels that share one input trait
    (properties.jl:47-63): k = scale * sum_t prod_f terms[t][f].  Power(k, p) (algebra.jl:50-63) of a single
    profile is that factor's `power`.  Mirrors include/covgram.h :: covgram_kernel_composite."""
    terms: tuple           # tuple of tuples of Kernel
    trait: int = ISOTROPIC
    scale: float = 1.0


# ----------------------------------------------------------------------------------------
# MaternP tables (stationary.jl:117-191)
# ----------------------------------------------------------------------------------------
MATERNP_MAX_P = 8


def maternp_coefficients(p: int):
    """stationary.jl:184-191: reverse([binomial(p,i) * (factorial(p+i) ÷ factorial(p)) for i in 1:p]).
    Entry k (0-based) multiplies (2r)^k; the (2r)^p coefficient is the implicit 1."""
    fp = math.factorial(p)
    c = [Fraction(math.comb(p, i) * (math.factorial(p + i) // fp)) for i in range(1, p + 1)]
    return list(reversed(c))


def maternp_norm(p: int) -> int:
    """stationary.jl:157: factorial(2p) ÷ factorial(p)."""
    return math.factorial(2 * p) // math.factorial(p)


def maternp_poly(p: int):
    """Normalised polynomial q_p(r) = sum_m h[m] r^m with MaternP_p(s) = exp(-r) q_p(r), r = sqrt((2p+1)s).
    (Expanded form of stationary.jl:147-157.)"""
    c = maternp_coefficients(p) + [Fraction(1)]
    nrm = maternp_norm(p)
    return [c[m] * (2 ** m) / nrm for m in range(p + 1)]


def _exp_neg_series(order: int):
    return [Fraction((-1) ** k, math.factorial(k)) for k in range(order + 1)]


def maternp_derivatives_at_zero(p: int):
    """stationary.jl:172-182 (SymEngine there): d_i = d^i/d(r²)^i MaternP(r², p) at r² = 0, i = 1..p.
    Derived here exactly with rationals from the power series of exp(-r) q_p(r): the odd powers of r
    below r^(2p+1) cancel, and the r^(2i) coefficient times (2p+1)^i is d_i / i!."""
    h = maternp_poly(p)
    e = _exp_neg_series(2 * p)
    c = 2 * p + 1
    d = []
    for i in range(1, p + 1):
        coef = sum(h[m] * e[2 * i - m] for m in range(0, min(p, 2 * i) + 1))
        d.append(coef * c ** i * math.factorial(i))
    # sanity: odd coefficients vanish
    for odd in range(1, 2 * p, 2):
        assert sum(h[m] * e[odd - m] for m in range(0, min(p, odd) + 1)) == 0
    return d


def _eps(dtype) -> float:
    return float(np.finfo(dtype).eps)


# ----------------------------------------------------------------------------------------
# scalar profiles phi(s), s = r² (isotropic) or x·y (dot product)
# ----------------------------------------------------------------------------------------
def _maternp(s, p: int, dtype):
    """stationary.jl:132-158, including the Taylor guard `r² < eps(T)^(1/p)` (:136-146)."""
    s = np.asarray(s, dtype=np.float64)
    c = 2 * p + 1
    r = np.sqrt(c * s)
    coef = [float(v) for v in maternp_coefficients(p)]
    y = np.zeros_like(s)
    ri = np.ones_like(s)
    for i in range(p):
        y = y + coef[i] * ri
        ri = ri * (2 * r)
    y = y + ri
    y = y * (np.exp(-r) / maternp_norm(p))
    if p >= 1:
        bound = _eps(dtype) ** (1.0 / p)
        d = [float(v) for v in maternp_derivatives_at_zero(p)]
        t = np.ones_like(s)
        si = s.copy()
        for i in range(1, p + 1):
            t = t + d[i - 1] * si / math.factorial(i)
            si = si * s
        y = np.where(s < bound, t, y)
    return y


def _matern_taylor(nu: float, dtype):
    """stationary.jl:99-110: taylor_bound and the polynomial the reference returns below it."""
    eps = _eps(dtype)
    bound = math.sqrt(eps) if nu > 2 else (eps if nu > 1 else 0.0)
    t1 = nu / (2 * (1 - nu)) if nu > 1 else 0.0
    t2 = nu ** 2 / (8 * (2 - 3 * nu + nu ** 2)) if nu > 2 else 0.0
    return bound, t1, t2


def _matern(s, nu: float, dtype):
    """stationary.jl:97-114: 2^(1-nu)/gamma(nu) * r^nu besselk(nu, r), r = sqrt(2 nu r²) (adbesselkxv = r^nu K_nu(r), whose
    limit at r = 0 is 2^(nu-1) gamma(nu), i.e. k = 1), with the Taylor polynomial below taylor_bound."""
    from scipy.special import gamma as sgamma, kv
    s = np.asarray(s, dtype=np.float64)
    bound, t1, t2 = _matern_taylor(nu, dtype)
    r = np.sqrt(2 * nu * s)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        y = 2 ** (1 - nu) / sgamma(nu) * r ** nu * kv(nu, r)
    y = np.where(r == 0, 1.0, y)
    return np.where(s < bound, 1 + t1 * s + t2 * s * s, y)


def profile(k, s, dtype=np.float64):
    """phi(s) for every family in scope.  `dtype` only selects eps(T) for the MaternP guard."""
    s = np.asarray(s, dtype=np.float64)
    if isinstance(k, Composite):
        total = np.zeros_like(s)
        for term in k.terms:                         # Sum: algebra.jl:36 (sum(h->h(x,y), S.args))
            prod = np.ones_like(s)
            for f in term:                           # Product: algebra.jl:17 (prod(h->h(x,y), P.args))
                prod = prod * profile(f, s, dtype)
            total = total + prod
        return k.scale * total
    if k.family == CONSTANT:
        return np.full_like(s, k.scale)              # stationary.jl:30-32
    if k.trait == ISOTROPIC and k.lengthscale != 1.0:
        s = s / (k.lengthscale ** 2)                 # transformation.jl:19
    f = k.family
    if f == EQ:
        v = np.exp(-s / 2)                           # stationary.jl:42
    elif f == EXP:
        v = np.exp(-np.sqrt(s))                      # stationary.jl:60
    elif f == RQ:
        v = (1 + s / (2 * k.param)) ** (-k.param)    # stationary.jl:53
    elif f == GAMMAEXP:
        v = np.exp(-(s ** (k.param / 2)) / 2)        # stationary.jl:71
    elif f == CAUCHY:
        v = 1.0 / (1 + s)                            # stationary.jl:224
    elif f == IMQ:
        v = 1.0 / np.sqrt(s + k.param ** 2)          # stationary.jl:235
    elif f == MATERNP:
        v = _maternp(s, k.p, dtype)                  # stationary.jl:132-158
    elif f == MATERN:
        v = _matern(s, k.param, dtype)               # stationary.jl:97-114
    elif f == DOT:
        v = s                                        # mercer.jl:9
    elif f == EXPDOT:
        v = np.exp(s)                                # mercer.jl:22
    elif f == ASINDOT:
        v = 2 / np.pi * np.arcsin(s)                 # mercer.jl:84 on normalised inputs
    else:
        raise ValueError(f"unknown family {f}")
    if k.power != 1:
        v = v ** k.power                             # algebra.jl:61-62
    if k.scale != 1.0:
        v = k.scale * v                              # algebra.jl:17 with Constant (stationary.jl:30-32)
    return v


def _maternp_H(q: int, r):
    h = [float(v) for v in maternp_poly(q)]
    acc = np.zeros_like(r)
    for m in range(q, -1, -1):
        acc = acc * r + h[m]
    return acc * np.exp(-r)


def profile_derivatives(k: Kernel, s, dtype=np.float64):
    """(phi, phi', phi'') w.r.t. s.  The reference obtains phi', phi'' by nested ForwardDiff
    (gradient.jl:584-600); these are the closed forms of the same functions.
    MaternP uses d/dr[r^nu K_nu] = -r^nu K_{nu-1}: phi_p' = d1 * H_{p-1}(r), phi_p'' = d2 * H_{p-2}(r)
    at the same r = sqrt((2p+1)s), and the polynomial (Taylor) branch below eps^(1/p) exactly as
    ForwardDiff differentiates stationary.jl:139-146."""
    s = np.asarray(s, dtype=np.float64)
    if isinstance(k, Composite):
        # what ForwardDiff does to algebra.jl:17,36: linearity over the Sum, Leibniz over each Product
        V = np.zeros_like(s); D1 = np.zeros_like(s); D2 = np.zeros_like(s)
        for term in k.terms:
            p0 = np.ones_like(s); p1 = np.zeros_like(s); p2 = np.zeros_like(s)
            for f in term:
                fv, f1, f2 = profile_derivatives(f, s, dtype)
                p0, p1, p2 = p0 * fv, p1 * fv + p0 * f1, p2 * fv + 2 * p1 * f1 + p0 * f2
            V = V + p0; D1 = D1 + p1; D2 = D2 + p2
        return k.scale * V, k.scale * D1, k.scale * D2
    if k.family == CONSTANT:
        return np.full_like(s, k.scale), np.zeros_like(s), np.zeros_like(s)
    inner = 1.0
    if k.trait == ISOTROPIC and k.lengthscale != 1.0:
        inner = 1.0 / (k.lengthscale ** 2)
        s = s * inner
    f = k.family
    with np.errstate(divide="ignore", invalid="ignore"):
        if f == EQ:
            v = np.exp(-s / 2); d1 = -v / 2; d2 = v / 4
        elif f == EXP:
            rt = np.sqrt(s); v = np.exp(-rt)
            d1 = -v / (2 * rt); d2 = v * (1 / (4 * s) + 1 / (4 * s * rt))
        elif f == RQ:
            a = k.param; u = 1 + s / (2 * a)
            v = u ** (-a); d1 = -0.5 * u ** (-a - 1); d2 = (a + 1) / (4 * a) * u ** (-a - 2)
        elif f == GAMMAEXP:
            g = k.param / 2
            v = np.exp(-(s ** g) / 2)
            d1 = -(g / 2) * s ** (g - 1) * v
            d2 = v * ((g / 2) ** 2 * s ** (2 * g - 2) - (g / 2) * (g - 1) * s ** (g - 2))
        elif f == CAUCHY:
            u = 1 + s; v = 1 / u; d1 = -1 / u ** 2; d2 = 2 / u ** 3
        elif f == IMQ:
            u = s + k.param ** 2
            v = u ** -0.5; d1 = -0.5 * u ** -1.5; d2 = 0.75 * u ** -2.5
        elif f == MATERNP:
            p = k.p
            v = _maternp(s, p, dtype)
            c = 2 * p + 1
            r = np.sqrt(c * s)
            if p == 0:
                d1 = -np.exp(-r) / (2 * r)
                d2 = np.exp(-r) * (1 / (4 * s) + 1 / (4 * s * r))
            else:
                dz = [float(x) for x in maternp_derivatives_at_zero(p)]
                d1 = dz[0] * _maternp_H(p - 1, r)
                if p >= 2:
                    d2 = dz[1] * _maternp_H(p - 2, r)
                else:  # p == 1: phi' = d1 exp(-r) -> phi'' = -d1 exp(-r) c / (2 r)
                    d2 = -dz[0] * np.exp(-r) * c / (2 * r)
                bound = _eps(dtype) ** (1.0 / p)
                t1 = np.zeros_like(s); t2 = np.zeros_like(s)
                for i in range(1, p + 1):
                    ci = dz[i - 1] / math.factorial(i)
                    t1 = t1 + ci * i * s ** (i - 1)
                    if i >= 2:
                        t2 = t2 + ci * i * (i - 1) * s ** (i - 2)
                d1 = np.where(s < bound, t1, d1)
                d2 = np.where(s < bound, t2, d2)
        elif f == MATERN:
            # d/dr[r^a K_a(r)] = -r^a K_{a-1}(r), dr/ds = nu/r: what ForwardDiff computes through adbesselkxv; the polynomial branch
            # differentiated termwise
            from scipy.special import gamma as sgamma, kv
            nu = k.param
            bound, t1, t2 = _matern_taylor(nu, dtype)
            C = 2 ** (1 - nu) / sgamma(nu)
            r = np.sqrt(2 * nu * s)
            v = _matern(s, nu, dtype)
            d1 = -C * nu * r ** (nu - 1) * kv(nu - 1, r)
            d2 = C * nu ** 2 * r ** (nu - 2) * kv(nu - 2, r)
            d1 = np.where(s < bound, t1 + 2 * t2 * s, d1)
            d2 = np.where(s < bound, 2 * t2, d2)
        elif f == DOT:
            v = s; d1 = np.ones_like(s); d2 = np.zeros_like(s)
        elif f == EXPDOT:
            v = np.exp(s); d1 = v; d2 = v
        elif f == ASINDOT:                               # f0, f1, f2 of gradient.jl:192-194
            v = 2 / np.pi * np.arcsin(s); d1 = 2 / np.pi / np.sqrt(1 - s * s); d2 = 2 / np.pi * s / np.sqrt(1 - s * s) ** 3
        else:
            raise ValueError(f"unknown family {f}")
        d1 = d1 * inner
        d2 = d2 * inner * inner
        if k.power != 1:
            q = k.power
            # (phi^q)' = q phi^(q-1) phi' ; (phi^q)'' = q(q-1) phi^(q-2) phi'^2 + q phi^(q-1) phi''
            vq2 = v ** (q - 2) if q >= 2 else np.zeros_like(v)
            d2 = q * (q - 1) * vq2 * d1 * d1 + q * v ** (q - 1) * d2
            d1 = q * v ** (q - 1) * d1
            v = v ** q
        if k.scale != 1.0:
            v = k.scale * v; d1 = k.scale * d1; d2 = k.scale * d2
    return v, d1, d2


# ----------------------------------------------------------------------------------------
# points: (n, d) C-contiguous array == Julia's d×n column-major matrix / vector of d-vectors
# ----------------------------------------------------------------------------------------
def as_points(x):
    x = np.asarray(x)
    if x.ndim == 1:
        x = x[:, None]
    return x


def pair_arg(k: Kernel, X, Y):
    """s[i,j]: euclidean2 by DIRECT differences (util.jl:40-47) or dot(x,y) (mercer.jl:3)."""
    X = as_points(X).astype(np.float64); Y = as_points(Y).astype(np.float64)
    if X.shape[1] != Y.shape[1]:
        raise ValueError("DimensionMismatch: inputs have to have the same length")  # util.jl:41
    if k.trait == ISOTROPIC:
        s = np.zeros((X.shape[0], Y.shape[0]))
        for l in range(X.shape[1]):
            diff = X[:, l][:, None] - Y[:, l][None, :]
            s += diff * diff
        return s
    return X @ Y.T


def matrix(k: Kernel, X, Y=None, dtype=np.float64):
    """Matrix(G) (gramian.jl:102-114): G[i,j] = k(x[i], y[j]) (gramian.jl:37-40)."""
    Y = X if Y is None else Y
    return profile(k, pair_arg(k, X, Y), dtype)


def mul(y, k: Kernel, X, Y, a, alpha=1.0, beta=0.0, dtype=np.float64, chunk=2048):
    """mul!(y, G, a, α, β) for vector or matrix right-hand sides (gramian.jl:78-99).
    β == 0 ⇒ previous contents of y (NaN included) are discarded (gramian.jl:80,90)."""
    X = as_points(X); Y = as_points(X if Y is None else Y)
    a = np.asarray(a, dtype=np.float64)
    n = X.shape[0]
    out = np.zeros((n,) + a.shape[1:], dtype=np.float64)
    if beta != 0:
        out += beta * np.asarray(y, dtype=np.float64)
    for i0 in range(0, n, chunk):
        G = matrix(k, X[i0:i0 + chunk], Y, dtype)
        out[i0:i0 + chunk] += alpha * (G @ a)
    return out


# ----------------------------------------------------------------------------------------
# GradientKernel (gradient.jl:7-24) block MVM
# ----------------------------------------------------------------------------------------
def grad_block(k: Kernel, x, y, dtype=np.float64):
    """Dense d×d block ∂x∂y' k(x,y) = K * I (gradient.jl:58: Matrix(K) = K * I(d))."""
    x = np.asarray(x, dtype=np.float64); y = np.asarray(y, dtype=np.float64)
    d = x.shape[0]
    if k.trait == ISOTROPIC:
        r = x - y
        _, k1, k2 = profile_derivatives(k, float(r @ r), dtype)
        return -2 * (k1 * np.eye(d) + 2 * k2 * np.outer(r, r))     # gradient.jl:86-92
    _, k1, k2 = profile_derivatives(k, float(x @ y), dtype)
    return k1 * np.eye(d) + k2 * np.outer(y, x)                     # gradient.jl:109-115


def grad_mul(yv, k: Kernel, X, Y, a, alpha=1.0, beta=0.0, dtype=np.float64, chunk=256):
    """blockmul!(y, G, x, α, β) (gramian.jl:241-253) with the per-block mul! of
    gradient.jl:86-92 (IsotropicInput) / :109-115 (DotProductInput).
    Flat vectors are point-major: block i = entries i*d .. (i+1)*d-1 (BlockFactorizations strided)."""
    X = as_points(X).astype(np.float64); Y = as_points(X if Y is None else Y).astype(np.float64)
    n, d = X.shape; m = Y.shape[0]
    A = np.asarray(a, dtype=np.float64).reshape(m, d)
    out = np.zeros((n, d))
    if beta != 0:
        out += beta * np.asarray(yv, dtype=np.float64).reshape(n, d)
    for i0 in range(0, n, chunk):
        Xi = X[i0:i0 + chunk]
        if k.trait == ISOTROPIC:
            R = Xi[:, None, :] - Y[None, :, :]                     # r = difference(x, y)
            s = np.einsum("ijl,ijl->ij", R, R)                     # r² = sum(abs2, r)
            _, k1, k2 = profile_derivatives(k, s, dtype)           # derivative_laplacian
            ra = np.einsum("ijl,jl->ij", R, A)                     # dot_r_a = r'a
            blk = -2 * (k1[:, :, None] * A[None] + 2 * (k2 * ra)[:, :, None] * R)
            if blk.size and not np.all(np.isfinite(blk)):
                pass  # non-differentiable profiles at s = 0 propagate NaN/Inf like the reference
            out[i0:i0 + chunk] += alpha * blk.sum(axis=1)
        else:
            s = Xi @ Y.T                                           # d² = dot(x, y)
            _, k1, k2 = profile_derivatives(k, s, dtype)
            xa = Xi @ A.T                                          # dot_x_a = x'a
            out[i0:i0 + chunk] += alpha * (k1 @ A + (k2 * xa) @ Y)
    return out.reshape(-1)


def grad_matrix(k: Kernel, X, Y=None, dtype=np.float64):
    X = as_points(X); Y = as_points(X if Y is None else Y)
    n, d = X.shape; m = Y.shape[0]
    M = np.zeros((n * d, m * d))
    for i in range(n):
        for j in range(m):
            M[i * d:(i + 1) * d, j * d:(j + 1) * d] = grad_block(k, X[i], Y[j], dtype)
    return M


# ----------------------------------------------------------------------------------------
# ValueGradientKernel (gradient.jl:400-474): blocks of d+1, value component first
# ----------------------------------------------------------------------------------------
def valgrad_block(k, x, y, dtype=np.float64):
    """Dense (d+1)×(d+1) block, laid out as Matrix(::DerivativeKernelElement) does (gradient.jl:353-375):
    [value_value, value_gradient'; gradient_value, gradient_gradient] with the closed forms of
    value_gradient_kernel! (gradient.jl:441-463)."""
    x = np.asarray(x, dtype=np.float64); y = np.asarray(y, dtype=np.float64)
    d = x.shape[0]
    M = np.zeros((d + 1, d + 1))
    if k.trait == ISOTROPIC:
        r = x - y
        k0, k1, _ = profile_derivatives(k, float(r @ r), dtype)
        M[0, 0] = k0
        M[0, 1:] = -2 * k1 * r                                   # value_gradient (gradient.jl:459-460)
        M[1:, 0] = 2 * k1 * r                                    # gradient_value (gradient.jl:461)
    else:
        k0, k1, _ = profile_derivatives(k, float(x @ y), dtype)
        M[0, 0] = k0
        M[0, 1:] = k1 * x                                        # gradient.jl:446
        M[1:, 0] = k1 * y                                        # gradient.jl:447
    M[1:, 1:] = grad_block(k, x, y, dtype)
    return M


def valgrad_matrix(k, X, Y=None, dtype=np.float64):
    X = as_points(X); Y = as_points(X if Y is None else Y)
    n, d = X.shape; m = Y.shape[0]
    b = d + 1
    M = np.zeros((n * b, m * b))
    for i in range(n):
        for j in range(m):
            M[i * b:(i + 1) * b, j * b:(j + 1) * b] = valgrad_block(k, X[i], Y[j], dtype)
    return M


def valgrad_mul(yv, k, X, Y, a, alpha=1.0, beta=0.0, dtype=np.float64, chunk=256):
    """blockmul! (gramian.jl:241-253) with the DerivativeKernelElement mul! (gradient.jl:319-351):
    y[1] += α (vv a[1] + value_gradient·a_g);  y_g += α (gradient_value a[1] + gradient_gradient a_g)."""
    X = as_points(X).astype(np.float64); Y = as_points(X if Y is None else Y).astype(np.float64)
    n, d = X.shape; m = Y.shape[0]
    A = np.asarray(a, dtype=np.float64).reshape(m, d + 1)
    a0, Ag = A[:, 0], A[:, 1:]
    out = np.zeros((n, d + 1))
    if beta != 0:
        out += beta * np.asarray(yv, dtype=np.float64).reshape(n, d + 1)
    for i0 in range(0, n, chunk):
        Xi = X[i0:i0 + chunk]
        if k.trait == ISOTROPIC:
            R = Xi[:, None, :] - Y[None, :, :]
            s = np.einsum("ijl,ijl->ij", R, R)
            k0, k1, k2 = profile_derivatives(k, s, dtype)
            ra = np.einsum("ijl,jl->ij", R, Ag)
            val = k0 @ a0 + (-2 * k1 * ra).sum(axis=1)
            blk = (2 * k1 * a0[None, :])[:, :, None] * R - 2 * (k1[:, :, None] * Ag[None] + 2 * (k2 * ra)[:, :, None] * R)
            out[i0:i0 + chunk, 0] += alpha * val
            out[i0:i0 + chunk, 1:] += alpha * blk.sum(axis=1)
        else:
            s = Xi @ Y.T
            k0, k1, k2 = profile_derivatives(k, s, dtype)
            xa = Xi @ Ag.T
            out[i0:i0 + chunk, 0] += alpha * (k0 @ a0 + (k1 * xa).sum(axis=1))
            out[i0:i0 + chunk, 1:] += alpha * ((k1 * a0[None, :]) @ Y + k1 @ Ag + (k2 * xa) @ Y)
    return out.reshape(-1)


# ----------------------------------------------------------------------------------------
# Toeplitz (gramian.jl:167-189; MVM is ToeplitzMatrices 0.7.1 — restated from its published
# algorithm: circulant embedding + FFT)
# ----------------------------------------------------------------------------------------
def srange(start, stop, length):
    """Julia's range(start, stop, length) as (start, step, length)."""
    step = (stop - start) / (length - 1) if length > 1 else 0.0
    return (float(start), float(step), int(length))


def srange_points(rg):
    x0, h, n = rg
    return x0 + h * np.arange(n, dtype=np.float64)


def toeplitz_vectors(k: Kernel, xr, yr=None, periodic=False, dtype=np.float64):
    """vc (first column) and vr (first row) as gramian.jl:172-189 forms them:
    x === y: vc = k.(x[1], x) → SymmetricToeplitz; same step: vc = k.(x, y[1]), vr = k.(x[1], y)."""
    x = srange_points(xr)
    if yr is None or yr == xr:
        vc = profile(k, (x[0] - x) ** 2 if k.trait == ISOTROPIC else x[0] * x, dtype)
        return vc, None
    y = srange_points(yr)
    if abs(xr[1] - yr[1]) > 0:
        raise ValueError("different step: reference falls back to a plain Gramian (gramian.jl:180-182)")
    vc = profile(k, (x - y[0]) ** 2, dtype)
    vr = profile(k, (x[0] - y) ** 2, dtype)
    return vc, vr


def toeplitz_dense(vc, vr=None, circulant=False):
    vc = np.asarray(vc, dtype=np.float64)
    n = vc.shape[0]
    if circulant:
        idx = (np.arange(n)[:, None] - np.arange(n)[None, :]) % n
        return vc[idx]
    vr = vc if vr is None else np.asarray(vr, dtype=np.float64)
    m = vr.shape[0]
    i = np.arange(n)[:, None]; j = np.arange(m)[None, :]
    return np.where(i >= j, vc[np.clip(i - j, 0, n - 1)], vr[np.clip(j - i, 0, m - 1)])


def toeplitz_mul(y, vc, vr, a, alpha=1.0, beta=0.0, circulant=False):
    """y ← α T a + β y by circulant embedding of size N ≥ n+m-1 (any such N gives the exact product)."""
    vc = np.asarray(vc, dtype=np.float64); a = np.asarray(a, dtype=np.float64)
    n = vc.shape[0]
    if circulant:
        t = np.fft.irfft(np.fft.rfft(vc) * np.fft.rfft(a), n)
    else:
        vr_ = vc if vr is None else np.asarray(vr, dtype=np.float64)
        m = vr_.shape[0]
        N = 1 << int(math.ceil(math.log2(max(n + m - 1, 2))))
        c = np.zeros(N)
        c[:n] = vc
        if m > 1:
            c[N - (m - 1):] = vr_[1:][::-1]
        ap = np.zeros(N); ap[:m] = a
        t = np.fft.irfft(np.fft.rfft(c) * np.fft.rfft(ap), N)[:n]
    out = alpha * t
    if beta != 0:
        out = out + beta * np.asarray(y, dtype=np.float64)
    return out


# ----------------------------------------------------------------------------------------
# Kronecker (algebra.jl:91-95, separable.jl:33-35, lazy_grid.jl:20-38; MVM is KroneckerProducts 1.1.1,
# restated from the identity (A ⊗ B) vec(V) = vec(B V Aᵀ) with column-major vec)
# ----------------------------------------------------------------------------------------
def lazy_grid_points(axes, rowmajor=False):
    """LazyGrid enumeration: default (lazy_grid.jl:20-38) the FIRST axis varies fastest;
    rowmajor=True (lazy_grid.jl:40-58) the LAST axis varies fastest."""
    axes = [np.asarray(a, dtype=np.float64) for a in axes]
    mesh = np.meshgrid(*axes, indexing="ij")
    order = "C" if rowmajor else "F"
    return np.stack([g.reshape(-1, order=order) for g in mesh], axis=1)


def kron_dense(factors):
    """Matrix(kronecker(G1, ..., Gq)) = G1 ⊗ ... ⊗ Gq, the standard Kronecker product (first factor =
    slowest index), which is what algebra.jl:94 and separable.jl:34 build.
    NOTE (reference quirk, kept): LazyGrid's default enumeration (lazy_grid.jl:20-38) makes the FIRST
    axis the fastest index, so for non-identical factors kronecker(G1..Gq) equals the dense Gramian on
    the ROW-major enumeration (lazy_grid.jl:40-58), not the default one.  The reference's only test
    (test/algebra.jl:70-89) uses identical factors, where both orders coincide."""
    out = np.asarray(factors[0], dtype=np.float64)
    for f in factors[1:]:
        out = np.kron(out, np.asarray(f, dtype=np.float64))
    return out


def kron_mul(y, factors, a, alpha=1.0, beta=0.0):
    """(F1 ⊗ F2 ⊗ ... ⊗ Fq) a by successive mode products; standard Kronecker order
    (F1 index slowest).  a has length prod(cols)."""
    fs = [np.asarray(f, dtype=np.float64) for f in factors]
    cols = [f.shape[1] for f in fs]
    t = np.asarray(a, dtype=np.float64).reshape(cols)        # C-order: first factor slowest
    for ax, f in enumerate(fs):
        t = np.moveaxis(np.tensordot(f, t, axes=([1], [ax])), 0, ax)
    out = alpha * t.reshape(-1)
    if beta != 0:
        out = out + beta * np.asarray(y, dtype=np.float64)
    return out


# ----------------------------------------------------------------------------------------
# Low rank (mercer.jl:53-70, lazy_linear_algebra.jl:78-85)
# ----------------------------------------------------------------------------------------
def lowrank_mul(y, U, V, a, alpha=1.0, beta=0.0):
    """LazyMatrixProduct(U, V') * a = U (V' a), right-to-left (lazy_linear_algebra.jl:78-85)."""
    z = np.asarray(V, dtype=np.float64).T @ np.asarray(a, dtype=np.float64)
    z = np.asarray(U, dtype=np.float64) @ z
    out = alpha * z
    if beta != 0:
        out = out + beta * np.asarray(y, dtype=np.float64)
    return out


# ----------------------------------------------------------------------------------------
# Toeplitz direct solvers (src/toeplitz.jl:12-145) — SURVEY §8(f)-4.  Sequential O(n^2) recurrences; restated so that
# the device's PCG-over-the-FFT-MVM solver (covgram.solve.toeplitz_solve) has a reference-faithful checker.
# ----------------------------------------------------------------------------------------
def durbin(r):
    """toeplitz.jl:14-27: y = K \\ (-r), K = SymmetricToeplitz([1, r[1:end-1]]) (Golub & Van Loan alg. 4.7.1)."""
    r = np.asarray(r, dtype=np.float64)
    n = r.shape[0]
    y = np.zeros(n)
    y[0] = -r[0]
    alpha, beta = -r[0], 1.0
    for k in range(1, n):
        beta *= (1 - alpha * alpha)
        alpha = -(r[k] + np.dot(r[:k], y[:k][::-1])) / beta          # reverse_dot (toeplitz.jl:114-122)
        y[:k] = y[:k] + alpha * y[:k][::-1]                          # reverse_increment! (toeplitz.jl:125-145)
        y[k] = alpha
    return y


def levinson(r, b):
    """toeplitz.jl:77-98: x = K \\ b, K = SymmetricToeplitz([1; r]) positive definite."""
    r = np.asarray(r, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    n = r.shape[0] + 1
    x = np.zeros(n); y = np.zeros(n)
    y[0] = -r[0]; x[0] = b[0]
    alpha, beta = -r[0], 1.0
    for k in range(1, n):
        beta *= (1 - alpha * alpha)
        mu = (b[k] - np.dot(r[:k], x[:k][::-1])) / beta
        x[:k] = x[:k] + mu * y[:k][::-1]
        x[k] = mu
        if k < n - 1:
            alpha = -(r[k] + np.dot(r[:k], y[:k][::-1])) / beta
            y[:k] = y[:k] + alpha * y[:k][::-1]
            y[k] = alpha
    return x


def levinson_toeplitz(vc, b):
    """toeplitz.jl:100-111 with the diagonal normalisation done right (the reference tests `r_0 == 1` where it means
    `r_0 != 1`): T = r_0 * SymmetricToeplitz([1; vc[1:]/r_0])  =>  x = levinson(vc[1:]/r_0, b) / r_0."""
    vc = np.asarray(vc, dtype=np.float64)
    r0 = vc[0]
    return levinson(vc[1:] / r0, b) / r0


def trench(r):
    """toeplitz.jl:57-71: inverse of K = SymmetricToeplitz([1; r]) (Golub & Van Loan alg. 4.7.3), full symmetric matrix."""
    r = np.asarray(r, dtype=np.float64)
    n = r.shape[0] + 1
    y = durbin(r)
    gamma = 1.0 / (1 + np.dot(r, y))
    nu = gamma * y[::-1]
    B = np.zeros((n, n))
    B[0, 0] = gamma
    B[0, 1:] = gamma * y
    for j in range(1, n):
        for i in range(1, j + 1):
            B[i, j] = B[i - 1, j - 1] + (nu[n - 1 - j] * nu[n - 1 - i] - nu[i - 1] * nu[j - 1]) / gamma
    return np.triu(B) + np.triu(B, 1).T


# ----------------------------------------------------------------------------------------
# Pivoted Cholesky (src/gramian.jl:192-213: cholesky(G, Val(true); tol) = LAPACK pstrf on Matrix(G)) — SURVEY §8(f)-3
# ----------------------------------------------------------------------------------------
def pivoted_cholesky(A, tol=0.0, max_rank=None):
    """P' A P = L L' by diagonal pivoting, stopping when the largest remaining diagonal is <= tol (LAPACK dpstrf's rule for
    a user tolerance >= 0).  Returns (L in ORIGINAL row order, n x rank; piv; rank)."""
    A = np.array(A, dtype=np.float64)
    n = A.shape[0]
    max_rank = n if max_rank is None else min(max_rank, n)
    d = np.diag(A).copy()
    piv = np.arange(n)
    L = np.zeros((n, max_rank))
    rank = 0
    for k in range(max_rank):
        j = k + int(np.argmax(d[piv[k:]]))
        if d[piv[j]] <= tol:
            break
        piv[[k, j]] = piv[[j, k]]
        p = piv[k]
        col = A[:, p] - L[:, :k] @ L[p, :k]
        L[:, k] = col / np.sqrt(d[p])
        d = d - L[:, k] ** 2
        d[piv[:k + 1]] = 0.0
        rank = k + 1
    return L[:, :rank], piv, rank


# ----------------------------------------------------------------------------------------
# Input / output transformations (src/transformation.jl) and the Cosine kernel (src/stationary.jl:197-211), by definition
# ----------------------------------------------------------------------------------------
def scaled_input_points(U, X):
    """(S::ScaledInputKernel)(x, y) = S.k(S.U*x, S.U*y) (transformation.jl:79); ARD(k, l) is U = Diagonal(1 ./ l) (:42-45).
    U: (d', d) matrix or a length-d diagonal."""
    U = np.asarray(U, dtype=np.float64); X = as_points(X).astype(np.float64)
    return X * U if U.ndim == 1 else X @ U.T


def periodic_matrix(k: Kernel, x, y=None, dtype=np.float64):
    """Periodic(k)(τ) = k((2 sin(π τ))²) for one-dimensional inputs (transformation.jl:61-65)."""
    x = np.asarray(x, dtype=np.float64).reshape(-1); y = x if y is None else np.asarray(y, dtype=np.float64).reshape(-1)
    tau = x[:, None] - y[None, :]
    return profile(k, (2 * np.sin(np.pi * tau)) ** 2, dtype)


def cosine_matrix(c, X, Y=None):
    """Cosine(c)(x, y) = cos(2π c·(x − y)) (stationary.jl:207-211)."""
    X = as_points(X).astype(np.float64); Y = X if Y is None else as_points(Y).astype(np.float64)
    c = np.asarray(c, dtype=np.float64).reshape(-1)
    return np.cos(2 * np.pi * ((X @ c)[:, None] - (Y @ c)[None, :]))


def cosine_grad_matrix(c, X, Y=None):
    """Gradient Gramian of Cosine: block = −k₂ c c' (gradient.jl:129-136), k₂ = d²/dz² cos(2π z) = −4π² cos(2π c·r)."""
    X = as_points(X).astype(np.float64); Y = X if Y is None else as_points(Y).astype(np.float64)
    c = np.asarray(c, dtype=np.float64).reshape(-1)
    return np.kron(4 * np.pi ** 2 * cosine_matrix(c, X, Y), np.outer(c, c))


def linear_map_grad_matrix(k: Kernel, U, X, Y=None, dtype=np.float64):
    """∂x ∂y' k(Ux, Uy) = U' B(Ux, Uy) U (chain rule through transformation.jl:79): dense (n d) × (m d) matrix."""
    U = np.asarray(U, dtype=np.float64)
    Um = np.diag(U) if U.ndim == 1 else U
    X = as_points(X).astype(np.float64); Y = X if Y is None else as_points(Y).astype(np.float64)
    TX, TY = scaled_input_points(U, X), scaled_input_points(U, Y)
    n, m, d, dp = X.shape[0], Y.shape[0], X.shape[1], Um.shape[0]
    inner = grad_matrix(k, TX, TY, dtype)
    M = np.zeros((n * d, m * d))
    for i in range(n):
        for j in range(m):
            M[i * d:(i + 1) * d, j * d:(j + 1) * d] = Um.T @ inner[i * dp:(i + 1) * dp, j * dp:(j + 1) * dp] @ Um
    return M


# ----------------------------------------------------------------------------------------
# NeuralNetwork kernel (src/mercer.jl:73-85) and its gradient block (src/gradient.jl:187-210), restated literally
# ----------------------------------------------------------------------------------------
def nn_matrix(sigma, X, Y=None):
    """k(x, y) = 2/π asin(l(x,y) / sqrt((1 + l(x,x)) (1 + l(y,y)))), l(x, y) = x·y + σ."""
    X = as_points(X).astype(np.float64); Y = X if Y is None else as_points(Y).astype(np.float64)
    lxy = X @ Y.T + sigma
    lxx = (X * X).sum(1) + sigma; lyy = (Y * Y).sum(1) + sigma
    return 2 / np.pi * np.arcsin(lxy / np.sqrt((1 + lxx)[:, None] * (1 + lyy)[None, :]))


def nn_grad_block(x, y):
    """∂x ∂y' of the NN kernel with σ = 0 as the reference assembles it: k1 I + [x y] C [x y]' (gradient.jl:187-210)."""
    x = np.asarray(x, dtype=np.float64); y = np.asarray(y, dtype=np.float64)
    Nx, Ny = x @ x + 1, y @ y + 1
    dxy, Nxy = x @ y, Nx * Ny
    dd = dxy / np.sqrt(Nxy)
    k1 = 2 / np.pi / np.sqrt(1 - dd * dd) / np.sqrt(Nxy)
    k2 = 2 / np.pi / np.sqrt(1 - dd * dd) ** 3 * dd / Nxy
    k12 = k1 + k2 * dxy
    C = np.array([[-k12 / Nx, k12 * dxy / Nxy], [k2, -k12 / Ny]])
    U = np.stack([x, y], axis=1)
    return k1 * np.eye(x.shape[0]) + U @ C @ U.T


def nn_grad_matrix(X, Y=None):
    X = as_points(X).astype(np.float64); Y = X if Y is None else as_points(Y).astype(np.float64)
    n, d = X.shape; m = Y.shape[0]
    M = np.zeros((n * d, m * d))
    for i in range(n):
        for j in range(m):
            M[i * d:(i + 1) * d, j * d:(j + 1) * d] = nn_grad_block(X[i], Y[j])
    return M
