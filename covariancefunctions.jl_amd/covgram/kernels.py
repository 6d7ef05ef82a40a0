"""Kernel structs and `input_trait` — the host-side mirror of the reference's kernel zoo for the hot path.

Same names, argument meaning and error behaviour as CovarianceFunctions.jl (citations relative to the
reference root).  A kernel object is a *description*; `device_spec(k)` lowers it to the C struct
`covgram_kernel` that selects the HIP kernel, exactly the role `input_trait(k)` plays in the
reference (src/properties.jl:31-63, README.md:90-99).  Calling a kernel, `k(x, y)`, evaluates the
scalar definition on the host for single pairs (API parity with the Julia call operators); the MVM,
`Matrix(G)` and Toeplitz/Kronecker constructions never use it — they run on the device.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import numpy as np

from . import _ffi

# ----------------------------------------------------------------------------------------------
# input traits (src/properties.jl:31-37)
# ----------------------------------------------------------------------------------------------


class InputTrait:
    def __eq__(self, other):
        return type(self) is type(other)

    def __hash__(self):
        return hash(type(self).__name__)

    def __repr__(self):
        return type(self).__name__ + "()"


class GenericInput(InputTrait): pass
class IsotropicInput(InputTrait): pass            # dependent on |r|²
class DotProductInput(InputTrait): pass           # dependent on x ⋅ y
class StationaryInput(InputTrait): pass           # dependent on r
class StationaryLinearFunctionalInput(InputTrait): pass   # dependent on c ⋅ r
class PeriodicInput(InputTrait): pass


class DomainError(ValueError):
    pass


def _dot(x, y):
    return float(np.dot(np.atleast_1d(np.asarray(x, dtype=np.float64)), np.atleast_1d(np.asarray(y, dtype=np.float64))))


def _euclidean2(x, y):
    """src/util.jl:40-47 — direct differences, DimensionMismatch on unequal lengths."""
    x = np.atleast_1d(np.asarray(x, dtype=np.float64)); y = np.atleast_1d(np.asarray(y, dtype=np.float64))
    if x.shape != y.shape:
        raise _ffi.DimensionMismatch(_ffi.EINVAL, f"inputs have to have the same length: {x.size}, {y.size}")
    d = x - y
    return float(np.dot(d, d))


# ----------------------------------------------------------------------------------------------
# type tree (src/CovarianceFunctions.jl:32-40)
# ----------------------------------------------------------------------------------------------
class AbstractKernel:
    def __mul__(self, other):
        if isinstance(other, (int, float)):
            return Product((Constant(other), self))          # src/algebra.jl:24-25
        if isinstance(other, AbstractKernel):
            return Product((self, other))
        return NotImplemented

    __rmul__ = __mul__

    def __add__(self, other):
        if isinstance(other, (int, float)):
            return Sum((self, Constant(other)))              # src/algebra.jl:46-47
        if isinstance(other, AbstractKernel):
            return Sum((self, other))
        return NotImplemented

    __radd__ = __add__

    def __pow__(self, p):
        return Power(self, p)                                # src/algebra.jl:63


class MercerKernel(AbstractKernel): pass
class StationaryKernel(MercerKernel): pass


class IsotropicKernel(StationaryKernel):
    """(k::IsotropicKernel)(x, y) = k(euclidean2(x, y)); one-argument form takes r² (src/stationary.jl:9-10)."""

    def profile(self, s: float) -> float:
        raise NotImplementedError

    def __call__(self, *args):
        if len(args) == 2:
            return self.profile(_euclidean2(args[0], args[1]))
        (a,) = args
        if np.ndim(a) == 0:
            return self.profile(float(a))                    # k(r²)
        a = np.asarray(a, dtype=np.float64)
        return self.profile(float(np.dot(a, a)))             # k(τ) = k(sum(abs2, τ))


class MultiKernel(AbstractKernel): pass


class Constant(IsotropicKernel):
    """src/stationary.jl:15-34."""

    def __init__(self, c, check: bool = True):
        if check and not (c >= 0):
            raise DomainError(f"Constant is not positive semi-definite: {c}")
        self.c = float(c)

    def profile(self, s):
        return self.c

    def __call__(self, *args):
        return self.c


class ExponentiatedQuadratic(IsotropicKernel):
    def profile(self, s):
        return math.exp(-s / 2)                              # src/stationary.jl:42


EQ = ExponentiatedQuadratic


class RationalQuadratic(IsotropicKernel):
    def __init__(self, alpha):
        if not alpha > 0:
            raise DomainError("α not positive")             # src/stationary.jl:47
        self.alpha = float(alpha)

    def profile(self, s):
        return (1 + s / (2 * self.alpha)) ** (-self.alpha)   # src/stationary.jl:53


RQ = RationalQuadratic


class Exponential(IsotropicKernel):
    def profile(self, s):
        return math.exp(-math.sqrt(s))                       # src/stationary.jl:60


Exp = Exponential


class GammaExponential(IsotropicKernel):
    def __init__(self, gamma):
        if not (0 <= gamma <= 2):
            raise DomainError("γ not in [0,2]")             # src/stationary.jl:65
        self.gamma = float(gamma)

    def profile(self, s):
        return math.exp(-(s ** (self.gamma / 2)) / 2)        # src/stationary.jl:71


GammaExp = GammaExponential


class Cauchy(IsotropicKernel):
    def profile(self, s):
        return 1.0 / (1.0 + s)                               # src/stationary.jl:224


class InverseMultiQuadratic(IsotropicKernel):
    def __init__(self, c):
        self.c = float(c)

    def profile(self, s):
        return 1.0 / math.sqrt(s + self.c ** 2)              # src/stationary.jl:235


def maternp_coefficients(p: int):
    """src/stationary.jl:184-191 (reversed binomial(p,i) * (p+i)!/p!)."""
    fp = math.factorial(p)
    return [math.comb(p, i) * (math.factorial(p + i) // fp) for i in range(1, p + 1)][::-1]


def maternp_derivatives_at_zero(p: int):
    """src/stationary.jl:172-182 — SymEngine there; here from the exact power series of exp(-r) q_p(r)."""
    from fractions import Fraction
    c = [Fraction(v) for v in maternp_coefficients(p)] + [Fraction(1)]
    nrm = math.factorial(2 * p) // math.factorial(p)
    h = [c[m] * 2 ** m / nrm for m in range(p + 1)]
    out = []
    for i in range(1, p + 1):
        coef = sum(h[m] * Fraction((-1) ** (2 * i - m), math.factorial(2 * i - m)) for m in range(0, min(p, 2 * i) + 1))
        out.append(float(coef * (2 * p + 1) ** i * math.factorial(i)))
    return out


class MaternP(IsotropicKernel):
    """Matern with ν = p + 1/2 (src/stationary.jl:117-158), including the Taylor guard near zero."""

    def __init__(self, p):
        if isinstance(p, Matern):
            p = int(math.floor(p.nu))                        # src/stationary.jl:130
        if p < 0:
            raise DomainError(f"p = {p} is negative")        # src/stationary.jl:124
        self.p = int(p)
        self.coefficients = [float(v) for v in maternp_coefficients(self.p)]
        self.derivatives = maternp_derivatives_at_zero(self.p)

    def profile(self, s, eps=np.finfo(np.float64).eps):
        p = self.p
        if p >= 1 and s < eps ** (1.0 / p):                  # src/stationary.jl:135-146
            y, si = 1.0, s
            for i in range(1, p + 1):
                y += self.derivatives[i - 1] * si / math.factorial(i)
                si *= s
            return y
        r = math.sqrt((2 * p + 1) * s)
        y, ri = 0.0, 1.0
        for i in range(p):
            y += self.coefficients[i] * ri
            ri *= 2 * r
        y += ri
        return y * math.exp(-r) / (math.factorial(2 * p) // math.factorial(p))


class Matern(IsotropicKernel):
    """src/stationary.jl:87-114 (Bessel form, real ν > 0); on the device Temme's series / Steed's continued fraction."""

    def __init__(self, nu):
        if not nu > 0:
            raise DomainError(f"ν = {nu} is negative")
        self.nu = float(nu)

    def profile(self, s):
        from scipy.special import gamma, kv
        nu = self.nu
        if s == 0:
            return 1.0
        r = math.sqrt(2 * nu * s)
        return 2 ** (1 - nu) / gamma(nu) * r ** nu * kv(nu, r)


class DotProductKernel(MercerKernel):
    """(k::DotProductKernel)(x, y) = k(dot(x, y)) (src/mercer.jl:2-3)."""

    def profile(self, s):
        raise NotImplementedError

    def __call__(self, *args):
        if len(args) == 2:
            return self.profile(_dot(args[0], args[1]))
        return self.profile(float(args[0]))


class Dot(DotProductKernel):
    def profile(self, s):
        return s                                             # src/mercer.jl:9


class ExponentialDot(DotProductKernel):
    def profile(self, s):
        return math.exp(s)                                   # src/mercer.jl:22


def Line(sigma: float = 0.0):
    """Line(σ) = Dot() + σ (src/mercer.jl:12)."""
    return Dot() + sigma


def Polynomial(d: int, sigma: float = 0.0):
    """Polynomial(d, σ) = Line(σ)^d (src/mercer.jl:13-14); lowers to a composite of Dot powers on the device."""
    return Line(sigma) ** int(d)


Poly = Polynomial


class CosineKernel(StationaryKernel):
    """k(x, y) = cos(2π c·(x − y)) (src/stationary.jl:197-211), input_trait = StationaryLinearFunctionalInput.
    cos(u_i − v_j) = cos u_i cos v_j + sin u_i sin v_j: its Gramian has rank 2 (the reference's own "IDEA: trig-identity ->
    low-rank gramian", :204) and is represented as such here."""

    def __init__(self, c):
        self.c = np.atleast_1d(np.asarray(c, dtype=np.float64))

    def __call__(self, *args):
        if len(args) == 2:
            x = np.atleast_1d(np.asarray(args[0], dtype=np.float64)); y = np.atleast_1d(np.asarray(args[1], dtype=np.float64))
            return math.cos(2 * math.pi * float(np.dot(self.c, x - y)))
        return math.cos(2 * math.pi * float(args[0]))


Cosine = Cos = CosineKernel


class Periodic(StationaryKernel):
    """Periodic(k)(τ) = k((2 sin(πτ))²) for 1-D inputs (src/transformation.jl:54-65): the isotropic kernel k on the circle
    embedding e(x) = (cos 2πx, sin 2πx), since |e(x) − e(y)|² = 4 sin²(π(x − y))."""

    def __init__(self, k):
        if not isinstance(k, IsotropicKernel):
            raise TypeError("Periodic(k::IsotropicKernel)")
        self.k = k

    def __call__(self, *args):
        tau = float(args[0]) - float(args[1]) if len(args) == 2 else float(args[0])
        return self.k.profile((2 * math.sin(math.pi * tau)) ** 2)


class ScaledInputKernel(AbstractKernel):
    """ScaledInputKernel(k, U)(x, y) = k(U x, U y) (src/transformation.jl:71-79); `gramian` pre-multiplies the points once
    (:82-90).  ARD(k, l) (:42-46) is the diagonal case U = Diagonal(1 ./ l)."""

    def __init__(self, k, U):
        self.k = k
        self.U = np.asarray(U, dtype=np.float64)
        self.diagonal = self.U.ndim == 1

    def __call__(self, x, y):
        x = np.atleast_1d(np.asarray(x, dtype=np.float64)); y = np.atleast_1d(np.asarray(y, dtype=np.float64))
        return self.k(self.U * x, self.U * y) if self.diagonal else self.k(self.U @ x, self.U @ y)


def ARD(k, l):
    """Automatic relevance determination (src/transformation.jl:42-46): k on inputs scaled by 1 ./ l; scalar l = Lengthscale."""
    if np.ndim(l) == 0:
        return Lengthscale(k, l)
    l = np.asarray(l, dtype=np.float64)
    if not np.all(l > 0):
        raise DomainError(f"l = {l} is non-positive")
    return ScaledInputKernel(k, 1.0 / l)


class Warped(AbstractKernel):
    """Warped(k, u)(x, y) = k(u(x), u(y)) (src/transformation.jl:98-110); u is a matrix or a callable that maps an (n, d) tensor of
    points to an (n, d') tensor (vectorised over points); `gramian` warps the points once (:114-118)."""

    def __init__(self, k, u):
        self.k = k
        self.u = u

    def __call__(self, x, y):
        u = (lambda z: np.asarray(self.u) @ z) if not callable(self.u) else (lambda z: np.asarray(self.u(np.atleast_2d(z)))[0])
        return self.k(u(np.atleast_1d(np.asarray(x, dtype=np.float64))), u(np.atleast_1d(np.asarray(y, dtype=np.float64))))


class VerticalRescaling(AbstractKernel):
    """VerticalRescaling(k, f)(x, y) = f(x) k(x, y) f(y) (src/transformation.jl:156-171): Diagonal · gramian(k) · Diagonal; f is
    vectorised over an (n, d) tensor of points and returns n values."""

    def __init__(self, k, f):
        self.k, self.f = k, f

    def __call__(self, x, y):
        fx = float(np.asarray(self.f(np.atleast_2d(np.asarray(x, dtype=np.float64)))).reshape(-1)[0])
        fy = float(np.asarray(self.f(np.atleast_2d(np.asarray(y, dtype=np.float64)))).reshape(-1)[0])
        return fx * self.k(x, y) * fy


class AsinDot(DotProductKernel):
    """φ(s) = (2/π) asin(s): what the NeuralNetwork kernel is on its normalised augmented inputs (device family ASINDOT)."""

    def profile(self, s):
        return 2 / math.pi * math.asin(s)


class NeuralNetwork(MercerKernel):
    """NN(σ)(x, y) = 2/π asin(l(x,y) / sqrt((1 + l(x,x)) (1 + l(y,y)))), l = Line(σ) = x·y + σ (src/mercer.jl:73-85).
    With x̂ = [x, √σ] / sqrt(1 + |x|² + σ) this is AsinDot on (x̂, ŷ): `gramian` warps the points once and the dot-product
    hot path does the rest; the gradient Gramian follows by the chain rule through the warp (per-point Jacobians, O(d) each),
    which for σ = 0 is exactly the reference's Woodbury block (src/gradient.jl:187-210)."""

    def __init__(self, sigma: float = 0.0):
        if sigma < 0:
            raise DomainError(f"σ = {sigma} is negative")
        self.sigma = float(sigma)

    def __call__(self, x, y):
        x = np.atleast_1d(np.asarray(x, dtype=np.float64)); y = np.atleast_1d(np.asarray(y, dtype=np.float64))
        l = lambda u, v: float(np.dot(u, v)) + self.sigma
        return 2 / math.pi * math.asin(l(x, y) / math.sqrt((1 + l(x, x)) * (1 + l(y, y))))


NN = NeuralNetwork


class FiniteBasis(MercerKernel):
    """src/mercer.jl:41-70: k(x,y) = Σ_b b(x) b(y); basis functions are vectorised callables."""

    def __init__(self, basis: Sequence):
        if len(basis) < 1:
            raise ValueError(f"basis is empty: length(basis) = {len(basis)}")
        self.basis = list(basis)

    def __call__(self, x, y):
        return float(sum(b(x) * b(y) for b in self.basis))


# ----------------------------------------------------------------------------------------------
# algebra (src/algebra.jl) and Lengthscale (src/transformation.jl:6-19)
# ----------------------------------------------------------------------------------------------
def sum_and_product_input_trait(args):
    """src/properties.jl:47-63: common trait of the non-Constant arguments, else GenericInput."""
    non_const = [k for k in args if not isinstance(k, Constant)]
    if not non_const:
        return IsotropicInput()
    trait = input_trait(non_const[0])
    for k in non_const[1:]:
        if input_trait(k) != trait:
            return GenericInput()
    return trait


class Product(AbstractKernel):
    def __init__(self, args):
        self.args = tuple(args)
        self.input_trait = sum_and_product_input_trait(self.args)

    def __call__(self, *a):
        out = 1.0
        for k in self.args:
            out *= k(*a)
        return out


class Sum(AbstractKernel):
    def __init__(self, args):
        self.args = tuple(args)
        self.input_trait = sum_and_product_input_trait(self.args)

    def __call__(self, *a):
        return sum(k(*a) for k in self.args)


class Power(AbstractKernel):
    def __init__(self, k, p):
        if int(p) != p:
            raise TypeError("Power exponent must be an Int (src/algebra.jl:52)")
        self.k, self.p = k, int(p)
        self.input_trait = input_trait(k)

    def __call__(self, *a):
        return self.k(*a) ** self.p


class Lengthscale(IsotropicKernel):
    def __init__(self, k, l):
        if not isinstance(k, IsotropicKernel):
            raise TypeError("Lengthscale(k::IsotropicKernel, l)")
        if np.ndim(l) != 0 and np.size(l) != 1:
            raise _ffi.DimensionMismatch(_ffi.EINVAL, "lengthscale l has to has length 1")
        l = float(np.asarray(l).reshape(()))
        if not l > 0:
            raise DomainError(f"l = {l} is non-positive")    # src/transformation.jl:10
        self.k, self.l = k, l

    def profile(self, s):
        return self.k.profile(s / self.l ** 2)               # src/transformation.jl:19


class SeparableProduct(AbstractKernel):
    """src/algebra.jl:68-95: product kernel evaluating component kernels on separate input dimensions."""

    def __init__(self, *args):
        self.args = tuple(args[0]) if len(args) == 1 and isinstance(args[0], (tuple, list)) else tuple(args)

    def __call__(self, x, y):
        x = np.atleast_1d(x); y = np.atleast_1d(y)
        if len(x) != len(y):
            raise _ffi.DimensionMismatch(_ffi.EINVAL, f"length(x) ({len(x)}) ≠ length(y) ({len(y)})")
        if len(self.args) != len(x):
            raise _ffi.DimensionMismatch(_ffi.EINVAL, f"SeparableProduct needs d = {len(x)} kernels but has r = {len(self.args)}")
        out = 1.0
        for ki, xi, yi in zip(self.args, x, y):
            out *= ki(xi, yi)
        return out


def separable(op, *k):
    """src/algebra.jl:140-143."""
    import operator
    if op in (operator.mul, "*"):
        return SeparableProduct(*k)
    if op in (operator.pow, "^", "**"):
        kern, d = k
        return SeparableProduct(*([kern] * int(d)))
    raise NotImplementedError("separable(+, ...) has no structured Gramian in the reference either (src/algebra.jl:125-138)")


class SeparableKernel(MultiKernel):
    """SeparableKernel(B, k): matrix-valued kernel B * k(x, y) (src/separable.jl:2-16)."""

    def __init__(self, B, k=None):
        if isinstance(B, AbstractKernel):   # Separable(k, B) argument order used in test/separable.jl:13
            B, k = k, B
        self.B = np.asarray(B, dtype=np.float64)
        self.k = k

    def __call__(self, x, y):
        return self.B * self.k(x, y)


Separable = SeparableKernel


class GradientKernel(MultiKernel):
    """src/gradient.jl:7-24: the d×d block kernel ∂x ∂yᵀ k(x, y); carries input_trait(k)."""

    def __init__(self, k, it: Optional[InputTrait] = None):
        self.k = k
        self.input_trait = input_trait(k) if it is None else it


class ValueGradientKernel(MultiKernel):
    """src/gradient.jl:400-411: the (d+1)×(d+1) block kernel of [f, ∂f]; carries input_trait(k)."""

    def __init__(self, k, it: Optional[InputTrait] = None):
        self.k = k
        self.input_trait = input_trait(k) if it is None else it


# ----------------------------------------------------------------------------------------------
# input_trait (src/properties.jl:39-45, gradient.jl:16) — user-extensible like the reference
# ----------------------------------------------------------------------------------------------
_USER_TRAITS = {}


def register_input_trait(obj_or_type, trait: InputTrait):
    """Counterpart of defining `CovarianceFunctions.input_trait(::typeof(k)) = ...` (test/gramian.jl:163)."""
    _USER_TRAITS[obj_or_type if isinstance(obj_or_type, type) else id(obj_or_type)] = trait


def input_trait(k) -> InputTrait:
    if id(k) in _USER_TRAITS:
        return _USER_TRAITS[id(k)]
    for t, tr in _USER_TRAITS.items():
        if isinstance(t, type) and isinstance(k, t):
            return tr
    if isinstance(k, (Product, Sum, Power, GradientKernel, ValueGradientKernel)):
        return k.input_trait
    if isinstance(k, (Dot, ExponentialDot, AsinDot)):
        return DotProductInput()
    if isinstance(k, CosineKernel):
        return StationaryLinearFunctionalInput()             # src/stationary.jl:206
    if isinstance(k, IsotropicKernel):
        return IsotropicInput()
    if isinstance(k, StationaryKernel):
        return StationaryInput()
    return GenericInput()


def ismercer(k): return isinstance(k, MercerKernel) or (isinstance(k, (Product, Sum)) and all(ismercer(a) for a in k.args)) or (isinstance(k, Power) and ismercer(k.k))
def isstationary(k): return isinstance(k, StationaryKernel) or (isinstance(k, (Product, Sum)) and all(isstationary(a) for a in k.args)) or (isinstance(k, Power) and isstationary(k.k))
def isisotropic(k): return isinstance(k, IsotropicKernel) or (isinstance(k, (Product, Sum)) and all(isisotropic(a) for a in k.args)) or (isinstance(k, Power) and isisotropic(k.k))
def isdot(k): return isinstance(k, (Dot, ExponentialDot)) or (isinstance(k, (Product, Sum)) and all(isdot(a) for a in k.args)) or (isinstance(k, Power) and isdot(k.k))


# ----------------------------------------------------------------------------------------------
# lowering to the C struct
# ----------------------------------------------------------------------------------------------
_BASE = {
    ExponentiatedQuadratic: _ffi.EQ, Exponential: _ffi.EXP, RationalQuadratic: _ffi.RQ, GammaExponential: _ffi.GAMMAEXP,
    Cauchy: _ffi.CAUCHY, InverseMultiQuadratic: _ffi.IMQ, MaternP: _ffi.MATERNP, Dot: _ffi.DOT, ExponentialDot: _ffi.EXPDOT,
    Matern: _ffi.MATERN, AsinDot: _ffi.ASINDOT,
}


def _simple_spec(k) -> Optional[_ffi.covgram_kernel]:
    """covgram_kernel for a single profile, or None.

    Handles base profiles, Lengthscale nesting, Power (exponents multiply; (c·k)^p = c^p k^p) and
    products with Constants (scale) — the compositions that keep the IsotropicInput / DotProductInput
    trait of a single profile."""
    scale, power, ls = 1.0, 1, 1.0
    while True:
        if isinstance(k, Product):
            consts = [a for a in k.args if isinstance(a, Constant)]
            rest = [a for a in k.args if not isinstance(a, Constant)]
            if len(rest) != 1:
                return None
            for c in consts:
                scale *= c.c ** power
            k = rest[0]
        elif isinstance(k, Power):
            if k.p < 1:
                return None
            power *= k.p
            k = k.k
        elif isinstance(k, Lengthscale):
            ls *= k.l
            k = k.k
        else:
            break
    fam = _BASE.get(type(k))
    if fam is None:
        return None
    if fam == _ffi.MATERN and float(k.nu - 0.5).is_integer() and 0 <= int(k.nu - 0.5) <= _ffi.MATERNP_MAX_P:
        # half-integer ν: the same function has the closed form of MaternP(ν − 1/2) (the reference's own "IDEA: use rational types
        # to dispatch to MaternP evaluation", src/stationary.jl:85) — ~50x cheaper per pair than the Bessel-function path
        k = MaternP(int(k.nu - 0.5))
        fam = _ffi.MATERNP
    spec = _ffi.covgram_kernel()
    spec.family = fam
    spec.trait = _ffi.DOTPRODUCT if fam in (_ffi.DOT, _ffi.EXPDOT, _ffi.ASINDOT) else _ffi.ISOTROPIC
    spec.p = getattr(k, "p", 0) if fam == _ffi.MATERNP else 0
    spec.power = power
    spec.param = {_ffi.RQ: getattr(k, "alpha", 0.0), _ffi.GAMMAEXP: getattr(k, "gamma", 0.0), _ffi.IMQ: getattr(k, "c", 0.0),
                  _ffi.MATERN: getattr(k, "nu", 0.0)}.get(fam, 0.0)
    spec.lengthscale = ls
    spec.scale = scale
    if spec.trait == _ffi.DOTPRODUCT and ls != 1.0:
        return None
    return spec


def _expand(k, budget=64):
    """Sum-of-products normal form of a kernel expression: list of (coefficient, [simple specs]), or None.

    Sum concatenates, Product distributes over sums, Power of a composite multiplies out; a Constant is a bare
    coefficient.  Anything without a compiled profile (closures, Matern(ν), FiniteBasis, ...) gives None."""
    if isinstance(k, Constant):
        return [(k.c, [])]
    s = _simple_spec(k)
    if s is not None:
        c, s.scale = s.scale, 1.0
        return [(c, [s])]
    if isinstance(k, Sum):
        out = []
        for a in k.args:
            e = _expand(a, budget)
            if e is None:
                return None
            out += e
        return out if len(out) <= budget else None
    if isinstance(k, Product) or (isinstance(k, Power) and k.p >= 1):
        parts = k.args if isinstance(k, Product) else (k.k,) * k.p
        out = [(1.0, [])]
        for a in parts:
            e = _expand(a, budget)
            if e is None:
                return None
            out = [(c1 * c2, f1 + f2) for (c1, f1) in out for (c2, f2) in e]
            if len(out) > budget:
                return None
        return out
    return None


def _copy_spec(dst, src):
    for name, _ in _ffi.covgram_kernel._fields_:
        setattr(dst, name, getattr(src, name))


def device_spec(k):
    """What the device runs for `k`: a covgram_kernel (single profile), a covgram_kernel_composite (Sum / Product /
    Power of kernels sharing one input trait, src/algebra.jl:5-63 with the trait rule of src/properties.jl:47-63),
    or None if k is GenericInput in the reference's terms or exceeds the composite limits (4 terms, 6 factors)."""
    s = _simple_spec(k)
    if s is not None:
        return s
    terms = _expand(k)
    if not terms:
        return None
    # within a term, identical profiles multiply into one factor with a higher Power: phi^a phi^b = phi^(a+b)
    def _pow_merge(fs):
        out = {}
        for f in fs:
            key = (f.family, f.p, f.param, f.lengthscale)
            if key in out:
                out[key].power += f.power
            else:
                g = _ffi.covgram_kernel(); _copy_spec(g, f); out[key] = g
        return list(out.values())
    terms = [(c, _pow_merge(fs)) for c, fs in terms]
    # merge like terms (pure constants included), drop zero terms
    merged = {}
    for c, fs in terms:
        fs = sorted(fs, key=lambda f: (f.family, f.p, f.power, f.param, f.lengthscale))
        key = tuple((f.family, f.p, f.power, f.param, f.lengthscale) for f in fs)
        if key in merged:
            merged[key] = (merged[key][0] + c, fs)
        else:
            merged[key] = (c, fs)
    terms = [(c, fs) for key, (c, fs) in merged.items() if key and c != 0.0]
    if () in merged and merged[()][0] != 0.0:
        terms.append((merged[()][0], []))
    traits = {f.trait for _, fs in terms for f in fs}
    if len(traits) != 1:
        return None                                           # mixed traits: GenericInput (src/properties.jl:56-62)
    nfac = sum(max(len(fs), 1) for _, fs in terms)
    if not (1 <= len(terms) <= _ffi.COMPOSITE_MAX_TERMS) or nfac > _ffi.COMPOSITE_MAX_FACTORS:
        return None
    comp = _ffi.covgram_kernel_composite()
    comp.head.family = _ffi.COMPOSITE
    comp.head.trait = traits.pop()
    comp.head.p, comp.head.power = 0, 1
    comp.head.param, comp.head.lengthscale, comp.head.scale = 0.0, 1.0, 1.0
    comp.nterms = len(terms)
    fi = 0
    for t, (c, fs) in enumerate(terms):
        if not fs:
            comp.nfactors[t] = 1
            comp.factors[fi].family = _ffi.CONSTANT
            comp.factors[fi].trait = comp.head.trait
            comp.factors[fi].power, comp.factors[fi].lengthscale, comp.factors[fi].scale = 1, 1.0, c
            fi += 1
            continue
        comp.nfactors[t] = len(fs)
        for q, f in enumerate(fs):
            _copy_spec(comp.factors[fi], f)
            comp.factors[fi].scale = c if q == 0 else 1.0     # the term's coefficient rides on its first factor
            fi += 1
    return comp


def require_device_spec(k):
    spec = device_spec(k)
    if spec is None:
        raise _ffi.UnsupportedKernel(
            _ffi.EUNSUPPORTED,
            f"{type(k).__name__} has input_trait {input_trait(k)!r} / no compiled device profile; the reference would run its "
            "generic threaded loop (src/gramian.jl:78-87) here — this engine has no CPU fallback")
    return spec
