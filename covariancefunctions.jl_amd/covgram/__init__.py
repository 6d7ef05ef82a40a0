"""covgram — MI355X-native lazy-Gramian MVM engine behind the CovarianceFunctions.jl API surface.

    import covgram as cg
    G = cg.gramian(cg.EQ(), x)          # x: (n, d) tensor on the GPU; O(1), lazy
    b = G @ a                           # mul!(b, G, a) — hand-written HIP kernel in libcovgram.so
    cg.mul_(b, G, a, alpha, beta)       # 5-argument mul!

Only the hot path `mul!(b, gramian(k, x), a)` and its GradientKernel / Toeplitz / Kronecker / low-rank
siblings exist here.  No CPU fallback: without libcovgram.so or a gfx950 device every
product raises.
"""
from . import _ffi
from ._ffi import CovgramError, DimensionMismatch, UnsupportedKernel, NoDevice
from .kernels import (AbstractKernel, MercerKernel, StationaryKernel, IsotropicKernel, MultiKernel, Constant,
                      ExponentiatedQuadratic, EQ, RationalQuadratic, RQ, Exponential, Exp, GammaExponential, GammaExp,
                      Cauchy, InverseMultiQuadratic, MaternP, Matern, Dot, ExponentialDot, FiniteBasis, Product, Sum, Power,
                      Lengthscale, ARD, ScaledInputKernel, Warped, Periodic, VerticalRescaling, CosineKernel, Cosine, Cos, Line, Polynomial, Poly, NeuralNetwork, NN, AsinDot,
                      SeparableProduct, separable, SeparableKernel, Separable, GradientKernel, ValueGradientKernel, InputTrait,
                      GenericInput, IsotropicInput, DotProductInput, StationaryInput, StationaryLinearFunctionalInput,
                      PeriodicInput, input_trait, register_input_trait, ismercer, isstationary, isisotropic, isdot,
                      device_spec, DomainError)
from .gramian import (Gramian, BlockGramian, SymmetricToeplitz, Toeplitz, Circulant, KroneckerProduct, kronecker,
                      SeparableGramian, LazyMatrixProduct, LazyMatrixSum, ScaledOperator, LinearMapBlockGramian, CosineBlockGramian, PointJacobianBlockGramian, Fill, LazyOperator, LazyGrid, StepRangeLen,
                      srange, gramian, mul_, get_ctx, set_option, get_info, kernel_time)
from .dist import ShardedGramian, shard_bounds
from .solve import cg, solve, toeplitz_solve, durbin, levinson, trench
from .factorize import cholesky, factorize, diagonal, CholeskyFactor, PivotedCholesky

__all__ = [n for n in dir() if not n.startswith("_")]
