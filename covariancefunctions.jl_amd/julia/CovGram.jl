# CovGram.jl — the reference-side binding a maintainer would add: `ccall` stubs for libcovgram.so and the
# `mul!` methods that route the hot path of CovarianceFunctions.jl (v0.3.5) to the MI355X engine.
#
# WRITTEN BLIND: Julia is not installed in the build image, so this file has never been executed.  All logic
# lives in the C library; this shim is declarative (struct mirror, kernel lowering, ccall).  The executed and
# tested host mirror is the Python package next to it (covgram/).
#
# Usage:   using CovarianceFunctions, CovGram
#          G = gramian(EQ(), X)                 # unchanged, lazy, O(1)
#          CovGram.enable!()                    # from here on mul!(b, G, a) runs on the GPU when input_trait allows
module CovGram

using LinearAlgebra
using CovarianceFunctions
using CovarianceFunctions: Gramian, GradientKernel, IsotropicInput, DotProductInput, GenericInput, input_trait,
                           EQ, RQ, Exp, γExp, Cauchy, InverseMultiQuadratic, MaternP, Dot, ExponentialDot,
                           Lengthscale, Power, Product, Constant
import BlockFactorizations

const libcovgram = get(ENV, "COVGRAM_LIB", joinpath(@__DIR__, "..", "lib", "libcovgram.so"))

# mirrors `covgram_kernel` of include/covgram.h (field order and types are ABI)
struct CKernel
    family::Int32; trait::Int32; p::Int32; power::Int32
    param::Float64; lengthscale::Float64; scale::Float64
end
const F_EQ, F_EXP, F_RQ, F_GAMMAEXP, F_CAUCHY, F_IMQ, F_MATERNP, F_DOT, F_EXPDOT = Int32.(0:8)
const ISO, DOTP = Int32(1), Int32(2)
const HOST, DEVICE = Int32(0), Int32(1)
dtype_code(::Type{Float32}) = Int32(0); dtype_code(::Type{Float64}) = Int32(1)

check(rc) = rc == 0 ? nothing :
    (msg = unsafe_string(ccall((:covgram_last_error, libcovgram), Cstring, ()));
     rc == -1 ? throw(DimensionMismatch(msg)) : error("libcovgram status $rc: $msg"))

# lowering: the same closed set `covgram.kernels.device_spec` handles; anything else -> nothing (GenericInput path)
lower(k; scale = 1.0, power = 1, l = 1.0) = nothing
lower(::EQ; kw...)            = ckernel(F_EQ, ISO; kw...)
lower(::Exp; kw...)           = ckernel(F_EXP, ISO; kw...)
lower(k::RQ; kw...)           = ckernel(F_RQ, ISO; param = k.α, kw...)
lower(k::γExp; kw...)         = ckernel(F_GAMMAEXP, ISO; param = k.γ, kw...)
lower(::Cauchy; kw...)        = ckernel(F_CAUCHY, ISO; kw...)
lower(k::InverseMultiQuadratic; kw...) = ckernel(F_IMQ, ISO; param = k.c, kw...)
lower(k::MaternP; kw...)      = k.p ≤ 8 ? ckernel(F_MATERNP, ISO; p = k.p, kw...) : nothing
lower(::Dot; kw...)           = ckernel(F_DOT, DOTP; kw...)
lower(::ExponentialDot; kw...) = ckernel(F_EXPDOT, DOTP; kw...)
lower(k::Lengthscale; scale = 1.0, power = 1, l = 1.0) = lower(k.k; scale = scale, power = power, l = l * k.l)
lower(k::Power; scale = 1.0, power = 1, l = 1.0) = k.p ≥ 1 ? lower(k.k; scale = scale, power = power * k.p, l = l) : nothing
function lower(k::Product; scale = 1.0, power = 1, l = 1.0)
    rest = [a for a in k.args if !(a isa Constant)]
    length(rest) == 1 || return nothing
    c = prod(Float64[a.c for a in k.args if a isa Constant]; init = 1.0)
    lower(rest[1]; scale = scale * c^power, power = power, l = l)
end
ckernel(f, t; p = 0, param = 0.0, scale = 1.0, power = 1, l = 1.0) =
    (t == DOTP && l != 1.0) ? nothing : CKernel(f, t, Int32(p), Int32(power), Float64(param), Float64(l), Float64(scale))

# context: one per process/device, default (null) stream
const CTX = Ref{Ptr{Cvoid}}(C_NULL)
function ctx()
    if CTX[] == C_NULL
        check(ccall((:covgram_ctx_create, libcovgram), Cint, (Ref{Ptr{Cvoid}}, Cint, Ptr{Cvoid}), CTX, 0, C_NULL))
    end
    CTX[]
end

# device-resident copy of a point set, cached per Gramian input so Krylov solvers upload x once
mutable struct Points
    handle::Ptr{Cvoid}
    function Points(x::AbstractVector, ::Type{T}) where {T}
        d = length(x[1]); n = length(x)
        X = Matrix{T}(undef, d, n)                      # d×n column-major == point-major, as gramian.jl:2,154-155
        for (j, xj) in enumerate(x); X[:, j] .= xj; end
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:covgram_points_create, libcovgram), Cint,
                    (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Int64, Int32, Int32, Int32), ctx(), h, X, n, d, dtype_code(T), HOST))
        p = new(h[]); finalizer(q -> ccall((:covgram_points_destroy, libcovgram), Cint, (Ptr{Cvoid},), q.handle), p); p
    end
end
const POINTS = IdDict{Any, Points}()
points(x, T) = get!(() -> Points(x, T), POINTS, x)

const ENABLED = Ref(false)
enable!() = (ENABLED[] = true); disable!() = (ENABLED[] = false)

# --- src/gramian.jl:78-87 / 89-99 --------------------------------------------------------------------------------
function device_mul!(y::StridedVecOrMat{T}, G::Gramian{T}, a::StridedVecOrMat{T}, α, β, spec::CKernel) where {T <: Union{Float32, Float64}}
    n, m = size(G)
    size(a, 1) == m && size(y, 1) == n && size(y, 2) == size(a, 2) || throw(DimensionMismatch("mul!: size mismatch"))
    X = points(G.x, T); Y = G.x === G.y ? X : points(G.y, T)
    check(ccall((:covgram_mvm, libcovgram), Cint,
                (Ptr{Cvoid}, Ref{CKernel}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Int32, Float64, Float64, Int32),
                ctx(), spec, X.handle, Y.handle, a, stride(a, 2) == 0 ? m : max(stride(a, 2), m), y, max(stride(y, 2), n),
                size(a, 2), Float64(α), Float64(β), HOST))
    return y
end

function LinearAlgebra.mul!(y::StridedVector{T}, G::Gramian{T}, a::StridedVector{T}, α::Real = 1, β::Real = 0) where {T <: Union{Float32, Float64}}
    spec = ENABLED[] && input_trait(G.k) isa Union{IsotropicInput, DotProductInput} ? lower(G.k) : nothing
    spec === nothing ? invoke(mul!, Tuple{AbstractVector, Gramian, AbstractVector, Real, Real}, y, G, a, α, β) :
                       device_mul!(y, G, a, α, β, spec)
end
function LinearAlgebra.mul!(Y::StridedMatrix{T}, G::Gramian{T}, A::StridedMatrix{T}, α::Real = 1, β::Real = 0) where {T <: Union{Float32, Float64}}
    spec = ENABLED[] && input_trait(G.k) isa Union{IsotropicInput, DotProductInput} ? lower(G.k) : nothing
    spec === nothing ? invoke(mul!, Tuple{AbstractMatrix, Gramian, AbstractMatrix, Real, Real}, Y, G, A, α, β) :
                       device_mul!(Y, G, A, α, β, spec)
end

# --- src/gramian.jl:241-257 with src/gradient.jl:86-115: flat point-major block vectors ---------------------------
function device_gradmul!(y::StridedVector{T}, G::Gramian, a::StridedVector{T}, α, β, spec::CKernel) where {T}
    X = points(G.x, T); Y = G.x === G.y ? X : points(G.y, T)
    check(ccall((:covgram_grad_mvm, libcovgram), Cint,
                (Ptr{Cvoid}, Ref{CKernel}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Int32),
                ctx(), spec, X.handle, Y.handle, a, y, Float64(α), Float64(β), HOST))
    return y
end
function LinearAlgebra.mul!(y::StridedVector{T}, B::BlockFactorizations.BlockFactorization{T, <:Gramian{<:Any, <:GradientKernel}},
                            a::StridedVector{T}, α::Real = 1, β::Real = 0) where {T <: Union{Float32, Float64}}
    G = B.A
    spec = ENABLED[] && input_trait(G.k) isa Union{IsotropicInput, DotProductInput} ? lower(G.k.k) : nothing
    spec === nothing ? invoke(mul!, Tuple{AbstractVector, BlockFactorizations.BlockFactorization, AbstractVector, Real, Real}, y, B, a, α, β) :
                       device_gradmul!(y, G, a, α, β, spec)
end

# --- src/gradient.jl:400-474 (ValueGradientKernel), block mul! :319-351: blocks of d+1, value component first ----
function LinearAlgebra.mul!(y::StridedVector{T}, B::BlockFactorizations.BlockFactorization{T, <:Gramian{<:Any, <:CovarianceFunctions.ValueGradientKernel}},
                            a::StridedVector{T}, α::Real = 1, β::Real = 0) where {T <: Union{Float32, Float64}}
    G = B.A
    spec = ENABLED[] && input_trait(G.k) isa Union{IsotropicInput, DotProductInput} ? lower(G.k.k) : nothing
    spec === nothing && return invoke(mul!, Tuple{AbstractVector, BlockFactorizations.BlockFactorization, AbstractVector, Real, Real}, y, B, a, α, β)
    X = points(G.x, T); Y = G.x === G.y ? X : points(G.y, T)
    check(ccall((:covgram_valgrad_mvm, libcovgram), Cint,
                (Ptr{Cvoid}, Ref{CKernel}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Int32),
                ctx(), spec, X.handle, Y.handle, a, y, Float64(α), Float64(β), HOST))
    return y
end

# --- multi-GPU symmetric form (one Julia process per GPU, e.g. under MPI.jl; x and a replicated) -----------------------
# rank r of `world` evaluates the upper-triangle tiles of gramian(k, x) in the 256-row panels p % world == r and returns the
# partial product of those entries and their mirror images in `part` (device memory); an all-reduce (sum) of `part` over the
# ranks is G * a.  `covgram_mvm_sym_supported` tells whether the symmetric matrix-core kernel serves (k, x) — it depends on k
# and x only, so all ranks take the same branch; otherwise shard rows and all-gather (covgram_mvm on X[lo:hi] × X).
function sym_partial!(part::Ptr{Cvoid}, G::Gramian, a::Ptr{Cvoid}, rank::Integer, world::Integer)
    spec = lower(G.k); X = points(G.x, Float32)
    ok = Ref{Int32}(0)
    check(ccall((:covgram_mvm_sym_supported, libcovgram), Cint, (Ptr{Cvoid}, Ref{CKernel}, Ptr{Cvoid}, Ref{Int32}), ctx(), spec, X.handle, ok))
    ok[] == 1 || return false
    check(ccall((:covgram_mvm_sym_partial, libcovgram), Cint, (Ptr{Cvoid}, Ref{CKernel}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32),
                ctx(), spec, X.handle, a, part, Int32(rank), Int32(world)))
    return true
end

# --- Sum / Product / Power with a common trait (src/algebra.jl:5-63, src/properties.jl:47-63) ---------------------
# mirrors `covgram_kernel_composite`: k = head.scale * sum_t prod_f factors; every entry point above accepts a pointer to
# its first member (`head`, family = 101) in place of a CKernel.  Lowering = flatten the Sum/Product tree to a sum of
# products of `lower`-able leaves (covgram/kernels.py::device_spec is the executed version of this), e.g.
#     k = 1.5 * Lengthscale(MaternP(2), 0.7) + 0.5 * EQ()   ->   nterms = 2, nfactors = (1, 1, 0, 0),
#         factors = (CKernel(F_MATERNP, ISO, 2, 1, 0.0, 0.7, 1.5), CKernel(F_EQ, ISO, 0, 1, 0.0, 1.0, 0.5), ...)
struct CComposite
    head::CKernel                       # family = 101 (COVGRAM_COMPOSITE), trait = common trait, power = 1, lengthscale = 1
    nterms::Int32
    nfactors::NTuple{4, Int32}
    factors::NTuple{6, CKernel}         # family 100 (COVGRAM_CONSTANT) = a bare constant factor
end
# ccall signature: replace `Ref{CKernel}` by `Ref{CComposite}` (same address as its head).

# --- Toeplitz (src/gramian.jl:167-189): handle caches the plan and spectrum -------------------------------------
mutable struct DeviceToeplitz{T}
    handle::Ptr{Cvoid}; n::Int; m::Int
end
function DeviceToeplitz(vc::Vector{T}, vr::Union{Nothing, Vector{T}} = nothing; circulant = false) where {T}
    h = Ref{Ptr{Cvoid}}(C_NULL); n = length(vc); m = vr === nothing ? n : length(vr)
    check(ccall((:covgram_toeplitz_create, libcovgram), Cint,
                (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int32, Int32, Int32),
                ctx(), h, vc, vr === nothing ? C_NULL : vr, n, m, dtype_code(T), HOST, circulant ? 1 : 0))
    t = DeviceToeplitz{T}(h[], n, m)
    finalizer(q -> ccall((:covgram_toeplitz_destroy, libcovgram), Cint, (Ptr{Cvoid},), q.handle), t); t
end
function LinearAlgebra.mul!(y::StridedVector{T}, A::DeviceToeplitz{T}, a::StridedVector{T}, α::Real = 1, β::Real = 0) where {T}
    check(ccall((:covgram_toeplitz_mvm, libcovgram), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Int32),
                A.handle, a, y, Float64(α), Float64(β), HOST)); y
end
Base.size(A::DeviceToeplitz) = (A.n, A.m)

end # module
