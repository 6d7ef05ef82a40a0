# CovGram.jl — the reference-side binding a maintainer would add: `ccall` stubs for libcovgram.so and the methods that route
# the hot path of CovarianceFunctions.jl (v0.3.5) — `gramian` dispatch, `mul!`, `Matrix` — to the MI355X engine.
#
# WRITTEN BLIND: Julia is not installed in the build image, so this file has never been executed.  All logic lives in the C
# library; this shim is declarative (struct mirrors, kernel lowering, ccall).  What CAN be checked without Julia is checked:
# tests/test_host.py::test_julia_shim_mirrors_the_header parses the `struct` blocks and the `ccall` argument-type tuples below
# and compares them with include/covgram.h (tests/abi_layout.c compiles the header and prints sizeof / offsetof of every
# field), so a drift between this file and the header fails the CPU test suite.  At load time the module additionally
# asserts sizeof(CKernel) / sizeof(CComposite) against the library's own covgram_sizeof_*().
# The executed and tested host mirror is the Python package next to it (covgram/).
#
# Usage:   using CovarianceFunctions, CovGram
#          CovGram.enable!()                    # from here on the methods below take the device path where input_trait allows
#          G = gramian(EQ(), X)                 # unchanged, lazy, O(1)
#          mul!(b, G, a)                        # covgram_mvm
#          T = gramian(Exp(), range(-1, 1, length = 2^22))     # DeviceToeplitz (covgram_toeplitz_*)
#
# One context = one device + one stream (include/covgram.h: covgram_ctx_create(ctx, device_id, stream)); a multi-GPU job is
# one Julia process per GPU (MPI.jl), each with its own context — the library holds no multi-device state.
module CovGram

using LinearAlgebra
using CovarianceFunctions
using CovarianceFunctions: Gramian, GradientKernel, ValueGradientKernel, IsotropicInput, DotProductInput, StationaryInput,
                           GenericInput, input_trait, EQ, RQ, Exp, γExp, Cauchy, InverseMultiQuadratic, MaternP, Dot,
                           ExponentialDot, Lengthscale, Power, Product, Sum, Constant, SeparableProduct, SeparableKernel, LazyGrid, FiniteBasis
import CovarianceFunctions: gramian
import BlockFactorizations

const libcovgram = get(ENV, "COVGRAM_LIB", joinpath(@__DIR__, "..", "lib", "libcovgram.so"))

# ---- struct mirrors of include/covgram.h (field order and types are ABI; checked by tests/test_host.py) ------------------
const COMPOSITE_MAX_TERMS = 8        # COVGRAM_COMPOSITE_MAX_TERMS
const COMPOSITE_MAX_FACTORS = 8      # COVGRAM_COMPOSITE_MAX_FACTORS

# mirrors `covgram_kernel`
struct CKernel
    family::Int32
    trait::Int32
    p::Int32
    power::Int32
    param::Float64
    lengthscale::Float64
    scale::Float64
end

# mirrors `covgram_kernel_composite`: k = head.scale * sum_t prod_f factors; every entry point accepts a pointer to its first
# member (`head`, family = 101) in place of a CKernel
struct CComposite
    head::CKernel
    nterms::Int32
    nfactors::NTuple{8, Int32}
    factors::NTuple{8, CKernel}
end

const F_EQ, F_EXP, F_RQ, F_GAMMAEXP, F_CAUCHY, F_IMQ, F_MATERNP, F_DOT, F_EXPDOT = Int32.(0:8)
const F_CONSTANT, F_COMPOSITE = Int32(100), Int32(101)
const ISO, DOTP = Int32(1), Int32(2)
const HOST, DEVICE = Int32(0), Int32(1)
const ABI_VERSION = 113              # COVGRAM_VERSION of the header these ccall signatures mirror
dtype_code(::Type{Float32}) = Int32(0); dtype_code(::Type{Float64}) = Int32(1)
const DevFloat = Union{Float32, Float64}

function __init__()
    isfile(libcovgram) || return          # enable!() reports the missing library
    v = ccall((:covgram_version, libcovgram), Cint, ())
    v == ABI_VERSION || error("libcovgram.so is ABI version $v, CovGram.jl binds version $ABI_VERSION (include/covgram.h COVGRAM_VERSION): rebuild the library")
    sk = ccall((:covgram_sizeof_kernel, libcovgram), Cint, ())
    sc = ccall((:covgram_sizeof_composite, libcovgram), Cint, ())
    (sk == sizeof(CKernel) && sc == sizeof(CComposite)) ||
        error("CovGram.jl and libcovgram.so disagree on the ABI: covgram_kernel $(sizeof(CKernel)) vs $sk bytes, covgram_kernel_composite $(sizeof(CComposite)) vs $sc")
end

check(rc) = rc == 0 ? nothing :
    (msg = unsafe_string(ccall((:covgram_last_error, libcovgram), Cstring, ()));
     rc == -1 ? throw(DimensionMismatch(msg)) : error("libcovgram status $rc: $msg"))

# ---- lowering: the same closed set `covgram.kernels.device_spec` handles; anything else -> nothing (GenericInput path) ----
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

# Sum / Product / Power with a common trait (src/algebra.jl:5-63, src/properties.jl:47-63) -> CComposite, or nothing.
# A term is a Product (or a single kernel); each of its non-Constant factors must lower to a CKernel of the common trait.
# (covgram/kernels.py::device_spec is the executed version of this flattening, including the merge of like factors.)
const NOKERNEL = CKernel(F_CONSTANT, Int32(0), Int32(0), Int32(1), 0.0, 1.0, 0.0)
function lower_composite(k)
    terms = k isa Sum ? collect(k.args) : [k]
    length(terms) ≤ COMPOSITE_MAX_TERMS || return nothing
    nfac = zeros(Int32, COMPOSITE_MAX_TERMS); facs = CKernel[]; trait = Int32(0)
    for (t, term) in enumerate(terms)
        parts = term isa Product ? collect(term.args) : [term]
        for part in parts
            f = part isa Constant ? CKernel(F_CONSTANT, Int32(0), Int32(0), Int32(1), 0.0, 1.0, Float64(part.c)) : lower(part)
            f === nothing && return nothing
            if f.family != F_CONSTANT
                trait == 0 && (trait = f.trait)
                trait == f.trait || return nothing                       # mixed traits: GenericInput (src/properties.jl:47-63)
            end
            push!(facs, f); nfac[t] += 1
        end
    end
    (trait != 0 && length(facs) ≤ COMPOSITE_MAX_FACTORS) || return nothing
    while length(facs) < COMPOSITE_MAX_FACTORS; push!(facs, NOKERNEL); end
    CComposite(CKernel(F_COMPOSITE, trait, Int32(0), Int32(1), 0.0, 1.0, 1.0), Int32(length(terms)), Tuple(nfac), Tuple(facs))
end
# what the entry points take: Ref{CKernel}, or Ref{CComposite} (same address as its head)
device_kernel(k) = (s = lower(k); s === nothing ? lower_composite(k) : s)
device_kernel_for(k) = ENABLED[] && input_trait(k) isa Union{IsotropicInput, DotProductInput} ? device_kernel(k) : nothing
kref(s::CKernel) = Ref(s)
kref(s::CComposite) = Ref(s)

# ---- context: one per process = one device + its default stream ---------------------------------------------------------
const CTX = Ref{Ptr{Cvoid}}(C_NULL)
const DEVICE_ID = Ref{Cint}(0)           # set before the first call (e.g. from the MPI local rank): one process per GPU
function ctx()
    if CTX[] == C_NULL
        check(ccall((:covgram_ctx_create, libcovgram), Cint, (Ref{Ptr{Cvoid}}, Cint, Ptr{Cvoid}), CTX, DEVICE_ID[], C_NULL))
    end
    CTX[]
end

# ---- device-resident copy of a point set ---------------------------------------------------------------------------------
# Cached per (input object, element type) so Krylov solvers upload x once.  The library's contract for a handle is "unchanged
# while the handle lives" (include/covgram.h: covgram_points_create), whereas the reference's lazy Gramian follows in-place
# updates of x; the cache therefore keeps a fingerprint (length, dimension, a hash of ≤ 4096 strided scalars) and re-uploads
# when it changes; `CovGram.refresh!(x)` forces it after an update the sample might miss.  Keys are weak: the Points
# finalizer runs (and frees the HBM copy) once x itself is garbage.
mutable struct Points
    handle::Ptr{Cvoid}
    fingerprint::UInt
    function Points(x::AbstractVector, ::Type{T}) where {T}
        d = length(x[1]); n = length(x)
        X = Matrix{T}(undef, d, n)                      # d×n column-major == point-major, as gramian.jl:2,154-155
        for (j, xj) in enumerate(x); X[:, j] .= xj; end
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:covgram_points_create, libcovgram), Cint,
                    (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Int64, Int32, Int32, Int32), ctx(), h, X, n, d, dtype_code(T), HOST))
        p = new(h[], fingerprint(x))
        finalizer(q -> ccall((:covgram_points_destroy, libcovgram), Cint, (Ptr{Cvoid},), q.handle), p)
        p
    end
end
function fingerprint(x::AbstractVector)
    n = length(x); n == 0 && return UInt(0)
    d = length(x[1]); h = hash((n, d))
    for j in 1:max(1, n ÷ 1024):n
        xj = x[j]
        for c in 1:max(1, d ÷ 4):d; h = hash(xj[c], h); end
    end
    h
end
const POINTS = WeakKeyDict{Any, Dict{DataType, Points}}()
function points(x, ::Type{T}) where {T}
    per = get!(() -> Dict{DataType, Points}(), POINTS, x)
    p = get(per, T, nothing)
    if p === nothing || p.fingerprint != fingerprint(x)
        p = Points(x, T); per[T] = p                   # the old handle is freed by its finalizer
    end
    p
end
refresh!(x) = (delete!(POINTS, x); nothing)

const ENABLED = Ref(false)
function enable!()
    isfile(libcovgram) || error("$libcovgram not found: build it with `make -C covariancefunctions.jl_amd -j8`; there is no CPU fallback inside the library")
    ENABLED[] = true
end
disable!() = (ENABLED[] = false)

# --- src/gramian.jl:78-87 / 89-99 --------------------------------------------------------------------------------
function device_mul!(y::StridedVecOrMat{T}, G::Gramian, a::StridedVecOrMat{T}, α, β, spec) where {T <: DevFloat}
    n, m = size(G)
    size(a, 1) == m && size(y, 1) == n && size(y, 2) == size(a, 2) || throw(DimensionMismatch("mul!: size mismatch"))
    X = points(G.x, T); Y = G.x === G.y ? X : points(G.y, T)
    check(ccall((:covgram_mvm, libcovgram), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Int32, Float64, Float64, Int32),
                ctx(), kref(spec), X.handle, Y.handle, a, stride(a, 2) == 0 ? m : max(stride(a, 2), m), y, max(stride(y, 2), n),
                size(a, 2), Float64(α), Float64(β), HOST))
    return y
end

function LinearAlgebra.mul!(y::StridedVector{T}, G::Gramian{T}, a::StridedVector{T}, α::Real = 1, β::Real = 0) where {T <: DevFloat}
    spec = device_kernel_for(G.k)
    spec === nothing ? invoke(mul!, Tuple{AbstractVector, Gramian, AbstractVector, Real, Real}, y, G, a, α, β) :
                       device_mul!(y, G, a, α, β, spec)
end
function LinearAlgebra.mul!(Y::StridedMatrix{T}, G::Gramian{T}, A::StridedMatrix{T}, α::Real = 1, β::Real = 0) where {T <: DevFloat}
    spec = device_kernel_for(G.k)
    spec === nothing ? invoke(mul!, Tuple{AbstractMatrix, Gramian, AbstractMatrix, Real, Real}, Y, G, A, α, β) :
                       device_mul!(Y, G, A, α, β, spec)
end

# --- src/gramian.jl:102-114: Matrix(G) -> covgram_matrix (HBM-write-bound tile instantiation, copied back) ----------------
function Base.Matrix(G::Gramian{T}) where {T <: DevFloat}
    spec = device_kernel_for(G.k)
    spec === nothing && return invoke(Matrix, Tuple{Gramian}, G)
    n, m = size(G)
    M = Matrix{T}(undef, n, m)
    X = points(G.x, T); Y = G.x === G.y ? X : points(G.y, T)
    check(ccall((:covgram_matrix, libcovgram), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int32),
                ctx(), kref(spec), X.handle, Y.handle, M, n, HOST))
    return M
end

# --- src/gramian.jl:241-257 with src/gradient.jl:86-115: flat point-major block vectors ---------------------------
function device_blockmul!(sym::Symbol, y::StridedVecOrMat{T}, G::Gramian, a::StridedVecOrMat{T}, α, β, spec) where {T}
    X = points(G.x, T); Y = G.x === G.y ? X : points(G.y, T)
    size(y, 2) == size(a, 2) || throw(DimensionMismatch("mul!: y has $(size(y, 2)) columns, a has $(size(a, 2))"))
    lda = max(stride(a, 2), size(a, 1)); ldy = max(stride(y, 2), size(y, 1)); nrhs = Int32(size(a, 2))
    if sym === :grad
        check(ccall((:covgram_grad_mvm, libcovgram), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Int32, Float64, Float64, Int32),
                    ctx(), kref(spec), X.handle, Y.handle, a, lda, y, ldy, nrhs, Float64(α), Float64(β), HOST))
    else   # src/gradient.jl:400-474 (ValueGradientKernel), block mul! :319-351: blocks of d+1, value component first
        check(ccall((:covgram_valgrad_mvm, libcovgram), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Int32, Float64, Float64, Int32),
                    ctx(), kref(spec), X.handle, Y.handle, a, lda, y, ldy, nrhs, Float64(α), Float64(β), HOST))
    end
    return y
end
function LinearAlgebra.mul!(y::StridedVector{T}, B::BlockFactorizations.BlockFactorization{T, <:Gramian{<:Any, <:GradientKernel}},
                            a::StridedVector{T}, α::Real = 1, β::Real = 0) where {T <: DevFloat}
    G = B.A
    spec = device_kernel_for(G.k.k)
    spec === nothing ? invoke(mul!, Tuple{AbstractVector, BlockFactorizations.BlockFactorization, AbstractVector, Real, Real}, y, B, a, α, β) :
                       device_blockmul!(:grad, y, G, a, α, β, spec)
end
# matrix right-hand sides (src/gramian.jl:241-257: AbstractVecOfVecOrMat): one library call for all columns
function LinearAlgebra.mul!(Y::StridedMatrix{T}, B::BlockFactorizations.BlockFactorization{T, <:Gramian{<:Any, <:GradientKernel}},
                            A::StridedMatrix{T}, α::Real = 1, β::Real = 0) where {T <: DevFloat}
    G = B.A
    spec = device_kernel_for(G.k.k)
    spec === nothing ? invoke(mul!, Tuple{AbstractMatrix, BlockFactorizations.BlockFactorization, AbstractMatrix, Real, Real}, Y, B, A, α, β) :
                       device_blockmul!(:grad, Y, G, A, α, β, spec)
end
function LinearAlgebra.mul!(y::StridedVector{T}, B::BlockFactorizations.BlockFactorization{T, <:Gramian{<:Any, <:ValueGradientKernel}},
                            a::StridedVector{T}, α::Real = 1, β::Real = 0) where {T <: DevFloat}
    G = B.A
    spec = device_kernel_for(G.k.k)
    spec === nothing ? invoke(mul!, Tuple{AbstractVector, BlockFactorizations.BlockFactorization, AbstractVector, Real, Real}, y, B, a, α, β) :
                       device_blockmul!(:valgrad, y, G, a, α, β, spec)
end
function LinearAlgebra.mul!(Y::StridedMatrix{T}, B::BlockFactorizations.BlockFactorization{T, <:Gramian{<:Any, <:ValueGradientKernel}},
                            A::StridedMatrix{T}, α::Real = 1, β::Real = 0) where {T <: DevFloat}
    G = B.A
    spec = device_kernel_for(G.k.k)
    spec === nothing ? invoke(mul!, Tuple{AbstractMatrix, BlockFactorizations.BlockFactorization, AbstractMatrix, Real, Real}, Y, B, A, α, β) :
                       device_blockmul!(:valgrad, Y, G, A, α, β, spec)
end

# --- src/separable.jl:38-42: mul!(y, G::Gramian{<:AbstractMatrix, <:SeparableKernel}, x) on vectors of vectors -----------------
# The reference multiplies by kronecker(G) = gramian(k.k, x, y) ⊗ k.B (:33-35): block i of the result is B Σ_j k(x_i, y_j) a_j.  With
# the blocks of `a` as the ROWS of an m × q matrix that is (G_scalar A) Bᵀ: ONE covgram_mvm with q right-hand sides (every entry of
# the scalar Gramian evaluated once, O(n) memory — the Kronecker factor is never densified) and an n × q by q × p host product.
function LinearAlgebra.mul!(y::AbstractVector{<:AbstractVector{T}}, G::Gramian{<:AbstractMatrix, <:SeparableKernel},
                            a::AbstractVector{<:AbstractVector{T}}) where {T <: DevFloat}
    spec = device_kernel_for(G.k.k)
    spec === nothing && return invoke(mul!, Tuple{CovarianceFunctions.AbstractVecOfVec, Gramian{<:AbstractMatrix, <:SeparableKernel}, CovarianceFunctions.AbstractVecOfVec}, y, G, a)
    n, m = length(G.x), length(G.y); Bm = G.k.B; p, q = size(Bm)
    (length(a) == m && length(y) == n) || throw(DimensionMismatch("mul!: $(length(y)) / $(length(a)) blocks for a $n × $m block Gramian"))
    A = Matrix{T}(undef, m, q)
    for j in 1:m
        length(a[j]) == q || throw(DimensionMismatch("mul!: block $j of a has length $(length(a[j])) ≠ $q"))
        A[j, :] .= a[j]
    end
    GA = Matrix{T}(undef, n, q)
    device_mul!(GA, Gramian(G.k.k, G.x, G.y), A, 1, 0, spec)
    for i in 1:n
        mul!(y[i], Bm, view(GA, i, :))
    end
    return y
end

# --- multi-GPU symmetric form (one Julia process per GPU, e.g. under MPI.jl; x and a replicated) -----------------------
# rank r of `world` evaluates its cyclic share of the upper triangle of gramian(k, x) — 256-row panels on the symmetric matrix-core
# kernels (Float32), 64-row / 64 R-row blocks on the direct-difference symmetric kernels (Float64, and Float32 where the matrix cores
# do not apply) — and returns the partial product of those entries and their mirror images in `part` (device memory); an all-reduce
# (sum) of `part` over the ranks is G * a.  `covgram_mvm_sym_supported` tells whether a symmetric kernel serves (k, x) at this world
# size — it depends on k, x and world only, so all ranks take the same branch; otherwise shard rows and all-gather (covgram_mvm on
# X[lo:hi] × X).  The element type is the Gramian's own (src/gramian.jl:27-33).
function sym_partial!(part::Ptr{Cvoid}, G::Gramian{T}, a::Ptr{Cvoid}, rank::Integer, world::Integer) where {T <: DevFloat}
    spec = device_kernel(G.k); spec === nothing && return false
    X = points(G.x, T)
    ok = Ref{Int32}(0)
    check(ccall((:covgram_mvm_sym_supported, libcovgram), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ref{Int32}),
                ctx(), kref(spec), X.handle, Int32(world), ok))
    ok[] == 1 || return false
    check(ccall((:covgram_mvm_sym_partial, libcovgram), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32),
                ctx(), kref(spec), X.handle, a, part, Int32(rank), Int32(world)))
    return true
end

# --- the collective behind the ABI (include/covgram.h, "the collective behind the ABI"): mul! on a Gramian sharded over the GPUs of a node ------
# One Julia process per GPU.  `comm_unique_id()` on rank 0, its 128 bytes broadcast by whatever the job already has (MPI.bcast!, a shared
# file), then `comm_create!(id, rank, world)` on every rank: the library owns the RCCL communicator from there on, and
#     mul!(y, ShardedGramian(G), a, α, β)
# is the reference's mul!(y, G, a, α, β) (src/gramian.jl:78-87) with its `@threads for i in 1:n` spread over the ranks: each evaluates its
# rows (or, for gramian(k, x), its cyclic panels of the upper triangle) and ONE all-gather (all-reduce) on the library's stream completes y
# on every rank.  x, a and y are device pointers here (replicated data stays resident between Krylov iterations).
const COMM_ID_BYTES = 128
function comm_unique_id()
    id = zeros(UInt8, COMM_ID_BYTES)
    check(ccall((:covgram_comm_unique_id, libcovgram), Cint, (Ptr{Cvoid}, Int64), id, Int64(COMM_ID_BYTES)))
    return id
end
comm_create!(id::Vector{UInt8}, rank::Integer, world::Integer) =
    check(ccall((:covgram_comm_create, libcovgram), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32), ctx(), id, Int32(rank), Int32(world)))
comm_destroy!() = check(ccall((:covgram_comm_destroy, libcovgram), Cint, (Ptr{Cvoid},), ctx()))
function comm_info()
    r = Ref{Int32}(0); w = Ref{Int32}(0)
    check(ccall((:covgram_comm_info, libcovgram), Cint, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}), ctx(), r, w))
    return Int(r[]), Int(w[])
end
all_gather!(recv::Ptr{Cvoid}, send::Ptr{Cvoid}, count::Integer, ::Type{T}) where {T <: DevFloat} =
    check(ccall((:covgram_comm_all_gather, libcovgram), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int32), ctx(), send, recv, Int64(count), dtype_code(T)))
all_reduce_sum!(buf::Ptr{Cvoid}, count::Integer, ::Type{T}) where {T <: DevFloat} =
    check(ccall((:covgram_comm_all_reduce_sum, libcovgram), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int32), ctx(), buf, Int64(count), dtype_code(T)))

struct ShardedGramian{T, GT <: Gramian{T}} <: AbstractMatrix{T}
    G::GT
end
Base.size(S::ShardedGramian) = size(S.G)
# y, a: device pointers to n resp. m replicated scalars of the Gramian's element type
function LinearAlgebra.mul!(y::Ptr{Cvoid}, S::ShardedGramian{T}, a::Ptr{Cvoid}, α::Real = 1, β::Real = 0) where {T <: DevFloat}
    G = S.G
    spec = device_kernel(G.k); spec === nothing && error("ShardedGramian: the kernel has no device path (GenericInput)")
    X = points(G.x, T)
    if G.x === G.y                                          # gramian(k, x): the symmetric form where a symmetric kernel serves it
        rc = ccall((:covgram_mvm_sym_allreduce, libcovgram), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64),
                   ctx(), kref(spec), X.handle, a, y, Float64(α), Float64(β))
        rc == 0 && return y
        rc == -2 || check(rc)                               # COVGRAM_EUNSUPPORTED: row shards below
    end
    Y = G.x === G.y ? X : points(G.y, T)
    check(ccall((:covgram_mvm_sharded, libcovgram), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64),
                ctx(), kref(spec), X.handle, Y.handle, a, y, Float64(α), Float64(β)))
    return y
end

# --- Toeplitz (src/gramian.jl:167-189): the handle caches the plans and the spectrum of the circulant embedding ----------
mutable struct DeviceToeplitz{T} <: AbstractMatrix{T}
    handle::Ptr{Cvoid}
    n::Int
    m::Int
    vc::Vector{T}
    vr::Union{Nothing, Vector{T}}
    circulant::Bool
end
function DeviceToeplitz(vc::Vector{T}, vr::Union{Nothing, Vector{T}} = nothing; circulant = false) where {T <: DevFloat}
    h = Ref{Ptr{Cvoid}}(C_NULL); n = length(vc); m = vr === nothing ? n : length(vr)
    check(ccall((:covgram_toeplitz_create, libcovgram), Cint,
                (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int32, Int32, Int32),
                ctx(), h, vc, vr === nothing ? C_NULL : vr, n, m, dtype_code(T), HOST, circulant ? 1 : 0))
    t = DeviceToeplitz{T}(h[], n, m, vc, vr, circulant)
    finalizer(q -> ccall((:covgram_toeplitz_destroy, libcovgram), Cint, (Ptr{Cvoid},), q.handle), t); t
end
function LinearAlgebra.mul!(y::StridedVector{T}, A::DeviceToeplitz{T}, a::StridedVector{T}, α::Real = 1, β::Real = 0) where {T}
    check(ccall((:covgram_toeplitz_mvm, libcovgram), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Int32),
                A.handle, a, y, Float64(α), Float64(β), HOST)); y
end
Base.size(A::DeviceToeplitz) = (A.n, A.m)
function Base.getindex(A::DeviceToeplitz, i::Integer, j::Integer)   # T[i,j] = vc[i-j+1] (i ≥ j), vr[j-i+1] (i < j)
    A.circulant && return A.vc[mod(i - j, A.n) + 1]
    i ≥ j ? A.vc[i - j + 1] : (A.vr === nothing ? A.vc[j - i + 1] : A.vr[j - i + 1])
end
LinearAlgebra.issymmetric(A::DeviceToeplitz) = A.vr === nothing && !A.circulant

# gramian(k, x::StepRangeLen, y::StepRangeLen, trait): same three cases as src/gramian.jl:172-183; floating-point ranges only
# (more specific than the reference's method, which stays the fallback)
function gramian(k, x::StepRangeLen{T}, y::StepRangeLen{T}, t::Union{IsotropicInput, StationaryInput}) where {T <: DevFloat}
    ENABLED[] || return invoke(gramian, Tuple{Any, StepRangeLen, StepRangeLen, Union{IsotropicInput, StationaryInput}}, k, x, y, t)
    if x === y
        DeviceToeplitz(T.(k.(x[1], x)))                                   # SymmetricToeplitz(k1), src/gramian.jl:174-175
    elseif x.step == y.step
        DeviceToeplitz(T.(k.(x, y[1])), T.(k.(x[1], y)))                  # Toeplitz(k1, k2), :177-179
    else
        Gramian(k, x, y)                                                  # :181
    end
end
# periodic boundary conditions: Circulant(k1), src/gramian.jl:186-189
function gramian(k::CovarianceFunctions.StationaryKernel, x::StepRangeLen{T}, p::CovarianceFunctions.PeriodicInput) where {T <: DevFloat}
    ENABLED[] || return invoke(gramian, Tuple{CovarianceFunctions.StationaryKernel, StepRangeLen, CovarianceFunctions.PeriodicInput}, k, x, p)
    DeviceToeplitz(T.(k.(x[1], x)); circulant = true)
end

# --- direct Toeplitz solvers (src/toeplitz.jl:12-111): the O(n²) chains on one workgroup of the device --------------------
# unit-diagonal forms, as the reference's vector methods; `levinson(T, b)` / `trench(T)` normalise by T.vc[1] (the reference's
# `r_0 == 1` test is inverted, src/toeplitz.jl:40-42,103-105; done right here)
function device_durbin(r::Vector{T}) where {T <: DevFloat}
    y = similar(r)
    check(ccall((:covgram_toeplitz_durbin, libcovgram), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int32, Int32),
                ctx(), r, length(r), y, dtype_code(T), HOST)); y
end
function device_levinson(r::Vector{T}, b::Vector{T}) where {T <: DevFloat}
    length(b) == length(r) + 1 || throw(DimensionMismatch("length(b) = $(length(b)) ≠ $(length(r) + 1) = length(r) + 1"))
    x = similar(b)
    check(ccall((:covgram_toeplitz_levinson, libcovgram), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int32, Int32),
                ctx(), r, b, length(b), x, dtype_code(T), HOST)); x
end
function device_trench(r::Vector{T}) where {T <: DevFloat}
    n = length(r) + 1; B = Matrix{T}(undef, n, n)
    check(ccall((:covgram_toeplitz_trench, libcovgram), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Int32, Int32),
                ctx(), r, n, B, n, dtype_code(T), HOST)); Symmetric(B)
end
device_levinson(A::DeviceToeplitz{T}, b::Vector{T}) where {T} = device_levinson(A.vc[2:end] ./ A.vc[1], b) ./ A.vc[1]
device_trench(A::DeviceToeplitz) = Symmetric(parent(device_trench(A.vc[2:end] ./ A.vc[1])) ./ A.vc[1])
# `\`: the reference's `\` on a SymmetricToeplitz is a DIRECT solve, and so is this one wherever the library serves it: Levinson is ONE
# launch of n - 1 dependent O(n) steps on one workgroup (n <= 16384: state in registers and LDS, 28 ms at n = 16384 fp64; above that the
# vectors live in global memory, ~8 us a step), up to COVGRAM_TOEPLITZ_DIRECT_MAX_N = 65536.  Beyond it the library returns
# COVGRAM_EUNSUPPORTED and the solve is conjugate gradients over the FFT MVM (`mul!` above: covgram_toeplitz_mvm; the reference's own
# solver for lazy operators, src/gramian.jl:229-238) with an explicit tolerance, and it ERRORS when CG did not converge — noise-free
# Gramians are ill-conditioned, and IterativeSolvers.cg returns its last iterate silently (ADVICE r3).
const TOEPLITZ_DIRECT_MAX_N = 65536          # COVGRAM_TOEPLITZ_DIRECT_MAX_N
function LinearAlgebra.:\(A::DeviceToeplitz{T}, b::Vector{T}) where {T}
    issymmetric(A) || error("only symmetric Toeplitz solves are served")
    length(b) <= TOEPLITZ_DIRECT_MAX_N && return device_levinson(A, b)
    x, hist = CovarianceFunctions.IterativeSolvers.cg(A, b; reltol = eps(T)^(2 / 3), maxiter = 4 * ceil(Int, sqrt(length(b))) + 200, log = true)
    hist.isconverged || error("Toeplitz solve of order $(length(b)): CG over the FFT MVM did not converge in $(hist.iters) iterations " *
                              "(relative tolerance $(eps(T)^(2 / 3))); add a noise term or use a preconditioned solver")
    x
end

# --- Kronecker (src/algebra.jl:91-95, src/separable.jl:33-42): dense factors, mode products on the matrix cores (csrc/kron.hip) --
struct DeviceKronecker{T} <: AbstractMatrix{T}
    factors::Vector{Matrix{T}}          # F_1 ⊗ F_2 ⊗ … ⊗ F_q in the reference's order (kronecker(G_1, …, G_q))
end
Base.size(K::DeviceKronecker) = (prod(size(F, 1) for F in K.factors), prod(size(F, 2) for F in K.factors))
function Base.getindex(K::DeviceKronecker, i::Integer, j::Integer)
    v = one(eltype(K)); i -= 1; j -= 1
    for F in reverse(K.factors)                                           # the LAST factor's index varies fastest
        r, c = size(F); v *= F[i % r + 1, j % c + 1]; i ÷= r; j ÷= c
    end
    v
end
function LinearAlgebra.mul!(y::StridedVecOrMat{T}, K::DeviceKronecker{T}, a::StridedVecOrMat{T}, α::Real = 1, β::Real = 0) where {T <: DevFloat}
    q = length(K.factors)
    n, m = size(K)
    size(a, 1) == m && size(y, 1) == n && size(y, 2) == size(a, 2) || throw(DimensionMismatch("mul!: size mismatch"))
    ptrs = Ptr{Cvoid}[pointer(F) for F in K.factors]
    rows = Int64[size(F, 1) for F in K.factors]; cols = Int64[size(F, 2) for F in K.factors]
    GC.@preserve K check(ccall((:covgram_kron_mvm, libcovgram), Cint,
        (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Int32, Int32, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Int32, Float64, Float64, Int32),
        ctx(), ptrs, rows, cols, rows, Int32(q), dtype_code(T), a, max(stride(a, 2), m), y, max(stride(y, 2), n), Int32(size(a, 2)),
        Float64(α), Float64(β), HOST))
    y
end
function gramian(k::SeparableProduct, X::LazyGrid{T}, Y::LazyGrid{T}) where {T <: DevFloat}
    ENABLED[] || return invoke(gramian, Tuple{SeparableProduct, LazyGrid, LazyGrid}, k, X, Y)
    length(X.args) == length(Y.args) || throw(DimensionMismatch("length(X.args) = $(length(X.args)) ≠ $(length(Y.args)) = length(Y.args)"))
    length(k.args) == length(X.args) || throw(DimensionMismatch("SeparableProduct needs d = $(length(X.args)) kernels but has r = $(length(k.args))"))
    DeviceKronecker{T}([Matrix{T}(Matrix(gramian(ki, xi, yi))) for (ki, xi, yi) in zip(k.args, X.args, Y.args)])
end

# --- low rank (src/mercer.jl:53-70, src/lazy_linear_algebra.jl:78-85): U (V' a), vector or matrix right-hand sides -------
struct DeviceLowRank{T} <: AbstractMatrix{T}
    U::Matrix{T}
    V::Matrix{T}
end
Base.size(L::DeviceLowRank) = (size(L.U, 1), size(L.V, 1))
Base.getindex(L::DeviceLowRank, i::Integer, j::Integer) = dot(view(L.U, i, :), view(L.V, j, :))
function LinearAlgebra.mul!(y::StridedVecOrMat{T}, L::DeviceLowRank{T}, a::StridedVecOrMat{T}, α::Real = 1, β::Real = 0) where {T <: DevFloat}
    n, m = size(L); r = size(L.U, 2)
    size(a, 1) == m && size(y, 1) == n && size(y, 2) == size(a, 2) || throw(DimensionMismatch("mul!: size mismatch"))
    check(ccall((:covgram_lowrank_mvm, libcovgram), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Int64, Int64, Int64, Int32, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Int32, Float64, Float64, Int32),
                ctx(), L.U, n, L.V, m, n, m, r, dtype_code(T), a, max(stride(a, 2), m), y, max(stride(y, 2), n), Int32(size(a, 2)),
                Float64(α), Float64(β), HOST))
    y
end
function gramian(k::FiniteBasis{T}, x::AbstractVector, y::AbstractVector) where {T <: DevFloat}
    r = length(k.basis)
    (ENABLED[] && length(x) > r && length(y) > r) || return invoke(gramian, Tuple{FiniteBasis, AbstractVector, AbstractVector}, k, x, y)
    U = CovarianceFunctions.basis(k, x)
    V = x === y ? U : CovarianceFunctions.basis(k, y)
    DeviceLowRank{T}(U, V)
end

end # module
