/* covgram.h — C ABI of libcovgram.so, the MI355X (gfx950) lazy-Gramian MVM engine.
 *
 * Drop-in boundary for ONE hot path of SebastianAment/CovarianceFunctions.jl (v0.3.5):
 *     mul!(b, gramian(k, x[, y]), a, α, β)
 * and its structured siblings.  The reference is pure Julia with no FFI of its own, so the
 * boundary is the set of Julia methods listed below; a Julia shim (`ccall`) or the Python
 * `ctypes` binding in covariancefunctions.jl_amd/covgram/_ffi.py binds exactly these symbols.
 * Citations are file:line relative to the reference repository root.
 *
 *   covgram_mvm            replaces  LinearAlgebra.mul!(y::AbstractVector, G::Gramian, x, α, β)   src/gramian.jl:78-87
 *                          and       LinearAlgebra.mul!(Y::AbstractMatrix, G::Gramian, X, α, β)   src/gramian.jl:89-99
 *                          and       Base.:*(G::Gramian, a)                                       src/gramian.jl:66-75
 *   covgram_matrix         replaces  Base.Matrix(G::Gramian) / Matrix!                            src/gramian.jl:102-114
 *   covgram_grad_mvm       replaces  BlockFactorizations.blockmul!(y, G::Gramian, x, α, β)        src/gramian.jl:241-257
 *                          with the  GradientKernelElement mul! (isotropic / dot-product)         src/gradient.jl:86-92, 109-115
 *   covgram_valgrad_mvm    the same  blockmul! with the ValueGradientKernel element                       src/gradient.jl:319-351, 400-474
 *   covgram_toeplitz_*     replaces  mul!(y, ::SymmetricToeplitz/Toeplitz/Circulant, a, α, β) of ToeplitzMatrices 0.7.1 as
 *                          constructed by gramian(k, x::StepRangeLen, y::StepRangeLen)            src/gramian.jl:167-189
 *   covgram_toeplitz_durbin / _levinson / _trench  replace durbin! / levinson! / trench!             src/toeplitz.jl:12-111
 *   covgram_kron_mvm       replaces  mul!(y, ::KroneckerProduct, a) of KroneckerProducts 1.1.1 as constructed at
 *                                                                                                 src/algebra.jl:91-95, src/separable.jl:33-42
 *   covgram_lowrank_mvm    replaces  mul!(y, L::LazyMatrixProduct(U, V'), a, α, β)                src/lazy_linear_algebra.jl:78-85
 *                          as built by gramian(k::FiniteBasis, x, y)                              src/mercer.jl:61-70
 *   covgram_kernel         encodes   the kernel value k together with input_trait(k)              src/properties.jl:31-45
 *   covgram_kernel_composite encodes Sum / Product / Power of same-trait kernels                  src/algebra.jl:5-63, src/properties.jl:47-63
 *
 * Conventions (same as the reference's at that boundary):
 *   - the caller owns every buffer it passes; handles own only what the library allocated;
 *   - points are point-major: d contiguous scalars per point (Julia d×n column-major matrix ==
 *     Vector of d-vectors, src/gramian.jl:2,154-155);
 *   - flat block vectors of the gradient Gramian are point-major (block i = entries i*d..i*d+d-1);
 *   - matrices (right-hand sides, dense factors) are column-major with an explicit leading dimension;
 *   - beta == 0 means the previous contents of y are NOT read (NaN-safe, src/gramian.jl:80,90,245);
 *   - every function returns 0 on success and a negative covgram_status on failure; no exception
 *     crosses the ABI; covgram_last_error() gives a thread-local message;
 *   - unlike the reference (which runs @inbounds), dimension mismatches are reported as errors;
 *   - there is NO CPU fallback: every compute entry point needs a gfx950 device.
 */
#ifndef COVGRAM_H
#define COVGRAM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COVGRAM_VERSION 113 /* 0.1.1: covgram_kron_mvm, covgram_grad_mvm and covgram_valgrad_mvm take (lda, ldy, nrhs); 111 adds
                               covgram_cg_step_shifted; 112: covgram_mvm_sym_supported takes `world` (the symmetric partial form's
                               column-sum slab depends on it), fp32 direct-difference symmetric partials; 113: the communicator
                               (covgram_comm_*), covgram_mvm_sharded, covgram_mvm_sym_allreduce.
                               A binding checks covgram_version() against the header it mirrors at load time */

typedef enum covgram_status {
    COVGRAM_OK = 0,
    COVGRAM_EINVAL = -1,       /* bad argument / dimension mismatch */
    COVGRAM_EUNSUPPORTED = -2, /* kernel family / dimension / dtype outside the compiled set */
    COVGRAM_EHIP = -3,         /* HIP / rocFFT runtime failure (message in covgram_last_error) */
    COVGRAM_ENODEVICE = -4,    /* no gfx950 device visible: the product path fails loudly */
    COVGRAM_ENOMEM = -5
} covgram_status;

/* Scalar profile phi(s); s = |x-y|^2 (isotropic) or x.y (dot product). */
typedef enum covgram_family {
    COVGRAM_EQ = 0,       /* exp(-s/2)                         src/stationary.jl:37-42   */
    COVGRAM_EXP = 1,      /* exp(-sqrt(s))                     src/stationary.jl:56-60   */
    COVGRAM_RQ = 2,       /* (1 + s/(2 alpha))^-alpha          src/stationary.jl:45-53   */
    COVGRAM_GAMMAEXP = 3, /* exp(-s^(gamma/2)/2)               src/stationary.jl:63-71   */
    COVGRAM_CAUCHY = 4,   /* 1/(1+s)                           src/stationary.jl:221-224 */
    COVGRAM_IMQ = 5,      /* 1/sqrt(s + c^2)                   src/stationary.jl:231-235 */
    COVGRAM_MATERNP = 6,  /* Matern nu = p + 1/2, 0 <= p <= 8  src/stationary.jl:117-158 */
    COVGRAM_DOT = 7,      /* s                                 src/mercer.jl:6-9         */
    COVGRAM_EXPDOT = 8,   /* exp(s)                            src/mercer.jl:19-22       */
    COVGRAM_MATERN = 9,   /* Matern, real nu > 0: 2^(1-nu)/Gamma(nu) r^nu K_nu(r), r = sqrt(2 nu s); Taylor guard near 0
                             (param = nu)                                        src/stationary.jl:87-114  */
    COVGRAM_ASINDOT = 10, /* (2/pi) asin(s): the NeuralNetwork kernel on its normalised augmented inputs x^ = [x, sqrt(sigma)] /
                             sqrt(1 + |x|^2 + sigma), for which x^.y^ is the argument of asin   src/mercer.jl:73-85 */
    COVGRAM_NFAMILY = 11,
    /* only inside / as the head of a covgram_kernel_composite: */
    COVGRAM_CONSTANT = 100, /* factor: the constant `scale`            src/stationary.jl:27-34   */
    COVGRAM_COMPOSITE = 101 /* head of a covgram_kernel_composite      src/algebra.jl:5-63       */
} covgram_family;

/* input_trait(k), src/properties.jl:31-45.  Only the two traits with a device path are encoded;
 * GenericInput kernels never reach the library (the host falls back exactly as gramian.jl:78-87). */
typedef enum covgram_trait { COVGRAM_ISOTROPIC = 1, COVGRAM_DOTPRODUCT = 2 } covgram_trait;

typedef enum covgram_dtype { COVGRAM_F32 = 0, COVGRAM_F64 = 1 } covgram_dtype;
typedef enum covgram_loc { COVGRAM_HOST = 0, COVGRAM_DEVICE = 1 } covgram_loc;

#define COVGRAM_MATERNP_MAX_P 8

typedef struct covgram_kernel {
    int32_t family;     /* covgram_family */
    int32_t trait;      /* covgram_trait; must agree with the family (checked) */
    int32_t p;          /* MaternP order */
    int32_t power;      /* Power(k, p) exponent, >= 1 (src/algebra.jl:50-63); 1 = none */
    double param;       /* RQ alpha | gammaExp gamma | IMQ c | Matern nu */
    double lengthscale; /* Lengthscale(k, l): s <- s/l^2, isotropic only (src/transformation.jl:6-19); 1 = none */
    double scale;       /* Constant(c) * k (src/algebra.jl:23-25); 1 = none */
} covgram_kernel;

/* Sum / Product / Power of kernels that share one input trait (src/algebra.jl:5-63; the common trait is what
 * src/properties.jl:47-63 computes, Constant arguments do not count):
 *     k(x, y) = head.scale * sum_{t < nterms}  prod_{f < nfactors[t]}  factor(x, y)
 * with the factors of term 0 first in `factors`, then term 1's, ...  Each factor is a simple covgram_kernel with its
 * own scale, lengthscale and power (or family COVGRAM_CONSTANT: just `scale`).  Every entry point that takes a
 * `const covgram_kernel*` accepts `&composite.head` (head.family == COVGRAM_COMPOSITE, head.trait = the common trait,
 * head.power == 1, head.lengthscale == 1). */
#define COVGRAM_COMPOSITE_MAX_TERMS 8
#define COVGRAM_COMPOSITE_MAX_FACTORS 8
typedef struct covgram_kernel_composite {
    covgram_kernel head;
    int32_t nterms;
    int32_t nfactors[COVGRAM_COMPOSITE_MAX_TERMS];
    covgram_kernel factors[COVGRAM_COMPOSITE_MAX_FACTORS];
} covgram_kernel_composite;

typedef struct covgram_ctx covgram_ctx;           /* one device + one stream + workspace + rocFFT plans */
typedef struct covgram_points covgram_points;     /* device-resident point set (stays resident across MVMs) */
typedef struct covgram_toeplitz covgram_toeplitz; /* cached circulant spectrum + plans */

int covgram_version(void);
/* sizeof() of the two structs that cross the ABI by pointer, as THIS build of the library sees them: a binding asserts its
 * own mirror against these at load time (julia/CovGram.jl does; tests/abi_layout.c checks every field offset). */
int covgram_sizeof_kernel(void);
int covgram_sizeof_composite(void);
const char* covgram_last_error(void);
int covgram_device_count(int* count);

/* hip_stream: the hipStream_t every kernel of this ctx is launched on (e.g. torch's current stream); NULL is the
 * device's default (null) stream.  The library never creates streams of its own. */
int covgram_ctx_create(covgram_ctx** ctx, int device_id, void* hip_stream);
int covgram_ctx_destroy(covgram_ctx* ctx);
int covgram_ctx_set_stream(covgram_ctx* ctx, void* hip_stream);
int covgram_ctx_get_stream(covgram_ctx* ctx, void** hip_stream);
/* tuning / A-B knobs: "dense_variant" (0 = auto: fp32 EQ runs on the matrix cores when the norm bound of dense_mfma.hip
 * holds and the plain dot-product Gramian is applied as X (Y' a), 1 = always the entry-by-entry direct kernel, 2 = matrix
 * cores whenever the shape allows), "rows_per_lane", "jsplit",
 * "target_wgs", "grad_keep_r", "time_kernels", "toeplitz_fused" (1 = fused row-FFT kernels, radix-16 stages at M' = 4096; 0 = rocFFT
 * batches; 2 = radix-4 stages everywhere), "toeplitz_colfft" (column FFT of the Toeplitz fast path: 16 = radix-16 register
 * butterflies, the default; 4 = the radix-4 LDS kernel), "toeplitz_persist" (fused radix-16 row kernel: persistent workgroups that prefetch the next row pair into registers; -1 = fp64 only, 0 = never, 1 = always), "toeplitz_real_spectrum" (read when a handle is created: 1 = a symmetric
 * matrix keeps the row kernel's spectrum copy as reals, 0 = always complex), "mfma_lds" (matrix-core EQ path: four waves share the
 * column tiles through LDS; -1 = when the column chunks are long enough, 0 = never, 1 = whenever compiled: d <= 8),
 * "matrix_variant" (covgram_matrix: 0 = rows in registers, 64-column strips — d <= 64 —, 1 = the generic entry-by-entry kernel),
 * "mfma_mrhs" (matrix right-hand sides on the fp32 matrix cores, dense_mfma_mrhs_kernel: -1 = from 5 columns (9 for the cheap profiles at d <= 3), 0 = never — four
 * columns at a time on the VALU —, 1 = from 2 columns),
 * "mfma_sym" (matrix-core EQ path on gramian(k, x), both sides the SAME device points: evaluate the upper triangle once;
 * -1 = from n = 12500 ... 18000 by the profile's cost, 0 = never, 1 = always),
 * "dense_sym" (the same for fp64 on the direct-difference path — the reference's default element type: gramian(k, x) with one
 * right-hand side evaluates every entry on or above the diagonal blocks once, dense_sym_kernel; -1 = from n = 6144 (16384 for
 * Cauchy / IMQ / Dot) while its column-sum slab of n^2 / 8 bytes stays within 1 GiB, 0 = never, 1 = always, up to a 2 GiB slab — the
 * slab lives in the ctx's workspace until the ctx is destroyed: 512 MiB at n = 65536, once per ctx),
 * "composite_termwise" (1 = a Sum runs one MVM per term on the term's own path, 0 = one pass of the composite kernels),
 * "sum_fused" (fp32 Sum of two or three single-profile isotropic terms — EQ, RQ, Cauchy, IMQ, MaternP(1..3), no Power wrapper — in ONE pass of
 * the matrix-core kernels, every term evaluated on the pair's shared distance as the reference does, src/algebra.jl:27-47: -1 = where it
 * measured faster than one MVM per term (three terms), 0 = never, 1 = wherever the one-pass kernels exist),
 * "mfma_sym_rt" (the symmetric fp32 matrix-core kernels at one or two MFMAs per tile — EQ, MaternP, RQ, Cauchy, IMQ: -1 / 2 = two row tiles per
 * wave in 4-wave workgroups, dense_mfma_sym2.hpp; 1 = one row tile per wave),
 * "grad_expand" (fp64 isotropic gradient / value-gradient Gramians in the expanded form — |x - y|^2 = |x|^2 + |y|^2 - 2 x.y with
 * cached norms, 4 instead of 6 fp64 instructions per dimension and pair: -1 = while the pre-scaled clouds lie within
 * gamma^2 R^2 <= 1000 of their common centre, 0 = never, 1 = always),
 * "inkernel_reduce" (the dense kernels' column-split partials: 1 = summed inside the kernel by the last workgroup of each row block
 * to arrive — fixed order, bit-identical to the separate launch, one launch and one dependent-launch gap less per MVM —, 0 = the separate
 * reduce launch, -1 = in the kernel only where it measured faster: the matrix-core EQ kernel up to n = 4096),
 * "grad_bcast" (fp64 expanded-form gradient / value-gradient MVM with the column records in vector registers, read by
 * v_fmac_f64_dpp row_newbcast — counted vector loads instead of the scalar stream: -1 = from padded d = 24, 0 = never, 1 / 4 = always,
 * with that many waves per workgroup),
 * "dense_bcast" (the same register-broadcast scheme for the fp64 dense value MVM of the isotropic kernels, one right-hand side:
 * |x - y|^2 expanded around cached norms, one v_fmac_f64_dpp per dimension and pair, all-entries and upper-triangle-once kernels:
 * -1 = from padded d = 16 up to d = 64 inside the gamma^2 R^2 <= 1000 gate of "grad_expand", 0 = never, 1 = wherever it applies, d >= 8),
 * "mfma_f16" (the fp32 matrix-core kernels' split of the coordinates: -1 / 1 = the fp16 two-way split — 3 products per coordinate, one MFMA per
 * FOUR coordinates, half the matrix-core work of the bf16 three-way split — while both clouds lie within g^2 R^2 <= 72 and the bf16 split beyond,
 * 0 = always the bf16 split, 2 = the fp16 split up to the matrix-core gate of 126: measurements only.  EQ kernels (general and symmetric) and,
 * since round 5, the generic ones — MaternP(p >= 1), RQ, Cauchy, IMQ, EQ^p, one-pass Sums; isotropic, d <= 30 — inside the same 72 / 126 of their gate),
 * "kron_fill" (1 / 2: workgroups per CU the Kronecker mode kernel's column tiling aims at; 2 measured slower, profiles/r05_kron_fill_ab.txt; 3 = never
 * fuse the last two modes: measurements, profiles/r05_kron_nopair_ab.txt),
 * "mfma_gate_pct" (1..100, default 100: both radius gates in percent.  BASELINE's 1e-5 is held NORM-wise at the full gates (<= 7.3e-6 measured);
 * the ROW-wise error |err_i| / (|K| |a|)_i of an adversarial cloud — points ON the gate's sphere, isolated rows, d >= 5 — reaches 1.5e-5 (bf16) /
 * 2.5e-5 (fp16) at the edge and scales with the gate: 40 holds 1e-5 row-wise on it.  Clouds outside run the direct-difference kernels),
 * "mfma_fuse_w" (the general matrix-core EQ kernel's column weights a_j exp2(f_j): -1 / 1 = formed inside the kernel, 0 = by a pack launch in front of
 * it — bit-identical results),
 * "mfma_stamp" (1 = the general matrix-core EQ kernel runs its clock-stamping DIAGNOSTIC build — s_memtime / s_memrealtime
 * around every workgroup's column loop, for bench.py's sustained-clock figure; never set in production). */
int covgram_ctx_set_option(covgram_ctx* ctx, const char* key, int64_t value);
/* read-only facts: "last_dense_path" (which kernel the last covgram_mvm ran: 0 none yet, 1 lane-per-row direct differences,
 * 2 matrix cores, 3 wide rows, 4 Gramian(Dot(), x, y) factored as X (Y' a)), "last_mfma_lds" (1: that matrix-core MVM shared its column tiles through LDS), "last_mfma_sym" (1: the last dense
 * MVM ran the symmetric upper-triangle kernel), "last_dense_sym" (1: it ran a direct-difference symmetric kernel, fp64 or fp32), "last_inkernel_reduce" (1: the last dense kernel summed its own split-J slab), "last_grad_expand" (1: the last gradient MVM ran the expanded form), "last_grad_bcast" (waves per workgroup of the broadcast kernel if the last gradient MVM ran it, else 0), "last_sum_fused" (1: the last covgram_mvm ran a Sum on the one-pass kernels), "last_mfma_instance" (which instance of the matrix-core EQ kernels the last launch was: the template arguments of dense_mfma_eq_kernel as K2 1e5 + RT 1e4 + WPB 1e3 + LDS 100 + STAMP 10 + FMT, -(K2 10 + FMT) for the symmetric kernel, 0 otherwise — bench.py checks its recorded PMC pass against it), "last_dense_bcast" (1: the last fp64 dense MVM ran a register-broadcast kernel), "last_mfma_f16" (1: the last general matrix-core EQ MVM ran the fp16 two-way split), "last_jsplit" (the column split of the last lane-per-row dense launch), "last_kron_path" (which kernels the last covgram_kron_mvm ran, as bits: 1 = the fused last-two-modes pass, 2 = the single-mode kernel, 4 = the last-mode kernel, 8 = a rocBLAS GEMM (a factor side >= 1024, >= 256 with >= 2 GFLOP, or a shape the kernels refuse), 16 = two small trailing factors multiplied out first), "num_cus", "last_clock_khz" (median shader clock over the workgroups of the last
 * launch made with "mfma_stamp" = 1; synchronises the stream; 0 = no stamped launch yet). */
int covgram_ctx_get_info(covgram_ctx* ctx, const char* key, int64_t* value);
int covgram_sync(covgram_ctx* ctx);
/* With option "time_kernels" = 1 every dense / gradient MVM brackets its dominant kernel with HIP events on the
 * ctx stream; this returns the summed device time and the number of launches since the last reset. */
int covgram_ctx_kernel_time(covgram_ctx* ctx, double* total_ms, int64_t* launches, int32_t reset);

/* x: n points of dimension d, point-major.  loc == HOST: copied to the device; loc == DEVICE: borrowed
 * (the caller keeps it alive and unchanged while the handle lives: creation caches a sample mean c of the set as the
 * common centre of the isotropic kernels and the extent max |x_i - c|^2 that gates the matrix-core path; after an
 * in-place update destroy and re-create the handle, as the Python Gramian does from the tensor's version counter). */
int covgram_points_create(covgram_ctx* ctx, covgram_points** out, const void* x, int64_t n, int32_t d,
                          int32_t dtype, int32_t loc);
/* rows [first, first+count) of an existing handle (row shard for multi-GPU); borrows the parent's memory. */
int covgram_points_slice(const covgram_points* parent, int64_t first, int64_t count, covgram_points** out);
int covgram_points_destroy(covgram_points* pts);
int covgram_points_info(const covgram_points* pts, int64_t* n, int32_t* d, int32_t* dtype);

/* y <- alpha * G(k; X, Y) * a + beta * y.   G is n×m (n = |X|, m = |Y|); a is m×nrhs (lda >= m),
 * y is n×nrhs (ldy >= n), column-major; dtype is that of the points.  loc applies to a and y.
 * Aliasing: y may be a itself (an in-place MVM with n == m: every dense path reads the weights from its own packed copy, or — the general
 * matrix-core EQ kernel, which reads a directly — falls back to the pack launch when the two ranges overlap); partially overlapping
 * columns of a multi-column a / y are not supported. */
int covgram_mvm(covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, const covgram_points* Y,
                const void* a, int64_t lda, void* y, int64_t ldy, int32_t nrhs, double alpha, double beta,
                int32_t loc);

/* out[i + j*ldo] = k(x_i, y_j): dense instantiation of the n×m Gramian (column-major). */
int covgram_matrix(covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, const covgram_points* Y,
                   void* out, int64_t ldo, int32_t loc);

/* Multi-GPU form of the symmetric dense MVM (one process per GPU, every rank holds all of x and a): rank r of `world`
 * evaluates the upper-triangle tiles of gramian(k, x) whose 256-row panel p satisfies p % world == r — cyclic, so every rank
 * gets the same share of the triangle — and returns in y (n scalars, device) the partial product of those entries AND their
 * mirror images; the partials of all ranks add up to G a, so ONE all-reduce (RCCL) completes b on every rank
 * (the rows of src/gramian.jl:81 are independent, and so are the unordered pairs {i, j}).
 * fp32: the symmetric matrix-core kernels where they apply (EQ / RQ / Cauchy / IMQ / MaternP(p >= 1) / Dot^p / ExponentialDot, d <= 32, norm
 * gate, n from 12500 ... 18000 by profile or option "mfma_sym" = 1).  Where NO matrix-core kernel takes the pair (Exponential,
 * gamma-exponential, MaternP(0), clouds outside the norm gate) and in fp64 (the reference's default element type): the direct-difference
 * symmetric kernels over the cyclic row blocks p % world == rank (64 rows in fp64, 64 R rows in fp32, R = 4 / 2 / 1 for d <= 8 / 32 / 64) —
 * any single profile without a Power wrapper, d <= 64, as long as one rank's column-sum slab, ceil(row blocks / world) x n scalars, stays
 * within 2 GiB (it lives in the ctx's workspace).  `*supported` of covgram_mvm_sym_supported says so for the given `world` (identically on
 * every rank: it depends on k, x and world only), and covgram_mvm_sym_partial returns COVGRAM_EUNSUPPORTED exactly when it says 0 —
 * callers then shard rows and all-gather (covgram_mvm). */
int covgram_mvm_sym_supported(covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, int32_t world, int32_t* supported);
int covgram_mvm_sym_partial(covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, const void* a, void* y,
                            int32_t rank, int32_t world);

/* ---- the collective behind the ABI (round 5) ------------------------------------------------------------------------------
 * One process per GPU; a ctx can own ONE RCCL communicator, and the multi-GPU MVMs below enqueue their single collective on the
 * ctx stream right behind the kernels (no second stream, no host synchronisation): the GPU analogue of the reference's only
 * parallel axis, `@threads for i in 1:n` over the output rows of mul! (src/gramian.jl:78-87).
 *   covgram_comm_unique_id   rank 0 calls it once and hands the COVGRAM_COMM_ID_BYTES bytes to the other ranks by any means of the
 *                            caller's (MPI.jl's bcast, a torch.distributed store, a file): nothing else crosses processes outside RCCL.
 *   covgram_comm_create      collective over all `world` ranks (ncclCommInitRank on the ctx's device).  RCCL is resolved at this first
 *                            use (dlopen of librccl.so.1 — in a PyTorch process the copy torch has loaded): single-GPU callers never load it.
 *   covgram_comm_destroy     also done by covgram_ctx_destroy.      covgram_comm_info: rank and world (world = 0: no communicator).
 *   covgram_comm_all_gather / covgram_comm_all_reduce_sum   the two collectives themselves on device buffers, stream-ordered (a block
 *                            Gramian's shards, a caller's own vectors); all_gather may run in place (send == recv + rank * count).
 *   covgram_mvm_sharded      y <- alpha G(k; X, Y) a + beta y with y COMPLETE ON EVERY RANK.  X (all n row points), Y, a and y are
 *                            replicated device data; rank r evaluates the rows [r per, (r + 1) per), per = ceil(n / world), with the
 *                            single-GPU kernels — straight into its slice of y when world divides n, the all-gather then in place — and
 *                            ONE ncclAllGather completes y, which is the replicated `a` of the next Krylov iteration.  One right-hand side.
 *   covgram_mvm_sym_allreduce  the same product of gramian(k, x) in the symmetric form: covgram_mvm_sym_partial(rank, world) + ONE
 *                            ncclAllReduce; COVGRAM_EUNSUPPORTED exactly when covgram_mvm_sym_supported says 0 (then: covgram_mvm_sharded). */
#define COVGRAM_COMM_ID_BYTES 128
int covgram_comm_unique_id(void* id, int64_t bytes);
int covgram_comm_create(covgram_ctx* ctx, const void* unique_id, int32_t rank, int32_t world);
int covgram_comm_destroy(covgram_ctx* ctx);
int covgram_comm_info(const covgram_ctx* ctx, int32_t* rank, int32_t* world);
int covgram_comm_all_gather(covgram_ctx* ctx, const void* send, void* recv, int64_t count, int32_t dtype);
int covgram_comm_all_reduce_sum(covgram_ctx* ctx, void* buf, int64_t count, int32_t dtype);
int covgram_mvm_sharded(covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, const covgram_points* Y, const void* a,
                        void* y, double alpha, double beta);
int covgram_mvm_sym_allreduce(covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, const void* a, void* y, double alpha,
                              double beta);

/* Gradient-kernel Gramian (nd × md): Y <- alpha * G A + beta * Y with flat point-major block vectors as the columns of
 * A (m*d × nrhs, lda >= m*d) and Y (n*d × nrhs, ldy >= n*d), column-major; a vector is nrhs = 1 (lda, ldy then unused beyond the check).
 * blockmul! takes vectors of matrices and the block mul! broadcasts over their columns (src/gramian.jl:241-257, src/gradient.jl:86-92):
 * two right-hand sides share one pass over the pairs (r, phi', phi'' evaluated once) where the lane-per-row kernel holds two
 * accumulators (fp64 d <= 32, fp32 d <= 64, single profiles without a Power wrapper); otherwise one pass per column. */
int covgram_grad_mvm(covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, const covgram_points* Y, const void* a,
                     int64_t lda, void* y, int64_t ldy, int32_t nrhs, double alpha, double beta, int32_t loc);

/* Value-and-gradient Gramian (n(d+1) × m(d+1)), src/gradient.jl:400-474 with the block mul! of :319-351: block (i,j) is
 *     [ k(x_i,y_j)        (d/dy k)^T      ]
 *     [ d/dx k            d/dx d/dy^T k   ]
 * flat point-major block vectors: entry i*(d+1) is the value component, i*(d+1)+1+l the l-th gradient component; right-hand sides
 * as for covgram_grad_mvm (lda >= m*(d+1), ldy >= n*(d+1)). */
int covgram_valgrad_mvm(covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, const covgram_points* Y, const void* a,
                        int64_t lda, void* y, int64_t ldy, int32_t nrhs, double alpha, double beta, int32_t loc);

/* Toeplitz T[i,j] = vc[i-j] (i >= j), vr[j-i] (i < j); vr == NULL: symmetric (vr = vc, m = n).
 * circulant != 0: T[i,j] = vc[(i-j) mod n] (vr must be NULL).  The spectrum of the circulant embedding
 * (N = next power of two >= n+m-1, real-to-complex rocFFT) is computed once here and cached. */
int covgram_toeplitz_create(covgram_ctx* ctx, covgram_toeplitz** out, const void* vc, const void* vr, int64_t n,
                            int64_t m, int32_t dtype, int32_t loc, int32_t circulant);
int covgram_toeplitz_mvm(covgram_toeplitz* T, const void* a, void* y, double alpha, double beta, int32_t loc);
int covgram_toeplitz_destroy(covgram_toeplitz* T);

/* Direct solvers of src/toeplitz.jl for the symmetric positive definite Toeplitz matrix K = SymmetricToeplitz([1; r])
 * (UNIT diagonal; r = the first column without its leading 1 — callers with T.vc[0] = r_0 != 1 pass vc[1:] / r_0 and scale the
 * result by 1 / r_0, src/toeplitz.jl:100-111 with its inverted `r_0 == 1` test put right):
 *   covgram_toeplitz_durbin    replaces durbin!(y, r)        src/toeplitz.jl:12-27    y = K_n \ (-r), K_n = SymmetricToeplitz([1, r[1:n-1]]), n = length(r)
 *   covgram_toeplitz_levinson  replaces levinson!(x, r, b)   src/toeplitz.jl:75-98    x = K \ b, n = length(b) = length(r) + 1
 *   covgram_toeplitz_trench    replaces trench!(B, r)        src/toeplitz.jl:52-71    B = inv(K), n x n column-major (ldb), BOTH triangles filled
 * Durbin / Levinson are chains of n - 1 dependent steps: one workgroup walks them in ONE launch that cannot be interrupted (n <= 16384: state
 * in registers and LDS, 28 ms at n = 16384 fp64; above: vectors in global memory, ~8 us a step — n = 65536: 0.5 s; it grows with n^2), so all
 * three return COVGRAM_EUNSUPPORTED above COVGRAM_TOEPLITZ_DIRECT_MAX_N: larger systems are
 * for the circulant-preconditioned CG over covgram_toeplitz_mvm (covgram/solve.py: toeplitz_solve; julia/CovGram.jl: `\`). */
#define COVGRAM_TOEPLITZ_DIRECT_MAX_N 65536
int covgram_toeplitz_durbin(covgram_ctx* ctx, const void* r, int64_t n, void* y, int32_t dtype, int32_t loc);
int covgram_toeplitz_levinson(covgram_ctx* ctx, const void* r, const void* b, int64_t n, void* x, int32_t dtype, int32_t loc);
int covgram_toeplitz_trench(covgram_ctx* ctx, const void* r, int64_t n, void* B, int64_t ldb, int32_t dtype, int32_t loc);

/* The vector work of ONE conjugate-gradient iteration, after the caller's Ap = A p (covgram_mvm & co.) — the recurrences of
 * IterativeSolvers.cg! 0.9.2, the reference's caller of mul! for `G \ b` (src/gramian.jl:229-238, src/lazy_linear_algebra.jl:135-144),
 * without preconditioner:   alpha = rho / (p . Ap);  x += alpha p;  r -= alpha Ap;  rho' = r . r;  p = r + (rho' / rho) p.
 * Three launches, all scalars on the device, sums in a fixed order.  Device pointers only.  scal: 2 + 512 elements of the vectors'
 * type; the caller sets scal[1] = r . r before the first step; after a step scal[1] = rho' (= |r|^2: the residual test) and
 * scal[0] = the rho it divided by.  The rest of scal is scratch. */
int covgram_cg_step(covgram_ctx* ctx, int64_t n, int32_t dtype, void* x, void* r, void* p, const void* Ap, void* scal);
/* The same step for A = G + Diagonal(diag) (the reference's G + sigma^2 I, kept lazy: src/gramian.jl:55-60, src/lazy_linear_algebra.jl:126-133)
 * after the caller's Ap = G p of the Gramian ALONE: the first launch completes Ap <- Ap + diag .* p in the pass that takes p . Ap, and the
 * last one leaves |r| = sqrt(rho') in scal[2 + 512] — the diagonal term and the residual norm cost no launch of their own (a graph-replayed
 * iteration at n = 16384 is launch-bound: tools/cg_rate.py).  diag: n device entries, or NULL (no shift).  scal: 2 + 512 + 1 elements. */
int covgram_cg_step_shifted(covgram_ctx* ctx, int64_t n, int32_t dtype, void* x, void* r, void* p, void* Ap, void* scal, const void* diag);

/* Y <- alpha * (F_1 ⊗ F_2 ⊗ ... ⊗ F_q) A + beta * Y, standard Kronecker order (F_1 = slowest index).
 * factors[i]: dense rows[i]×cols[i] column-major matrix with leading dimension lds[i] (device or host per loc).
 * A: (prod cols)×nrhs (lda), Y: (prod rows)×nrhs (ldy), column-major; a vector is nrhs = 1.  Hand-written mode-product kernels on the
 * matrix cores of the data's own precision (csrc/kron.hip): the last two modes in ONE pass when cols[q-1] <= 128, so q = 3 costs two
 * passes over the tensor; a mode whose factor has a side >= 1024 (a compute-bound dense GEMM) goes to rocBLAS. */
int covgram_kron_mvm(covgram_ctx* ctx, const void* const* factors, const int64_t* rows, const int64_t* cols,
                     const int64_t* lds, int32_t q, int32_t dtype, const void* a, int64_t lda, void* y, int64_t ldy,
                     int32_t nrhs, double alpha, double beta, int32_t loc);

/* Y <- alpha * U (V' A) + beta * Y;  U: n×r (ldu), V: m×r (ldv), A: m×nrhs (lda >= m), Y: n×nrhs (ldy >= n), column-major.
 * nrhs >= 8: both tall-skinny products run on the matrix cores in the data's own precision (v_mfma_f32_32x32x2_f32 /
 * v_mfma_f64_16x16x4_f64); fewer right-hand sides: streaming GEMV kernels, one pair per column. */
int covgram_lowrank_mvm(covgram_ctx* ctx, const void* U, int64_t ldu, const void* V, int64_t ldv, int64_t n, int64_t m,
                        int64_t r, int32_t dtype, const void* a, int64_t lda, void* y, int64_t ldy, int32_t nrhs,
                        double alpha, double beta, int32_t loc);

/* Test hook: the double-precision parameter block handed to the device kernels for `k`
 * (out45[0..8] = gamma, gamma^2, scale, param, c0, 2p+1, Taylor bound, d1, d2; then the MaternP tables
 * H_p, H_{p-1}, H_{p-2} and Taylor coefficients, 9 doubles each).  Needs no device. */
int covgram_debug_kernel_params(const covgram_kernel* k, int32_t dtype, int32_t for_gradient, double* out45);

#ifdef __cplusplus
}
#endif
#endif /* COVGRAM_H */
