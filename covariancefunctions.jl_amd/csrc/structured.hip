// structured.hip — Kronecker and low-rank MVMs.
//
// Kronecker: gramian(::SeparableProduct, ::LazyGrid, ::LazyGrid) = kronecker(G_1, ..., G_q)
// (src/algebra.jl:91-95) and kronecker(G) = gramian(k.k, x, y) ⊗ B for SeparableKernel
// (src/separable.jl:33-42); the MVM is KroneckerProducts 1.1.1 (third party, source not in the
// reference tree) — restated from the identity it implements: with a viewed as a c_1×...×c_q tensor
// (first factor = slowest index), (F_1 ⊗ ... ⊗ F_q) a = a ×_1 F_1 ×_2 F_2 ... ×_q F_q (mode products).
// Each mode product is a batched small GEMM  Out[b] (r_k × post) = F_k (r_k × c_k) · T[b] (c_k × post): rocBLAS.
//
// Low rank: gramian(k::FiniteBasis, x, y) = LazyMatrixProduct(U, V') (src/mercer.jl:61-70) whose
// mul! applies the factors right to left (src/lazy_linear_algebra.jl:78-85): y = α U (Vᵀ a) + β y.
// For a vector right-hand side both products are GEMVs, i.e. HBM-streaming of U and V.
#include <algorithm>

#include <rocblas/rocblas.h>

#include "common.hpp"

namespace covgram {

// ---- low rank: y = alpha U (V^T a) + beta y, two HBM-streaming passes --------------------------------------------------
template <typename T, int VEC> struct VecOf;
template <> struct VecOf<float, 4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct VecOf<double, 2> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct VecOf<float, 1> { typedef float type; };
template <> struct VecOf<double, 1> { typedef double type; };

template <typename T, int VEC>
__device__ __forceinline__ T vdot(typename VecOf<T, VEC>::type u, typename VecOf<T, VEC>::type w, T acc) {
    if constexpr (VEC == 1) return __builtin_fma(u, w, acc);
    else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc = __builtin_fma(u[e], w[e], acc);
        return acc;
    }
}

constexpr int LR_RC = 32;   // columns of V carried per pass over a row slab (accumulators per thread)

// zpart[slab][k] = sum_{j in slab} V[j + k*ldv] a[j].  One workgroup per row slab; a thread streams VEC rows of up to 32
// columns at a time (16-byte loads, `a` read once per 32 columns instead of once per column), then a deterministic
// shuffle + LDS reduction per column.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void lowrank_vta_kernel(const T* __restrict__ V, int64_t ldv, int64_t m, int64_t r, const T* __restrict__ a,
                                                          T* __restrict__ zpart, int64_t per) {
    using VT = typename VecOf<T, VEC>::type;
    const int64_t j0 = (int64_t)blockIdx.x * per, j1 = (j0 + per < m) ? (j0 + per) : m;
    __shared__ T red[LR_RC][256 + 1];                           // [column][thread] (+1: the column sums read down a row)
    for (int64_t c0 = 0; c0 < r; c0 += LR_RC) {
        const int nc = (int)((r - c0 < LR_RC) ? (r - c0) : LR_RC);
        T acc[LR_RC];
#pragma unroll
        for (int c = 0; c < LR_RC; ++c) acc[c] = (T)0;
        for (int64_t j = j0 + (int64_t)threadIdx.x * VEC; j < j1; j += 256 * VEC) {
            if (j + VEC <= j1) {
                const VT av = *reinterpret_cast<const VT*>(a + j);
#pragma unroll
                for (int c = 0; c < LR_RC; ++c)
                    if (c < nc) acc[c] = vdot<T, VEC>(*reinterpret_cast<const VT*>(V + (c0 + c) * ldv + j), av, acc[c]);
            } else {                                            // ragged end of the matrix
                for (int64_t jj = j; jj < j1; ++jj)
#pragma unroll
                    for (int c = 0; c < LR_RC; ++c)
                        if (c < nc) acc[c] = __builtin_fma(V[(c0 + c) * ldv + jj], a[jj], acc[c]);
            }
        }
        // block reduction through LDS: thread t sums 32 of column (t / 8)'s 256 partials, then 8 lanes combine (fixed order)
#pragma unroll
        for (int c = 0; c < LR_RC; ++c) red[c][threadIdx.x] = acc[c];
        __syncthreads();
        {
            const int c = threadIdx.x >> 3, part = threadIdx.x & 7;
            T s = (T)0;
#pragma unroll
            for (int e = 0; e < 32; ++e) s += red[c][part * 32 + e];
            s += __shfl_down(s, 4, 8); s += __shfl_down(s, 2, 8); s += __shfl_down(s, 1, 8);
            if (part == 0 && c < nc) zpart[(int64_t)blockIdx.x * r + c0 + c] = s;
        }
        __syncthreads();
    }
}

// z[k] = sum_slab zpart[slab][k] (fixed order): one workgroup per 32 columns, 8 slab subsets in parallel, 4 loads in flight each
template <typename T>
__global__ __launch_bounds__(256) void lowrank_zsum_kernel(const T* __restrict__ zpart, int64_t nslab, int64_t r, T* __restrict__ z) {
    const int kk = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int64_t k = (int64_t)blockIdx.x * 32 + kk;
    __shared__ T red[8][32];
    T s0 = (T)0, s1 = (T)0, s2 = (T)0, s3 = (T)0;
    if (k < r) {
        int64_t sl = part;
        for (; sl + 24 < nslab; sl += 32) {
            s0 += zpart[sl * r + k]; s1 += zpart[(sl + 8) * r + k]; s2 += zpart[(sl + 16) * r + k]; s3 += zpart[(sl + 24) * r + k];
        }
        for (; sl < nslab; sl += 8) s0 += zpart[sl * r + k];
    }
    red[part][kk] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (part == 0 && k < r) {
        T s = (T)0;
#pragma unroll
        for (int p = 0; p < 8; ++p) s += red[p][kk];
        z[k] = s;
    }
}

// y[i] = alpha * sum_k U[i + k*ldu] z[k] + beta * y[i]; VEC rows per thread, z in LDS
template <typename T, int VEC>
__global__ __launch_bounds__(256) void lowrank_uz_kernel(const T* __restrict__ U, int64_t ldu, int64_t n, int64_t r, const T* __restrict__ zg,
                                                         T* __restrict__ y, T alpha, T beta) {
    using VT = typename VecOf<T, VEC>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* z = reinterpret_cast<T*>(smem);
    for (int64_t k = threadIdx.x; k < r; k += 256) z[k] = zg[k];
    __syncthreads();
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (i >= n) return;
    if (i + VEC <= n) {
        VT s = (VT)0;
#pragma unroll 8
        for (int64_t k = 0; k < r; ++k) s += *reinterpret_cast<const VT*>(U + k * ldu + i) * z[k];
        VT v = alpha * s;
        if (beta != (T)0) v += beta * *reinterpret_cast<const VT*>(y + i);
        *reinterpret_cast<VT*>(y + i) = v;
    } else {
        for (int64_t ii = i; ii < n; ++ii) {
            T s = (T)0;
            for (int64_t k = 0; k < r; ++k) s = __builtin_fma(U[ii + k * ldu], z[k], s);
            T v = alpha * s;
            if (beta != (T)0) v = __builtin_fma(beta, y[ii], v);
            y[ii] = v;
        }
    }
}

template <typename T, int VEC>
static void lowrank_launch(covgram_ctx* ctx, const T* U, int64_t ldu, const T* V, int64_t ldv, int64_t n, int64_t m, int64_t r, const T* a,
                           T* y, T alpha, T beta, T* zpart, T* z, int64_t nslab, int64_t per) {
    hipLaunchKernelGGL((lowrank_vta_kernel<T, VEC>), dim3((unsigned)nslab), dim3(256), 0, ctx->stream, V, ldv, m, r, a, zpart, per);
    hipLaunchKernelGGL(lowrank_zsum_kernel<T>, dim3((unsigned)((r + 31) / 32)), dim3(256), 0, ctx->stream, (const T*)zpart, nslab, r, z);
    const int64_t rows_per_block = 256 * VEC;
    hipLaunchKernelGGL((lowrank_uz_kernel<T, VEC>), dim3((unsigned)((n + rows_per_block - 1) / rows_per_block)), dim3(256), (size_t)r * sizeof(T),
                       ctx->stream, U, ldu, n, r, (const T*)z, y, alpha, beta);
}

template <typename T>
static void lowrank_run(covgram_ctx* ctx, const void* U, int64_t ldu, const void* V, int64_t ldv, int64_t n, int64_t m, int64_t r, const void* a,
                        void* y, double alpha, double beta, void* zpart, void* z, int64_t nslab, int64_t per) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const bool aligned = (((uintptr_t)U | (uintptr_t)V | (uintptr_t)a | (uintptr_t)y) % 16 == 0) && ldu % VEC == 0 && ldv % VEC == 0 && per % VEC == 0;
    if (aligned) lowrank_launch<T, VEC>(ctx, (const T*)U, ldu, (const T*)V, ldv, n, m, r, (const T*)a, (T*)y, (T)alpha, (T)beta, (T*)zpart, (T*)z, nslab, per);
    else lowrank_launch<T, 1>(ctx, (const T*)U, ldu, (const T*)V, ldv, n, m, r, (const T*)a, (T*)y, (T)alpha, (T)beta, (T*)zpart, (T*)z, nslab, per);
}

// rocBLAS handle of a ctx (created on first use; every call re-binds the ctx stream)
static int blas_handle(covgram_ctx* ctx, rocblas_handle* out) {
    if (!ctx->blas) {
        rocblas_handle h = nullptr;
        if (rocblas_create_handle(&h) != rocblas_status_success) { set_error("rocblas_create_handle failed"); return COVGRAM_EHIP; }
        rocblas_set_pointer_mode(h, rocblas_pointer_mode_host);
        ctx->blas = h;
    }
    *out = (rocblas_handle)ctx->blas;
    if (rocblas_set_stream(*out, ctx->stream) != rocblas_status_success) { set_error("rocblas_set_stream failed"); return COVGRAM_EHIP; }
    return COVGRAM_OK;
}

void ctx_blas_destroy(covgram_ctx* ctx) {
    if (ctx->blas) { (void)rocblas_destroy_handle((rocblas_handle)ctx->blas); ctx->blas = nullptr; }
}

static rocblas_status gemm_sb(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int64_t m, int64_t n, int64_t k, float alpha,
                              const float* A, int64_t lda, int64_t sa, const float* B, int64_t ldb, int64_t sb, float beta, float* C, int64_t ldc,
                              int64_t sc, int64_t batch) {
    return rocblas_sgemm_strided_batched(h, ta, tb, (rocblas_int)m, (rocblas_int)n, (rocblas_int)k, &alpha, A, (rocblas_int)lda, sa, B,
                                         (rocblas_int)ldb, sb, &beta, C, (rocblas_int)ldc, sc, (rocblas_int)batch);
}
static rocblas_status gemm_sb(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int64_t m, int64_t n, int64_t k, double alpha,
                              const double* A, int64_t lda, int64_t sa, const double* B, int64_t ldb, int64_t sb, double beta, double* C,
                              int64_t ldc, int64_t sc, int64_t batch) {
    return rocblas_dgemm_strided_batched(h, ta, tb, (rocblas_int)m, (rocblas_int)n, (rocblas_int)k, &alpha, A, (rocblas_int)lda, sa, B,
                                         (rocblas_int)ldb, sb, &beta, C, (rocblas_int)ldc, sc, (rocblas_int)batch);
}

// (F_1 (x) ... (x) F_q) a by successive mode products.  Each one is a plain strided-batched GEMM — the tensor viewed as
// [pre][K][post] with `post` contiguous is, per `pre`, a column-major (post x K) matrix S, and the product is S F_k^T — so it
// goes to rocBLAS (the first version used a hand-written 64x64 register-tiled kernel: 47 us per mode on the README's 128^3
// fp64 case, 15 % of the fp64 peak).  The last mode applies alpha / beta and writes y directly.
template <typename T>
static int kron_run(covgram_ctx* ctx, const void* const* factors, const int64_t* rows, const int64_t* cols, const int64_t* lds,
                    int q, const T* a_dev, T* y_dev, T alpha, T beta, T* bufA, T* bufB) {
    rocblas_handle h;
    int rc = blas_handle(ctx, &h);
    if (rc) return rc;
    const T* src = a_dev;
    T* dst = bufA;
    for (int k = 0; k < q; ++k) {
        int64_t pre = 1, post = 1;
        for (int i = 0; i < k; ++i) pre *= rows[i];
        for (int i = k + 1; i < q; ++i) post *= cols[i];
        const int64_t M = rows[k], K = cols[k];
        const bool last = (k == q - 1);
        T* out = last ? y_dev : dst;
        const T al = last ? alpha : (T)1, be = last ? beta : (T)0;
        rocblas_status st;
        if (post == 1)      // [pre][K] -> [pre][M]: one GEMM  out (M x pre) = F (M x K) * src (K x pre)
            st = gemm_sb(h, rocblas_operation_none, rocblas_operation_none, M, pre, K, al, (const T*)factors[k], lds[k], 0, src, K, 0, be, out, M, 0, 1);
        else                // per pre: out (post x M) = S (post x K) * F^T
            st = gemm_sb(h, rocblas_operation_none, rocblas_operation_transpose, post, M, K, al, src, post, K * post, (const T*)factors[k], lds[k], 0,
                         be, out, post, M * post, pre);
        if (st != rocblas_status_success) { set_error("kron_mvm: rocBLAS gemm failed with status %d (mode %d)", (int)st, k); return COVGRAM_EHIP; }
        src = dst;
        dst = (dst == bufA) ? bufB : bufA;
    }
    return COVGRAM_OK;
}

}  // namespace covgram

using namespace covgram;

extern "C" {

int covgram_kron_mvm(covgram_ctx* ctx, const void* const* factors, const int64_t* rows, const int64_t* cols, const int64_t* lds,
                     int32_t q, int32_t dtype, const void* a, void* y, double alpha, double beta, int32_t loc) {
    CG_REQUIRE(ctx && factors && rows && cols && lds && a && y, COVGRAM_EINVAL, "NULL argument");
    CG_REQUIRE(q >= 1 && q <= 16, COVGRAM_EINVAL, "kron: need 1 <= q <= 16 factors");
    CG_REQUIRE(dtype == COVGRAM_F32 || dtype == COVGRAM_F64, COVGRAM_EINVAL, "unknown dtype %d", dtype);
    const size_t ts = dtype_size(dtype);
    int64_t nin = 1, nout = 1, maxel = 1;
    for (int i = 0; i < q; ++i) {
        CG_REQUIRE(rows[i] >= 1 && cols[i] >= 1 && lds[i] >= rows[i], COVGRAM_EINVAL, "kron: bad factor %d shape", i);
        nin *= cols[i]; nout *= rows[i];
    }
    {   // largest intermediate tensor
        int64_t cur = nin;
        maxel = cur;
        for (int k = 0; k < q; ++k) { cur = cur / cols[k] * rows[k]; maxel = std::max(maxel, cur); }
    }
    CG_CHECK_HIP(hipSetDevice(ctx->device));
    // workspace: [bufA | bufB | staged a | staged y | staged factors]
    size_t fbytes = 0;
    if (loc == COVGRAM_HOST) for (int i = 0; i < q; ++i) fbytes += (size_t)rows[i] * cols[i] * ts;
    const size_t need = 2 * (size_t)maxel * ts + ((loc == COVGRAM_HOST) ? ((size_t)nin + nout) * ts + fbytes : 0) + 1024;
    void* w; int rc = ws_reserve(ctx, 1, need, &w); if (rc) return rc;
    char* base = (char*)w;
    void* bufA = base; void* bufB = base + (size_t)maxel * ts;
    const void* a_dev = a; void* y_dev = y;
    const void* fdev[16]; int64_t ldd[16];
    for (int i = 0; i < q; ++i) { fdev[i] = factors[i]; ldd[i] = lds[i]; }
    if (loc == COVGRAM_HOST) {
        char* p = base + 2 * (size_t)maxel * ts;
        CG_CHECK_HIP(hipMemcpyAsync(p, a, (size_t)nin * ts, hipMemcpyHostToDevice, ctx->stream)); a_dev = p; p += (size_t)nin * ts;
        if (beta != 0.0) CG_CHECK_HIP(hipMemcpyAsync(p, y, (size_t)nout * ts, hipMemcpyHostToDevice, ctx->stream));
        y_dev = p; p += (size_t)nout * ts;
        for (int i = 0; i < q; ++i) {
            CG_CHECK_HIP(hipMemcpy2DAsync(p, (size_t)rows[i] * ts, factors[i], (size_t)lds[i] * ts, (size_t)rows[i] * ts, cols[i], hipMemcpyHostToDevice, ctx->stream));
            fdev[i] = p; ldd[i] = rows[i]; p += (size_t)rows[i] * cols[i] * ts;
        }
    }
    if (dtype == COVGRAM_F32) rc = kron_run<float>(ctx, fdev, rows, cols, ldd, q, (const float*)a_dev, (float*)y_dev, (float)alpha, (float)beta, (float*)bufA, (float*)bufB);
    else rc = kron_run<double>(ctx, fdev, rows, cols, ldd, q, (const double*)a_dev, (double*)y_dev, alpha, beta, (double*)bufA, (double*)bufB);
    if (rc) return rc;
    if (loc == COVGRAM_HOST) {
        CG_CHECK_HIP(hipMemcpyAsync(y, y_dev, (size_t)nout * ts, hipMemcpyDeviceToHost, ctx->stream));
        CG_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return COVGRAM_OK;
}

int covgram_lowrank_mvm(covgram_ctx* ctx, const void* U, int64_t ldu, const void* V, int64_t ldv, int64_t n, int64_t m, int64_t r,
                        int32_t dtype, const void* a, void* y, double alpha, double beta, int32_t loc) {
    CG_REQUIRE(ctx && U && V && a && y, COVGRAM_EINVAL, "NULL argument");
    CG_REQUIRE(n >= 1 && m >= 1 && r >= 1 && ldu >= n && ldv >= m, COVGRAM_EINVAL, "lowrank: bad shape");
    CG_REQUIRE(dtype == COVGRAM_F32 || dtype == COVGRAM_F64, COVGRAM_EINVAL, "unknown dtype %d", dtype);
    CG_REQUIRE(r <= 8192, COVGRAM_EUNSUPPORTED, "lowrank: r = %lld exceeds 8192", (long long)r);
    const size_t ts = dtype_size(dtype);
    CG_CHECK_HIP(hipSetDevice(ctx->device));
    // row slabs of V: ~4 workgroups per CU, slab length a multiple of one sweep of the block (256 threads x 16 bytes)
    const int64_t sweep = 256 * (16 / (int64_t)ts);
    int64_t per = (m + (int64_t)ctx->num_cus * 4 - 1) / ((int64_t)ctx->num_cus * 4);
    per = std::max<int64_t>(sweep, ((per + sweep - 1) / sweep) * sweep);
    const int64_t nslab = (m + per - 1) / per;
    size_t need = ((size_t)nslab * r + (size_t)r) * ts + 512;
    if (loc == COVGRAM_HOST) need += ((size_t)n * r + (size_t)m * r + m + n) * ts + 64;
    void* w; int rc = ws_reserve(ctx, 1, need, &w); if (rc) return rc;
    char* p = (char*)w;
    void* zpart = p; p += (((size_t)nslab * r * ts) + 255) & ~(size_t)255;
    void* z = p; p += (((size_t)r * ts) + 255) & ~(size_t)255;
    const void *Ud = U, *Vd = V, *ad = a; void* yd = y; int64_t ldud = ldu, ldvd = ldv;
    if (loc == COVGRAM_HOST) {
        CG_CHECK_HIP(hipMemcpy2DAsync(p, (size_t)n * ts, U, (size_t)ldu * ts, (size_t)n * ts, r, hipMemcpyHostToDevice, ctx->stream)); Ud = p; ldud = n; p += (size_t)n * r * ts;
        CG_CHECK_HIP(hipMemcpy2DAsync(p, (size_t)m * ts, V, (size_t)ldv * ts, (size_t)m * ts, r, hipMemcpyHostToDevice, ctx->stream)); Vd = p; ldvd = m; p += (size_t)m * r * ts;
        CG_CHECK_HIP(hipMemcpyAsync(p, a, (size_t)m * ts, hipMemcpyHostToDevice, ctx->stream)); ad = p; p += (size_t)m * ts;
        if (beta != 0.0) CG_CHECK_HIP(hipMemcpyAsync(p, y, (size_t)n * ts, hipMemcpyHostToDevice, ctx->stream));
        yd = p;
    }
    if (dtype == COVGRAM_F32) lowrank_run<float>(ctx, Ud, ldud, Vd, ldvd, n, m, r, ad, yd, alpha, beta, zpart, z, nslab, per);
    else lowrank_run<double>(ctx, Ud, ldud, Vd, ldvd, n, m, r, ad, yd, alpha, beta, zpart, z, nslab, per);
    CG_CHECK_HIP(hipGetLastError());
    if (loc == COVGRAM_HOST) {
        CG_CHECK_HIP(hipMemcpyAsync(y, yd, (size_t)n * ts, hipMemcpyDeviceToHost, ctx->stream));
        CG_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return COVGRAM_OK;
}

}  // extern "C"
