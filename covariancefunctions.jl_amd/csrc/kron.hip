// kron.hip — Kronecker MVMs  y <- alpha (F_1 (x) ... (x) F_q) a + beta y: planner and C entry point.  The mode-product kernels
// (matrix cores of the data's own precision, csrc/kron_kernels.hpp) are instantiated in kron_f32.hip / kron_f64.hip.
//
// Reference: gramian(::SeparableProduct, ::LazyGrid, ::LazyGrid) = kronecker(G_1, ..., G_q) (src/algebra.jl:91-95) and
// kronecker(G) = gramian(k.k, x, y) (x) B for a SeparableKernel (src/separable.jl:33-42); README.md:205-210 is the timed case.
// Plan: the last two modes in ONE pass (kron_pair_kernel) when c_q <= 128, the other modes one pass each (kron_mode_kernel),
// the last mode alone through kron_modet_kernel; a mode whose factor has a side of 1024 and more is a compute-bound dense GEMM
// and goes to rocBLAS (a plain library GEMM).
#include <algorithm>

#include <rocblas/rocblas.h>

#include "common.hpp"
#include "kron_limits.hpp"

namespace covgram {
namespace kron {

// kron_f32.hip / kron_f64.hip
template <typename T>
int run_pair(covgram_ctx* ctx, const T* in, T* out, const T* F2, int64_t ld2, int64_t M1, int64_t K1, const T* F3, int64_t ld3, int64_t N2,
             int64_t K2, int64_t pre, T alpha, T beta);
template <typename T>
int run_mode(covgram_ctx* ctx, const T* in, T* out, const T* F, int64_t ld, int64_t M, int64_t K, int64_t pre, int64_t post, T alpha, T beta);
template <typename T>
int run_modet(covgram_ctx* ctx, const T* in, T* out, const T* F, int64_t ld, int64_t M, int64_t K, int64_t pre, T alpha, T beta);
extern template int run_pair<float>(covgram_ctx*, const float*, float*, const float*, int64_t, int64_t, int64_t, const float*, int64_t, int64_t, int64_t, int64_t, float, float);
extern template int run_pair<double>(covgram_ctx*, const double*, double*, const double*, int64_t, int64_t, int64_t, const double*, int64_t, int64_t, int64_t, int64_t, double, double);
extern template int run_mode<float>(covgram_ctx*, const float*, float*, const float*, int64_t, int64_t, int64_t, int64_t, int64_t, float, float);
extern template int run_mode<double>(covgram_ctx*, const double*, double*, const double*, int64_t, int64_t, int64_t, int64_t, int64_t, double, double);
extern template int run_modet<float>(covgram_ctx*, const float*, float*, const float*, int64_t, int64_t, int64_t, int64_t, float, float);
extern template int run_modet<double>(covgram_ctx*, const double*, double*, const double*, int64_t, int64_t, int64_t, int64_t, double, double);

// ---------------------------------------------------------------------------------------------------------------------------
// rocBLAS for the compute-bound modes (a factor side of 1024 and more)
// ---------------------------------------------------------------------------------------------------------------------------
static int blas_handle(covgram_ctx* ctx, rocblas_handle* out) {
    if (!ctx->blas) {
        // rocblas_create_handle allocates and synchronises: not legal inside a stream capture.  A graph user runs one
        // eager MVM of the shape first (as for every workspace of this library); inside a capture the call is refused, not crashed.
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (ctx->stream && hipStreamIsCapturing(ctx->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
            set_error("kron_mvm: the first library-GEMM mode on this ctx must run outside stream capture (it creates the rocBLAS handle)");
            return COVGRAM_EUNSUPPORTED;
        }
        rocblas_handle h = nullptr;
        if (rocblas_create_handle(&h) != rocblas_status_success) { set_error("rocblas_create_handle failed"); return COVGRAM_EHIP; }
        rocblas_set_pointer_mode(h, rocblas_pointer_mode_host);
        ctx->blas = h;
    }
    *out = (rocblas_handle)ctx->blas;
    if (rocblas_set_stream(*out, ctx->stream) != rocblas_status_success) { set_error("rocblas_set_stream failed"); return COVGRAM_EHIP; }
    return COVGRAM_OK;
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

// one mode as a library GEMM: the tensor seen as [pre][K][post] is, per leading index, a column-major post x K matrix S and the
// product is S F^T; post == 1: one GEMM out (M x pre) = F (M x K) in (K x pre)
template <typename T>
static int run_blas(covgram_ctx* ctx, const T* in, T* out, const T* F, int64_t ld, int64_t M, int64_t K, int64_t pre, int64_t post, T alpha, T beta) {
    // rocblas_int is 32 bits: every extent, leading dimension and the batch count must fit (they were cast unchecked)
    const int64_t lim = ((int64_t)1 << 31) - 1;
    CG_REQUIRE(M <= lim && K <= lim && pre <= lim && post <= lim && ld <= lim, COVGRAM_EUNSUPPORTED,
               "kron: a mode of extents (%lld, %lld, pre %lld, post %lld) exceeds the library GEMM's 32-bit sizes", (long long)M, (long long)K, (long long)pre, (long long)post);
    rocblas_handle h;
    int rc = blas_handle(ctx, &h);
    if (rc) return rc;
    ctx->last_kron_path |= 8;
    rocblas_status st;
    if (post == 1) st = gemm_sb(h, rocblas_operation_none, rocblas_operation_none, M, pre, K, alpha, F, ld, 0, in, K, 0, beta, out, M, 0, 1);
    else st = gemm_sb(h, rocblas_operation_none, rocblas_operation_transpose, post, M, K, alpha, in, post, K * post, F, ld, 0, beta, out, post, M * post, pre);
    if (st != rocblas_status_success) { set_error("kron_mvm: rocBLAS gemm failed with status %d", (int)st); return COVGRAM_EHIP; }
    return COVGRAM_OK;
}

// The explicit Kronecker product of two SMALL factors, column-major: P[(i1 r2 + i2) + (j1 c2 + j2) R] = F1[i1, j1] F2[i2, j2].  Two trailing modes
// of 16 x 16 factors are two launch-bound passes over the tensor (16^5: 30 + 11 us); as ONE 256 x 256 factor they are one pass of the last-mode
// kernel, whose extra flops are free at these sizes.
constexpr int64_t MERGE_MAX_SIDE = 256;
template <typename T>
__global__ __launch_bounds__(256) void kron_factor_kernel(const T* __restrict__ F1, int64_t ld1, int r1, int c1, const T* __restrict__ F2, int64_t ld2, int r2, int c2,
                                                          T* __restrict__ P) {
    const int R = r1 * r2, C = c1 * c2;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= R * C) return;
    const int i = e % R, j = e / R;
    P[e] = F1[(i / r2) + (int64_t)(j / c2) * ld1] * F2[(i % r2) + (int64_t)(j % c2) * ld2];
}

constexpr int64_t BLAS_MIN_SIDE = 1024;       // a factor side from which a mode is always a library GEMM
constexpr int64_t BLAS_MID_SIDE = 256;        // ... and from this side on when the mode has BLAS_MID_FLOPS of work: compute-bound, where a
constexpr double BLAS_MID_FLOPS = 2.0e9;      // register-blocked GEMM wins (256^3 fp64: 547 us on the kernels here, 471 us on rocBLAS)

// (F_1 (x) ... (x) F_q) applied to `batch` tensors that lie one after the other (the right-hand sides)
template <typename T>
static int kron_run(covgram_ctx* ctx, const void* const* factors, const int64_t* rows, const int64_t* cols, const int64_t* lds, int q, int64_t batch,
                    const T* a_dev, T* y_dev, T alpha, T beta, T* bufA, T* bufB, T* merged) {
    int64_t cur[16];
    for (int i = 0; i < q; ++i) cur[i] = cols[i];
    if (merged != nullptr) ctx->last_kron_path = 0;      // (the outer call; the recursion below keeps what it has)
    const T* src = a_dev;
    T* dst = bufA;
    auto next_out = [&](bool final) -> T* { return final ? y_dev : dst; };
    auto advance = [&]() { src = dst; dst = (dst == bufA) ? bufB : bufA; };
    double total_in = (double)batch;
    for (int i = 0; i < q; ++i) total_in *= (double)cols[i];
    auto big = [&](int k) {
        const int64_t side = std::max(rows[k], cols[k]);
        return side >= BLAS_MIN_SIDE || (side >= BLAS_MID_SIDE && 2.0 * total_in * (double)rows[k] >= BLAS_MID_FLOPS);
    };
    // the last two modes fused when the slab's rows fit the accumulators and there are enough slabs to fill the chip
    int64_t pre2 = batch;
    for (int i = 0; i + 2 < q; ++i) pre2 *= rows[i];     // leading extent once the other modes are done
    int64_t pre2_first = batch;
    for (int i = 0; i + 2 < q; ++i) pre2_first *= cols[i];
    bool pair = q >= 2 && !big(q - 1) && !big(q - 2) && pair_ok(cols[q - 2], cols[q - 1], lds[q - 2], lds[q - 1]);
    // the pair first when it shrinks the tensor, last otherwise (fewer bytes through the other modes)
    const bool pair_first = pair && q > 2 && rows[q - 2] * rows[q - 1] < cols[q - 2] * cols[q - 1];
    if (ctx->kron_fill == 3) pair = false;               // (measurements: every mode on its own kernel)
    // fp32 slabs up to 64 x 64: the fused pass is a latency chain of eight barrier-separated steps per 16 KB slab (64^4: 225 us for what the two
    // single-mode kernels do in 2 x 51; tools/kron_fill_ab.py nopair: 64^4 304 -> 170 us, 48^4 152 -> 80) — in fp64 the fused pass still wins (271 / 311)
    if (sizeof(T) == 4 && q >= 2 && std::max(std::max(rows[q - 2], cols[q - 2]), std::max(rows[q - 1], cols[q - 1])) <= 64) pair = false;
    if (pair) {
        const int64_t units = (pair_first ? pre2_first : pre2) * ((rows[q - 2] + 63) / 64);   // workgroups of the fused pass
        if (units < ctx->num_cus / 2) pair = false;   // a handful of slabs: the two modes one after the other spread wider
        // small slabs leave most of the fused pass's eight waves idle (16^5: 104 us fused against 2 x 7 us mode by mode)
        if (rows[q - 2] < PAIR_MIN_SIDE || cols[q - 1] < PAIR_MIN_SIDE) pair = false;
    }
    int rc;
    // two small trailing factors the fused pass does not take: multiply them out (<= 256 x 256) and run ONE last-mode pass
    if (!pair && q >= 2 && rows[q - 2] * rows[q - 1] <= MERGE_MAX_SIDE && cols[q - 2] * cols[q - 1] <= MERGE_MAX_SIDE && merged != nullptr) {
        const int r1 = (int)rows[q - 2], c1 = (int)cols[q - 2], r2 = (int)rows[q - 1], c2 = (int)cols[q - 1];
        const int tot = r1 * r2 * c1 * c2;
        ctx->last_kron_path |= 16;
        hipLaunchKernelGGL(kron_factor_kernel<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, (const T*)factors[q - 2], lds[q - 2], r1, c1,
                           (const T*)factors[q - 1], lds[q - 1], r2, c2, merged);
        const void* f2[16]; int64_t rows2[16], cols2[16], lds2[16];
        for (int i = 0; i + 2 < q; ++i) { f2[i] = factors[i]; rows2[i] = rows[i]; cols2[i] = cols[i]; lds2[i] = lds[i]; }
        f2[q - 2] = merged; rows2[q - 2] = (int64_t)r1 * r2; cols2[q - 2] = (int64_t)c1 * c2; lds2[q - 2] = (int64_t)r1 * r2;
        return kron_run<T>(ctx, f2, rows2, cols2, lds2, q - 1, batch, a_dev, y_dev, alpha, beta, bufA, bufB, nullptr);
    }
    const int nsingle = pair ? q - 2 : q;
    if (pair) ctx->last_kron_path |= 1;
    if (pair && pair_first) {
        const bool final = (nsingle == 0);
        rc = run_pair<T>(ctx, src, next_out(final), (const T*)factors[q - 2], lds[q - 2], rows[q - 2], cols[q - 2], (const T*)factors[q - 1], lds[q - 1],
                         rows[q - 1], cols[q - 1], pre2_first, final ? alpha : (T)1, final ? beta : (T)0);
        if (rc) return rc;
        cur[q - 2] = rows[q - 2]; cur[q - 1] = rows[q - 1];
        advance();
    }
    for (int k = 0; k < nsingle; ++k) {
        int64_t pre = batch, post = 1;
        for (int i = 0; i < k; ++i) pre *= rows[i];
        for (int i = k + 1; i < q; ++i) post *= cur[i];
        const bool final = (k == nsingle - 1) && !(pair && !pair_first);
        T* out = next_out(final);
        const T al = final ? alpha : (T)1, be = final ? beta : (T)0;
        const bool fits = post == 1 ? modet_ok(cols[k], lds[k]) : mode_ok(cols[k], post, lds[k]);
        if (big(k) || !fits) rc = run_blas<T>(ctx, src, out, (const T*)factors[k], lds[k], rows[k], cols[k], pre, post, al, be);
        else if (post == 1) { ctx->last_kron_path |= 4; rc = run_modet<T>(ctx, src, out, (const T*)factors[k], lds[k], rows[k], cols[k], pre, al, be); }
        else { ctx->last_kron_path |= 2; rc = run_mode<T>(ctx, src, out, (const T*)factors[k], lds[k], rows[k], cols[k], pre, post, al, be); }
        if (rc) return rc;
        cur[k] = rows[k];
        advance();
    }
    if (pair && !pair_first) {
        rc = run_pair<T>(ctx, src, y_dev, (const T*)factors[q - 2], lds[q - 2], rows[q - 2], cols[q - 2], (const T*)factors[q - 1], lds[q - 1], rows[q - 1],
                         cols[q - 1], pre2, alpha, beta);
        if (rc) return rc;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("kron_mvm: kernel launch failed: %s", hipGetErrorString(e)); return COVGRAM_EHIP; }
    return COVGRAM_OK;
}

}  // namespace kron

void ctx_blas_destroy(covgram_ctx* ctx) {
    if (ctx->blas) { (void)rocblas_destroy_handle((rocblas_handle)ctx->blas); ctx->blas = nullptr; }
}

}  // namespace covgram

using namespace covgram;

extern "C" {

int covgram_kron_mvm(covgram_ctx* ctx, const void* const* factors, const int64_t* rows, const int64_t* cols, const int64_t* lds,
                     int32_t q, int32_t dtype, const void* a, int64_t lda, void* y, int64_t ldy, int32_t nrhs, double alpha, double beta,
                     int32_t loc) {
    CG_REQUIRE(ctx && factors && rows && cols && lds && a && y, COVGRAM_EINVAL, "NULL argument");
    CG_REQUIRE(q >= 1 && q <= 16, COVGRAM_EINVAL, "kron: need 1 <= q <= 16 factors");
    CG_REQUIRE(dtype == COVGRAM_F32 || dtype == COVGRAM_F64, COVGRAM_EINVAL, "unknown dtype %d", dtype);
    CG_REQUIRE(nrhs >= 1, COVGRAM_EINVAL, "kron: nrhs must be >= 1");
    const size_t ts = dtype_size(dtype);
    int64_t nin = 1, nout = 1, maxel = 1;
    for (int i = 0; i < q; ++i) {
        CG_REQUIRE(rows[i] >= 1 && cols[i] >= 1 && lds[i] >= rows[i], COVGRAM_EINVAL, "kron: bad factor %d shape", i);
        CG_REQUIRE(rows[i] < ((int64_t)1 << 31) && cols[i] < ((int64_t)1 << 31), COVGRAM_EUNSUPPORTED, "kron: factor %d too large", i);
        nin *= cols[i]; nout *= rows[i];
    }
    CG_REQUIRE(lda >= nin && ldy >= nout, COVGRAM_EINVAL, "kron: lda / ldy smaller than the operator's sides (%lld, %lld)", (long long)nin, (long long)nout);
    {   // largest intermediate tensor, over both orders kron_run may take
        maxel = std::max(nin, nout);
        int64_t cur = nin;
        for (int k = 0; k < q; ++k) { cur = cur / cols[k] * rows[k]; maxel = std::max(maxel, cur); }
        cur = nin;
        for (int k = q - 1; k >= 0; --k) { cur = cur / cols[k] * rows[k]; maxel = std::max(maxel, cur); }
        if (q >= 2) {   // the pair first, then the leading modes
            cur = nin / cols[q - 1] * rows[q - 1] / cols[q - 2] * rows[q - 2];
            maxel = std::max(maxel, cur);
            for (int k = 0; k + 2 < q; ++k) { cur = cur / cols[k] * rows[k]; maxel = std::max(maxel, cur); }
        }
    }
    CG_DEVICE(ctx);
    // right-hand sides that lie one after the other are one more (slowest) tensor index; padded ones are packed first
    const bool packed_a = (lda == nin) || nrhs == 1, packed_y = (ldy == nout) || nrhs == 1;
    const bool host = (loc == COVGRAM_HOST);
    // workspace: [bufA | bufB | staged a | staged y | staged factors]
    size_t fbytes = 0;
    if (host) for (int i = 0; i < q; ++i) fbytes += (((size_t)rows[i] * cols[i] * ts) + 15) & ~(size_t)15;
    const size_t tens = (((size_t)maxel * nrhs * ts) + 255) & ~(size_t)255;
    const size_t abytes = (((size_t)nin * nrhs * ts) + 255) & ~(size_t)255, ybytes = (((size_t)nout * nrhs * ts) + 255) & ~(size_t)255;
    const bool stage_a = host || !packed_a, stage_y = host || !packed_y;
    const size_t mbytes = (size_t)kron::MERGE_MAX_SIDE * kron::MERGE_MAX_SIDE * ts;      // the multiplied-out pair of small trailing factors
    const size_t need = 2 * tens + (stage_a ? abytes : 0) + (stage_y ? ybytes : 0) + fbytes + mbytes + 1024;
    void* w; int rc = ws_reserve(ctx, 1, need, &w); if (rc) return rc;
    char* base = (char*)w;
    void* bufA = base; void* bufB = base + tens;
    char* p = base + 2 * tens;
    const void* a_dev = a; void* y_dev = y;
    const void* fdev[16]; int64_t ldd[16];
    for (int i = 0; i < q; ++i) { fdev[i] = factors[i]; ldd[i] = lds[i]; }
    const hipMemcpyKind up = host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    if (stage_a) {
        CG_CHECK_HIP(hipMemcpy2DAsync(p, (size_t)nin * ts, a, (size_t)lda * ts, (size_t)nin * ts, nrhs, up, ctx->stream));
        a_dev = p; p += abytes;
    }
    if (stage_y) {
        if (beta != 0.0) CG_CHECK_HIP(hipMemcpy2DAsync(p, (size_t)nout * ts, y, (size_t)ldy * ts, (size_t)nout * ts, nrhs, up, ctx->stream));
        y_dev = p; p += ybytes;
    }
    if (host) {
        for (int i = 0; i < q; ++i) {
            CG_CHECK_HIP(hipMemcpy2DAsync(p, (size_t)rows[i] * ts, factors[i], (size_t)lds[i] * ts, (size_t)rows[i] * ts, cols[i], hipMemcpyHostToDevice, ctx->stream));
            fdev[i] = p; ldd[i] = rows[i]; p += (((size_t)rows[i] * cols[i] * ts) + 15) & ~(size_t)15;
        }
    }
    void* merged = (void*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    if (dtype == COVGRAM_F32) rc = kron::kron_run<float>(ctx, fdev, rows, cols, ldd, q, nrhs, (const float*)a_dev, (float*)y_dev, (float)alpha, (float)beta, (float*)bufA, (float*)bufB, (float*)merged);
    else rc = kron::kron_run<double>(ctx, fdev, rows, cols, ldd, q, nrhs, (const double*)a_dev, (double*)y_dev, alpha, beta, (double*)bufA, (double*)bufB, (double*)merged);
    if (rc) return rc;
    if (stage_y) {
        const hipMemcpyKind down = host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
        CG_CHECK_HIP(hipMemcpy2DAsync(y, (size_t)ldy * ts, y_dev, (size_t)nout * ts, (size_t)nout * ts, nrhs, down, ctx->stream));
    }
    if (host) CG_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return COVGRAM_OK;
}

}  // extern "C"
