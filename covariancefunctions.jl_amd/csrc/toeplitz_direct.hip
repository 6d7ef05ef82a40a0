// toeplitz_direct.hip — the direct solvers of src/toeplitz.jl on the device: Durbin (:14-27), Levinson
// (:77-98) and Trench (:57-71) for the symmetric positive definite Toeplitz matrix K = SymmetricToeplitz([1; r]) (unit diagonal).
//
// Durbin / Levinson are chains of n - 1 DEPENDENT steps, each a reversed dot product and a reversed axpy of length k
// (reverse_dot :114-122, reverse_increment! :125-145): there is no parallelism across steps, so ONE workgroup of 1024 threads
// walks the chain — every step is two strided passes over the first k entries (the two dot products fused into one pass, the
// two updates into the other: x and y are read as the pair (i, k-1-i) so that the in-place reversed update of y needs no
// temporary, exactly as the reference's x === y branch) and a fixed-order block reduction.  The vectors live in global
// memory (L2-resident; a workgroup's own stores are visible to it after its barrier).  O(n^2 / 1024) thread steps + 2 barriers
// per step: n = 16384 takes tens of milliseconds against the reference's 0.17 s (README.md:141-142); for large n the PCG over
// the FFT MVM (covgram.solve.toeplitz_solve, O(n log n) per iteration) is the faster path and stays the default `\`.
//
// Trench: B = inv(K) from the Durbin solution; the reference's double loop B[i,j] = B[i-1,j-1] + (v[n+1-j] v[n+1-i] -
// v[i-1] v[j-1]) / gamma is a running sum along each diagonal, so a thread owns a diagonal and walks it in the reference's
// order (bit-identical sums); consecutive lanes own consecutive diagonals -> the LOWER triangle is written coalesced, a tiled
// mirror kernel fills the upper one (the reference fills the upper triangle and wraps it in Symmetric).
#include <algorithm>

#include "common.hpp"

namespace covgram {

constexpr int LV_THREADS = 1024;

// fixed-order sum of two per-thread values over the workgroup; `buf` alternates between two halves so one barrier suffices
template <typename T>
__device__ __forceinline__ void block_sum2(T& s0, T& s1, T (*red)[2][16], int parity) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[parity][0][w] = s0; red[parity][1][w] = s1; }
    __syncthreads();
    T t0 = 0, t1 = 0;
#pragma unroll
    for (int q = 0; q < LV_THREADS / 64; ++q) { t0 += red[parity][0][q]; t1 += red[parity][1][q]; }
    s0 = t0; s1 = t1;
}

// SOLVE = false: Durbin, y = K_n \ (-r) with K_n = SymmetricToeplitz([1, r[0 .. n-2]]), n = length(r) = length(y).
// SOLVE = true:  Levinson, x = K \ b with K = SymmetricToeplitz([1; r]), n = length(b) = length(r) + 1 (y: workspace of n).
template <typename T, bool SOLVE>
__global__ __launch_bounds__(LV_THREADS) void levinson_kernel(const T* __restrict__ r, const T* __restrict__ b, T* x, T* y, int64_t n) {
    __shared__ T red[2][2][16];
    const int tid = threadIdx.x;
    if (tid == 0) { y[0] = -r[0]; if (SOLVE) x[0] = b[0]; }
    T alpha = -r[0], beta = (T)1;
    __syncthreads();
    for (int64_t k = 1; k < n; ++k) {
        beta *= ((T)1 - alpha * alpha);
        T sx = 0, sy = 0;
        for (int64_t i = tid; i < k; i += LV_THREADS) {           // reverse_dot(r_k, x_k), reverse_dot(r_k, y_k)
            const T ri = r[i];
            if (SOLVE) sx = __builtin_fma(ri, x[k - 1 - i], sx);
            sy = __builtin_fma(ri, y[k - 1 - i], sy);
        }
        block_sum2(sx, sy, red, (int)(k & 1));
        const T mu = SOLVE ? (b[k] - sx) / beta : (T)0;
        const bool more = !SOLVE || k < n - 1;                     // Levinson's last step updates x only (:90-94)
        const T an = more ? -(r[k] + sy) / beta : (T)0;
        for (int64_t i = tid; 2 * i < k; i += LV_THREADS) {        // pairs (i, j = k-1-i): both updates read the OLD y
            const int64_t j = k - 1 - i;
            const T yi = y[i], yj = y[j];
            if (SOLVE) { x[i] = __builtin_fma(mu, yj, x[i]); if (j != i) x[j] = __builtin_fma(mu, yi, x[j]); }
            if (more) { y[i] = __builtin_fma(an, yj, yi); if (j != i) y[j] = __builtin_fma(an, yi, yj); }
        }
        if (tid == 0) { if (SOLVE) x[k] = mu; if (more) y[k] = an; }
        alpha = an;
        __syncthreads();                                           // the step's stores are visible to the whole workgroup
    }
}

// gamma = 1 / (1 + r . y), nu = gamma * reverse(y): one workgroup
template <typename T>
__global__ __launch_bounds__(LV_THREADS) void trench_head_kernel(const T* __restrict__ r, const T* __restrict__ y, int64_t m, T* __restrict__ nu,
                                                                 T* __restrict__ gamma_out) {
    __shared__ T red[2][2][16];
    T s = 0, z = 0;
    for (int64_t i = threadIdx.x; i < m; i += LV_THREADS) s = __builtin_fma(r[i], y[i], s);
    block_sum2(s, z, red, 0);
    const T g = (T)1 / ((T)1 + s);
    for (int64_t i = threadIdx.x; i < m; i += LV_THREADS) nu[i] = g * y[m - 1 - i];
    if (threadIdx.x == 0) *gamma_out = g;
}

// lower triangle of B = inv(SymmetricToeplitz([1; r])): thread dl owns the diagonal j - i = dl and adds along it in the
// reference's order.  0-based: B[0][dl] = gamma (dl = 0) or gamma y[dl-1]; B[i][i+dl] = B[i-1][i-1+dl] +
// (nu[n-1-(i+dl)] nu[n-1-i] - nu[i-1] nu[i+dl-1]) / gamma, stored at (row i + dl, column i).
template <typename T>
__global__ __launch_bounds__(256) void trench_diag_kernel(const T* __restrict__ y, const T* __restrict__ nu, const T* __restrict__ gamma_p, int64_t n,
                                                          T* __restrict__ B, int64_t ldb) {
    const int64_t dl = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (dl >= n) return;
    const T g = *gamma_p;
    T v = dl == 0 ? g : g * y[dl - 1];
    B[dl] = v;                                                     // (row dl, column 0)
    for (int64_t i = 1; i + dl < n; ++i) {
        const int64_t j = i + dl;
        v += (nu[n - 1 - j] * nu[n - 1 - i] - nu[i - 1] * nu[j - 1]) / g;
        B[j + i * ldb] = v;
    }
}

// B[i][j] = B[j][i] for i < j (32 x 32 tiles through LDS: both sides coalesced)
template <typename T>
__global__ __launch_bounds__(256) void mirror_lower_kernel(T* __restrict__ B, int64_t n, int64_t ldb) {
    __shared__ T tile[32][33];
    const int64_t bi = blockIdx.x, bj = blockIdx.y;                // source tile: rows bj*32.., columns bi*32.. with bj >= bi
    if (bj < bi) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int q = ty; q < 32; q += 8) {
        const int64_t row = bj * 32 + tx, col = bi * 32 + q;
        tile[q][tx] = (row < n && col < n) ? B[row + col * ldb] : (T)0;
    }
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {
        const int64_t row = bi * 32 + tx, col = bj * 32 + q;       // destination (row, col) = source (col, row)
        if (row < n && col < n && row < col) B[row + col * ldb] = tile[tx][q];
    }
}

template <typename T>
static int durbin_run(covgram_ctx* ctx, const T* r, int64_t n, T* y) {
    hipLaunchKernelGGL((levinson_kernel<T, false>), dim3(1), dim3(LV_THREADS), 0, ctx->stream, r, (const T*)nullptr, (T*)nullptr, y, n);
    return COVGRAM_OK;
}

}  // namespace covgram

using namespace covgram;

extern "C" {

// staging helper: device copies of host inputs in workspace slot 1 (layout decided by the caller)
static int stage_in(covgram_ctx* ctx, char*& p, const void* src, size_t bytes, int32_t loc, const void** dev) {
    if (loc == COVGRAM_DEVICE) { *dev = src; return COVGRAM_OK; }
    CG_CHECK_HIP(hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    *dev = p; p += (bytes + 255) & ~(size_t)255;
    return COVGRAM_OK;
}

int covgram_toeplitz_durbin(covgram_ctx* ctx, const void* r, int64_t n, void* y, int32_t dtype, int32_t loc) {
    CG_REQUIRE(ctx && r && y, COVGRAM_EINVAL, "NULL argument");
    CG_REQUIRE(n >= 1, COVGRAM_EINVAL, "durbin: need n >= 1");
    CG_REQUIRE(n <= COVGRAM_TOEPLITZ_DIRECT_MAX_N, COVGRAM_EUNSUPPORTED, "durbin: n = %lld is above the direct solvers' cap of %d (one uninterruptible launch of n - 1 dependent O(n) steps): use the preconditioned CG over the FFT MVM", (long long)n, COVGRAM_TOEPLITZ_DIRECT_MAX_N);
    CG_REQUIRE(dtype == COVGRAM_F32 || dtype == COVGRAM_F64, COVGRAM_EINVAL, "unknown dtype %d", dtype);
    const size_t ts = dtype_size(dtype), vb = ((size_t)n * ts + 255) & ~(size_t)255;
    CG_DEVICE(ctx);
    void* w; int rc = ws_reserve(ctx, 1, 2 * vb + 512, &w); if (rc) return rc;
    char* p = (char*)w;
    const void* rd; rc = stage_in(ctx, p, r, (size_t)n * ts, loc, &rd); if (rc) return rc;
    void* yd = loc == COVGRAM_DEVICE ? y : (void*)p;
    if (dtype == COVGRAM_F32) durbin_run<float>(ctx, (const float*)rd, n, (float*)yd);
    else durbin_run<double>(ctx, (const double*)rd, n, (double*)yd);
    CG_CHECK_HIP(hipGetLastError());
    if (loc == COVGRAM_HOST) { CG_CHECK_HIP(hipMemcpyAsync(y, yd, (size_t)n * ts, hipMemcpyDeviceToHost, ctx->stream)); CG_CHECK_HIP(hipStreamSynchronize(ctx->stream)); }
    return COVGRAM_OK;
}

int covgram_toeplitz_levinson(covgram_ctx* ctx, const void* r, const void* b, int64_t n, void* x, int32_t dtype, int32_t loc) {
    CG_REQUIRE(ctx && b && x && (r || n == 1), COVGRAM_EINVAL, "NULL argument");
    CG_REQUIRE(n >= 1, COVGRAM_EINVAL, "levinson: need n >= 1");
    CG_REQUIRE(n <= COVGRAM_TOEPLITZ_DIRECT_MAX_N, COVGRAM_EUNSUPPORTED, "levinson: n = %lld is above the direct solvers' cap of %d (one uninterruptible launch of n - 1 dependent O(n) steps): use the preconditioned CG over the FFT MVM", (long long)n, COVGRAM_TOEPLITZ_DIRECT_MAX_N);
    CG_REQUIRE(dtype == COVGRAM_F32 || dtype == COVGRAM_F64, COVGRAM_EINVAL, "unknown dtype %d", dtype);
    const size_t ts = dtype_size(dtype), vb = ((size_t)n * ts + 255) & ~(size_t)255;
    CG_DEVICE(ctx);
    void* w; int rc = ws_reserve(ctx, 1, 4 * vb + 512, &w); if (rc) return rc;
    char* p = (char*)w;
    void* yw = p; p += vb;                                         // the recursion's y (workspace)
    if (n == 1) {                                                  // K = [1]
        if (loc == COVGRAM_HOST) memcpy(x, b, ts); else CG_CHECK_HIP(hipMemcpyAsync(x, b, ts, hipMemcpyDeviceToDevice, ctx->stream));
        return COVGRAM_OK;
    }
    const void *rd, *bd;
    rc = stage_in(ctx, p, r, (size_t)(n - 1) * ts, loc, &rd); if (rc) return rc;
    rc = stage_in(ctx, p, b, (size_t)n * ts, loc, &bd); if (rc) return rc;
    void* xd = loc == COVGRAM_DEVICE ? x : (void*)p;
    if (dtype == COVGRAM_F32)
        hipLaunchKernelGGL((levinson_kernel<float, true>), dim3(1), dim3(LV_THREADS), 0, ctx->stream, (const float*)rd, (const float*)bd, (float*)xd, (float*)yw, n);
    else
        hipLaunchKernelGGL((levinson_kernel<double, true>), dim3(1), dim3(LV_THREADS), 0, ctx->stream, (const double*)rd, (const double*)bd, (double*)xd, (double*)yw, n);
    CG_CHECK_HIP(hipGetLastError());
    if (loc == COVGRAM_HOST) { CG_CHECK_HIP(hipMemcpyAsync(x, xd, (size_t)n * ts, hipMemcpyDeviceToHost, ctx->stream)); CG_CHECK_HIP(hipStreamSynchronize(ctx->stream)); }
    return COVGRAM_OK;
}

int covgram_toeplitz_trench(covgram_ctx* ctx, const void* r, int64_t n, void* B, int64_t ldb, int32_t dtype, int32_t loc) {
    CG_REQUIRE(ctx && B && (r || n == 1), COVGRAM_EINVAL, "NULL argument");
    CG_REQUIRE(n >= 1 && ldb >= n, COVGRAM_EINVAL, "trench: need n >= 1 and ldb >= n");
    CG_REQUIRE(n <= COVGRAM_TOEPLITZ_DIRECT_MAX_N, COVGRAM_EUNSUPPORTED, "trench: n = %lld is above the direct solvers' cap of %d", (long long)n, COVGRAM_TOEPLITZ_DIRECT_MAX_N);
    CG_REQUIRE(dtype == COVGRAM_F32 || dtype == COVGRAM_F64, COVGRAM_EINVAL, "unknown dtype %d", dtype);
    const size_t ts = dtype_size(dtype), vb = ((size_t)n * ts + 255) & ~(size_t)255;
    CG_DEVICE(ctx);
    const size_t mat = loc == COVGRAM_HOST ? (size_t)n * n * ts : 0;
    void* w; int rc = ws_reserve(ctx, 1, 4 * vb + mat + 1024, &w); if (rc) return rc;
    char* p = (char*)w;
    void* yw = p; p += vb;
    void* nu = p; p += vb;
    void* gam = p; p += 256;
    void* Bd = B; int64_t ld = ldb;
    const void* rd = nullptr;
    if (n > 1) { rc = stage_in(ctx, p, r, (size_t)(n - 1) * ts, loc, &rd); if (rc) return rc; }
    if (loc == COVGRAM_HOST) { Bd = p; ld = n; }
    const unsigned gd = (unsigned)((n + 255) / 256), gt = (unsigned)((n + 31) / 32);
#define CG_TRENCH(T)                                                                                                                   \
    {                                                                                                                                  \
        if (n > 1) {                                                                                                                   \
            durbin_run<T>(ctx, (const T*)rd, n - 1, (T*)yw);                                                                           \
            hipLaunchKernelGGL(trench_head_kernel<T>, dim3(1), dim3(LV_THREADS), 0, ctx->stream, (const T*)rd, (const T*)yw, n - 1, (T*)nu, (T*)gam); \
        } else {                                                                                                                       \
            const T one = (T)1;                                                                                                        \
            CG_CHECK_HIP(hipMemcpyAsync(gam, &one, sizeof(T), hipMemcpyHostToDevice, ctx->stream));                                    \
            CG_CHECK_HIP(hipStreamSynchronize(ctx->stream));                                                                           \
        }                                                                                                                              \
        hipLaunchKernelGGL(trench_diag_kernel<T>, dim3(gd), dim3(256), 0, ctx->stream, (const T*)yw, (const T*)nu, (const T*)gam, n, (T*)Bd, ld); \
        hipLaunchKernelGGL(mirror_lower_kernel<T>, dim3(gt, gt), dim3(256), 0, ctx->stream, (T*)Bd, n, ld);                            \
    }
    if (dtype == COVGRAM_F32) CG_TRENCH(float) else CG_TRENCH(double)
#undef CG_TRENCH
    CG_CHECK_HIP(hipGetLastError());
    if (loc == COVGRAM_HOST) {
        CG_CHECK_HIP(hipMemcpy2DAsync(B, (size_t)ldb * ts, Bd, (size_t)n * ts, (size_t)n * ts, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        CG_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return COVGRAM_OK;
}

}  // extern "C"
