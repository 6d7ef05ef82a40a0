// toeplitz.hip — O(N log N) MVM with SymmetricToeplitz / Toeplitz / Circulant Gramians.
//
// Reference: gramian(k, x::StepRangeLen, y::StepRangeLen) builds SymmetricToeplitz(k.(x[1], x)),
// Toeplitz(k.(x, y[1]), k.(x[1], y)) or Circulant(k.(x[1], x)) (src/gramian.jl:167-189); the MVM itself
// is ToeplitzMatrices 0.7.1 (+ FFTW 1.5.0), a third-party dependency whose source is not part of the
// reference tree.  Its published algorithm is restated here: embed T (n×m) in an N×N circulant
// C with first column  c = [vc_0 .. vc_{n-1}, 0 .., vr_{m-1} .. vr_1], then  T a = (C [a; 0])_{0..n-1}
// = irfft(rfft(c) ⊙ rfft([a; 0])).  Any N >= n+m-1 gives the identical product, so N is the next
// power of two (real-to-complex rocFFT, half the traffic of the complex transform the reference uses),
// and — unlike the reference, which re-plans and re-transforms c on every mul! — the plan and the
// spectrum of c (pre-divided by N) are cached in the handle.
//
// HBM-bound: per MVM one pad pass, R2C, one pointwise pass, C2R, one epilogue pass.
#include <rocfft/rocfft.h>

#include "common.hpp"

namespace covgram {

static int g_rocfft_users = 0;

#define CG_CHECK_FFT(expr)                                                                     \
    do {                                                                                       \
        rocfft_status _s = (expr);                                                             \
        if (_s != rocfft_status_success) {                                                     \
            ::covgram::set_error("%s failed: rocfft_status %d (%s:%d)", #expr, (int)_s, __FILE__, __LINE__); \
            return COVGRAM_EHIP;                                                               \
        }                                                                                      \
    } while (0)

// c[0..n) = vc, c[n..N-m+1) = 0, c[N-(m-1)+k] = vr[m-1-k]  (k = 0..m-2)
template <typename T>
__global__ __launch_bounds__(256) void embed_kernel(const T* __restrict__ vc, const T* __restrict__ vr, int64_t n, int64_t m,
                                                    int64_t N, T* __restrict__ c) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    T v = (T)0;
    if (i < n) v = vc[i];
    else if (i >= N - (m - 1)) v = vr[N - i];
    c[i] = v;
}

// buf[0..m) = a, buf[m..N) = 0   (16-byte vectorised where aligned)
template <typename T>
__global__ __launch_bounds__(256) void pad_kernel(const T* __restrict__ a, int64_t m, int64_t N, T* __restrict__ buf) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    buf[i] = (i < m) ? a[i] : (T)0;
}

// spec <- spec * s   (complex, interleaved); also used to pre-scale the cached spectrum by 1/N
template <typename T>
__global__ __launch_bounds__(256) void cmul_kernel(T* __restrict__ z, const T* __restrict__ s, int64_t nc, T scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    const T zr = z[2 * i], zi = z[2 * i + 1];
    if (s) {
        const T sr = s[2 * i], si = s[2 * i + 1];
        z[2 * i] = (zr * sr - zi * si) * scale;
        z[2 * i + 1] = (zr * si + zi * sr) * scale;
    } else {
        z[2 * i] = zr * scale;
        z[2 * i + 1] = zi * scale;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void toep_epilogue_kernel(const T* __restrict__ t, T* __restrict__ y, int64_t n, T alpha, T beta) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T v = alpha * t[i];
    if (beta != (T)0) v = __builtin_fma(beta, y[i], v);
    y[i] = v;
}

}  // namespace covgram

using namespace covgram;

struct covgram_toeplitz {
    covgram_ctx* ctx = nullptr;
    int64_t n = 0, m = 0, N = 0;
    int32_t dtype = 0;
    bool circulant = false;
    rocfft_plan fwd = nullptr, inv = nullptr;
    rocfft_execution_info info = nullptr;
    void* work = nullptr; size_t work_bytes = 0;
    void* spec = nullptr;   // (N/2+1) complex, cached, pre-divided by N
    void* rbuf = nullptr;   // N reals
    void* cbuf = nullptr;   // (N/2+1) complex
    void* stage_a = nullptr; void* stage_y = nullptr;  // device staging for loc == HOST
};

static int64_t next_pow2(int64_t v) { int64_t p = 1; while (p < v) p <<= 1; return p; }

extern "C" {

int covgram_toeplitz_destroy(covgram_toeplitz* T) {
    if (!T) return COVGRAM_OK;
    (void)hipSetDevice(T->ctx->device);
    (void)hipStreamSynchronize(T->ctx->stream);
    if (T->fwd) rocfft_plan_destroy(T->fwd);
    if (T->inv) rocfft_plan_destroy(T->inv);
    if (T->info) rocfft_execution_info_destroy(T->info);
    void* bufs[] = {T->work, T->spec, T->rbuf, T->cbuf, T->stage_a, T->stage_y};
    for (void* b : bufs) if (b) (void)hipFree(b);
    T->ctx->live_handles--;
    if (--g_rocfft_users == 0) rocfft_cleanup();
    delete T;
    return COVGRAM_OK;
}

int covgram_toeplitz_create(covgram_ctx* ctx, covgram_toeplitz** out, const void* vc, const void* vr, int64_t n, int64_t m,
                            int32_t dtype, int32_t loc, int32_t circulant) {
    CG_REQUIRE(ctx && out && vc, COVGRAM_EINVAL, "NULL argument");
    CG_REQUIRE(n >= 1, COVGRAM_EINVAL, "toeplitz: n must be >= 1");
    CG_REQUIRE(dtype == COVGRAM_F32 || dtype == COVGRAM_F64, COVGRAM_EINVAL, "unknown dtype %d", dtype);
    if (!vr) m = n;
    CG_REQUIRE(m >= 1, COVGRAM_EINVAL, "toeplitz: m must be >= 1");
    CG_REQUIRE(!(circulant && vr), COVGRAM_EINVAL, "circulant takes a first column only");
    CG_CHECK_HIP(hipSetDevice(ctx->device));
    const size_t ts = dtype_size(dtype);
    if (g_rocfft_users++ == 0) rocfft_setup();
    covgram_toeplitz* T = new covgram_toeplitz();
    T->ctx = ctx; T->n = n; T->m = m; T->dtype = dtype; T->circulant = circulant != 0;
    T->N = circulant ? n : next_pow2(std::max<int64_t>(n + m - 1, 2));
    ctx->live_handles++;
    const int64_t N = T->N, NC = N / 2 + 1;
    int rc = COVGRAM_OK;
    auto fail = [&](int code) { covgram_toeplitz_destroy(T); return code; };
#define TRY_HIP(e) do { hipError_t _e = (e); if (_e != hipSuccess) { set_error("%s failed: %s", #e, hipGetErrorString(_e)); return fail(COVGRAM_EHIP); } } while (0)
#define TRY_FFT(e) do { rocfft_status _s = (e); if (_s != rocfft_status_success) { set_error("%s failed: rocfft_status %d", #e, (int)_s); return fail(COVGRAM_EHIP); } } while (0)
    TRY_HIP(hipMalloc(&T->spec, (size_t)NC * 2 * ts));
    TRY_HIP(hipMalloc(&T->rbuf, (size_t)N * ts));
    TRY_HIP(hipMalloc(&T->cbuf, (size_t)NC * 2 * ts));
    const rocfft_precision prec = (dtype == COVGRAM_F64) ? rocfft_precision_double : rocfft_precision_single;
    size_t len[1] = {(size_t)N};
    TRY_FFT(rocfft_plan_create(&T->fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward, prec, 1, len, 1, nullptr));
    TRY_FFT(rocfft_plan_create(&T->inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, prec, 1, len, 1, nullptr));
    size_t w1 = 0, w2 = 0;
    TRY_FFT(rocfft_plan_get_work_buffer_size(T->fwd, &w1));
    TRY_FFT(rocfft_plan_get_work_buffer_size(T->inv, &w2));
    T->work_bytes = std::max(w1, w2);
    TRY_FFT(rocfft_execution_info_create(&T->info));
    if (T->work_bytes) {
        TRY_HIP(hipMalloc(&T->work, T->work_bytes));
        TRY_FFT(rocfft_execution_info_set_work_buffer(T->info, T->work, T->work_bytes));
    }
    TRY_FFT(rocfft_execution_info_set_stream(T->info, ctx->stream));

    // first column / row to the device
    const void* dvc = vc; const void* dvr = vr;
    void *tmp_c = nullptr, *tmp_r = nullptr;
    if (loc == COVGRAM_HOST) {
        TRY_HIP(hipMalloc(&tmp_c, (size_t)n * ts));
        TRY_HIP(hipMemcpyAsync(tmp_c, vc, (size_t)n * ts, hipMemcpyHostToDevice, ctx->stream));
        dvc = tmp_c;
        if (vr) {
            TRY_HIP(hipMalloc(&tmp_r, (size_t)m * ts));
            TRY_HIP(hipMemcpyAsync(tmp_r, vr, (size_t)m * ts, hipMemcpyHostToDevice, ctx->stream));
            dvr = tmp_r;
        }
    }
    if (!dvr) dvr = dvc;   // symmetric: vr = vc
    const unsigned gN = (unsigned)((N + 255) / 256), gC = (unsigned)((NC + 255) / 256);
    const int64_t me = T->circulant ? 1 : m;   // circulant: plain copy of vc (no wrapped row part)
    if (dtype == COVGRAM_F32) hipLaunchKernelGGL(embed_kernel<float>, dim3(gN), dim3(256), 0, ctx->stream, (const float*)dvc, (const float*)dvr, n, me, N, (float*)T->rbuf);
    else hipLaunchKernelGGL(embed_kernel<double>, dim3(gN), dim3(256), 0, ctx->stream, (const double*)dvc, (const double*)dvr, n, me, N, (double*)T->rbuf);
    void* in[1] = {T->rbuf}; void* outb[1] = {T->spec};
    TRY_FFT(rocfft_execute(T->fwd, in, outb, T->info));
    if (dtype == COVGRAM_F32) hipLaunchKernelGGL(cmul_kernel<float>, dim3(gC), dim3(256), 0, ctx->stream, (float*)T->spec, (const float*)nullptr, NC, 1.0f / (float)N);
    else hipLaunchKernelGGL(cmul_kernel<double>, dim3(gC), dim3(256), 0, ctx->stream, (double*)T->spec, (const double*)nullptr, NC, 1.0 / (double)N);
    TRY_HIP(hipStreamSynchronize(ctx->stream));
    if (tmp_c) (void)hipFree(tmp_c);
    if (tmp_r) (void)hipFree(tmp_r);
#undef TRY_HIP
#undef TRY_FFT
    (void)rc;
    *out = T;
    return COVGRAM_OK;
}

int covgram_toeplitz_mvm(covgram_toeplitz* T, const void* a, void* y, double alpha, double beta, int32_t loc) {
    CG_REQUIRE(T && a && y, COVGRAM_EINVAL, "NULL argument");
    covgram_ctx* ctx = T->ctx;
    CG_CHECK_HIP(hipSetDevice(ctx->device));
    const size_t ts = dtype_size(T->dtype);
    const int64_t n = T->n, m = T->m, N = T->N, NC = N / 2 + 1;
    const void* a_dev = a; void* y_dev = y;
    if (loc == COVGRAM_HOST) {
        if (!T->stage_a) CG_CHECK_HIP(hipMalloc(&T->stage_a, (size_t)m * ts));
        if (!T->stage_y) CG_CHECK_HIP(hipMalloc(&T->stage_y, (size_t)n * ts));
        CG_CHECK_HIP(hipMemcpyAsync(T->stage_a, a, (size_t)m * ts, hipMemcpyHostToDevice, ctx->stream));
        if (beta != 0.0) CG_CHECK_HIP(hipMemcpyAsync(T->stage_y, y, (size_t)n * ts, hipMemcpyHostToDevice, ctx->stream));
        a_dev = T->stage_a; y_dev = T->stage_y;
    }
    CG_CHECK_FFT(rocfft_execution_info_set_stream(T->info, ctx->stream));
    const unsigned gN = (unsigned)((N + 255) / 256), gC = (unsigned)((NC + 255) / 256), gn = (unsigned)((n + 255) / 256);
    if (T->dtype == COVGRAM_F32) hipLaunchKernelGGL(pad_kernel<float>, dim3(gN), dim3(256), 0, ctx->stream, (const float*)a_dev, m, N, (float*)T->rbuf);
    else hipLaunchKernelGGL(pad_kernel<double>, dim3(gN), dim3(256), 0, ctx->stream, (const double*)a_dev, m, N, (double*)T->rbuf);
    { void* in[1] = {T->rbuf}; void* ob[1] = {T->cbuf}; CG_CHECK_FFT(rocfft_execute(T->fwd, in, ob, T->info)); }
    if (T->dtype == COVGRAM_F32) hipLaunchKernelGGL(cmul_kernel<float>, dim3(gC), dim3(256), 0, ctx->stream, (float*)T->cbuf, (const float*)T->spec, NC, 1.0f);
    else hipLaunchKernelGGL(cmul_kernel<double>, dim3(gC), dim3(256), 0, ctx->stream, (double*)T->cbuf, (const double*)T->spec, NC, 1.0);
    { void* in[1] = {T->cbuf}; void* ob[1] = {T->rbuf}; CG_CHECK_FFT(rocfft_execute(T->inv, in, ob, T->info)); }
    if (T->dtype == COVGRAM_F32) hipLaunchKernelGGL(toep_epilogue_kernel<float>, dim3(gn), dim3(256), 0, ctx->stream, (const float*)T->rbuf, (float*)y_dev, n, (float)alpha, (float)beta);
    else hipLaunchKernelGGL(toep_epilogue_kernel<double>, dim3(gn), dim3(256), 0, ctx->stream, (const double*)T->rbuf, (double*)y_dev, n, alpha, beta);
    CG_CHECK_HIP(hipGetLastError());
    if (loc == COVGRAM_HOST) {
        CG_CHECK_HIP(hipMemcpyAsync(y, y_dev, (size_t)n * ts, hipMemcpyDeviceToHost, ctx->stream));
        CG_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return COVGRAM_OK;
}

}  // extern "C"
