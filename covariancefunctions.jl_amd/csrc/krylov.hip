// covgram — the vector recurrences of one conjugate-gradient iteration in three launches (the noise term of G + sigma^2 I and the
// residual norm ride along in the first and the last one: covgram_cg_step_shifted).
//
// The reference solves G \ b with IterativeSolvers.cg! (src/gramian.jl:229-238, src/lazy_linear_algebra.jl:135-144); the package
// is a dependency (Manifest.toml: IterativeSolvers 0.9.2), not part of /root/reference.  Its published iteration, restated:
//     alpha = rho / (p . A p);   x += alpha p;   r -= alpha A p;   rho' = r . r;   p = r + (rho' / rho) p
// (no preconditioner).  Done with library vector operations that is eleven small launches per iteration — 45 us on a GPU, next to
// a 40 us MVM at n = 16384 (tools/cg_rate.py).  Here: three launches, every scalar stays on the device, sums in a fixed order.
#include <cstdint>

#include "profiles.hpp"

namespace covgram {

constexpr int CG_BLOCKS = 256, CG_THREADS = 256;

template <typename T>
__device__ __forceinline__ double block_sum(double v, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) sh[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < CG_THREADS / 64; ++i) t += sh[i];          // every thread: the same fixed-order total
    __syncthreads();
    return t;
}

// scal[0] <- scal[1] (the rho committed by the previous step);  partA[b] = sum over block b's elements of p . Ap.
// SHIFT: Ap holds G p of the Gramian alone and A = G + Diagonal(diag): the pass completes Ap <- Ap + diag .* p on its way.
template <typename T, bool SHIFT>
__global__ __launch_bounds__(CG_THREADS) void cg_dot_kernel(int64_t n, const T* __restrict__ p, T* __restrict__ Ap, T* __restrict__ scal, const T* __restrict__ diag) {
    __shared__ double sh[CG_THREADS / 64];
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[0] = scal[1];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * CG_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * CG_THREADS) {
        T ap = Ap[i];
        if (SHIFT) { ap = cg_fma(diag[i], p[i], ap); Ap[i] = ap; }
        s += (double)p[i] * (double)ap;
    }
    const double t = block_sum<T>(s, sh);
    if (threadIdx.x == 0) scal[2 + blockIdx.x] = (T)t;
}

// alpha = scal[0] / sum(partA);  x += alpha p;  r -= alpha Ap;  partB[b] = sum of r^2 over block b's elements
template <typename T>
__global__ __launch_bounds__(CG_THREADS) void cg_update_kernel(int64_t n, T* __restrict__ x, T* __restrict__ r, const T* __restrict__ p,
                                                               const T* __restrict__ Ap, T* __restrict__ scal, int nblocks) {
    __shared__ double sh[CG_THREADS / 64];
    const double pAp = block_sum<T>(threadIdx.x < nblocks ? (double)scal[2 + threadIdx.x] : 0.0, sh);
    // a converged system (r = 0 exactly: rho = 0 and p . Ap = 0) keeps iterating under a replayed graph: 0 / 0 must not reach x
    const T alpha = (scal[0] == (T)0 || pAp == 0.0) ? (T)0 : (T)((double)scal[0] / pAp);
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * CG_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * CG_THREADS) {
        x[i] = cg_fma(alpha, p[i], x[i]);
        const T ri = cg_fma(-alpha, Ap[i], r[i]);
        r[i] = ri;
        s += (double)ri * (double)ri;
    }
    const double t = block_sum<T>(s, sh);
    if (threadIdx.x == 0) scal[2 + CG_BLOCKS + blockIdx.x] = (T)t;
}

// rho' = sum(partB);  p = r + (rho' / scal[0]) p;  scal[1] <- rho'
template <typename T>
__global__ __launch_bounds__(CG_THREADS) void cg_direction_kernel(int64_t n, T* __restrict__ p, const T* __restrict__ r, T* __restrict__ scal, int nblocks, T* __restrict__ norm_out) {
    __shared__ double sh[CG_THREADS / 64];
    const double rho = block_sum<T>(threadIdx.x < nblocks ? (double)scal[2 + CG_BLOCKS + threadIdx.x] : 0.0, sh);
    const T beta = (scal[0] == (T)0) ? (T)0 : (T)(rho / (double)scal[0]);
    for (int64_t i = (int64_t)blockIdx.x * CG_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * CG_THREADS) p[i] = cg_fma(beta, p[i], r[i]);
    if (blockIdx.x == 0 && threadIdx.x == 0) { scal[1] = (T)rho; if (norm_out) *norm_out = (T)__builtin_sqrt(rho); }
}

// The whole step in ONE launch for vectors that fit one workgroup's registers (fp32: n <= 16384, fp64: n <= 8192): a thread holds
// up to 4 groups of 16 bytes of p, Ap and r, the two sums are block sums, nothing is re-read.  A graph-replayed iteration at GP sizes is bound by
// its launches (~4.6 us each for 64 KB vectors, tools/cg_rate.py): this is one instead of three, at the price of one CU's
// load / store rate (~64 B/clk: ~6 us for n = 16384 fp32).  Same recurrences, same guards; sums in a fixed order.
constexpr int CG1_THREADS = 1024;
template <typename T> struct CgVec { static constexpr int V = 16 / sizeof(T); typedef T type __attribute__((ext_vector_type(16 / sizeof(T)))); };
// V consecutive entries from i (16-byte loads / stores where the whole group is inside the vector; the ragged end entry by entry)
template <typename T> __device__ __forceinline__ void cg_load(const T* __restrict__ v, int i, int n, T (&out)[CgVec<T>::V]) {
    constexpr int V = CgVec<T>::V;
    if (i + V <= n) { const typename CgVec<T>::type q = *reinterpret_cast<const typename CgVec<T>::type*>(v + i); for (int c = 0; c < V; ++c) out[c] = q[c]; }
    else for (int c = 0; c < V; ++c) out[c] = i + c < n ? v[i + c] : (T)0;
}
template <typename T> __device__ __forceinline__ void cg_store(T* __restrict__ v, int i, int n, const T (&in)[CgVec<T>::V]) {
    constexpr int V = CgVec<T>::V;
    if (i + V <= n) { typename CgVec<T>::type q; for (int c = 0; c < V; ++c) q[c] = in[c]; *reinterpret_cast<typename CgVec<T>::type*>(v + i) = q; }
    else for (int c = 0; c < V; ++c) if (i + c < n) v[i + c] = in[c];
}
// G groups of V entries per thread: group g of thread t starts at (t + 1024 g) V (n <= 1024 G V; vectors 16-byte aligned)
template <typename T, int G, bool SHIFT>
__global__ __launch_bounds__(CG1_THREADS) void cg_step_one_kernel(int n, T* __restrict__ x, T* __restrict__ r, T* __restrict__ p, T* __restrict__ Ap,
                                                                 T* __restrict__ scal, const T* __restrict__ diag, T* __restrict__ norm_out) {
    constexpr int V = CgVec<T>::V;
    __shared__ double sh[2][CG1_THREADS / 64];
    const int t = threadIdx.x;
    T pv[G][V], av[G][V], rv[G][V];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int i = (t + g * CG1_THREADS) * V;
        cg_load(p, i, n, pv[g]); cg_load(Ap, i, n, av[g]); cg_load(r, i, n, rv[g]);
        if (SHIFT) {
            T dv[V];
            cg_load(diag, i, n, dv);
#pragma unroll
            for (int c = 0; c < V; ++c) av[g][c] = cg_fma(dv[c], pv[g][c], av[g][c]);
        }
    }
    const T rho = scal[1];
    auto total = [&](double v, int slot) {                         // fixed-order workgroup sum, the same value in every thread
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((t & 63) == 0) sh[slot][t >> 6] = v;
        __syncthreads();
        double s = 0.0;
        for (int w = 0; w < CG1_THREADS / 64; ++w) s += sh[slot][w];
        return s;
    };
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int c = 0; c < V; ++c) s += (double)pv[g][c] * (double)av[g][c];
    const double pAp = total(s, 0);
    const T alpha = (rho == (T)0 || pAp == 0.0) ? (T)0 : (T)((double)rho / pAp);
    s = 0.0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int i = (t + g * CG1_THREADS) * V;
        T xv[V];
        cg_load(x, i, n, xv);
#pragma unroll
        for (int c = 0; c < V; ++c) {
            rv[g][c] = cg_fma(-alpha, av[g][c], rv[g][c]);
            s += (double)rv[g][c] * (double)rv[g][c];
            xv[c] = cg_fma(alpha, pv[g][c], xv[c]);
        }
        cg_store(x, i, n, xv); cg_store(r, i, n, rv[g]);
        if (SHIFT) cg_store(Ap, i, n, av[g]);
    }
    const double rho_new = total(s, 1);
    const T beta = (rho == (T)0) ? (T)0 : (T)(rho_new / (double)rho);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int i = (t + g * CG1_THREADS) * V;
#pragma unroll
        for (int c = 0; c < V; ++c) pv[g][c] = cg_fma(beta, pv[g][c], rv[g][c]);
        cg_store(p, i, n, pv[g]);
    }
    if (t == 0) { scal[0] = rho; scal[1] = (T)rho_new; if (norm_out) *norm_out = (T)__builtin_sqrt(rho_new); }
}

template <typename T, int G>
static void cg_step_one(hipStream_t st, int64_t n, T* x, T* r, T* p, T* Ap, T* scal, const T* diag, bool norm) {
    T* const no = norm ? scal + 2 + 2 * CG_BLOCKS : (T*)nullptr;
    if (diag) hipLaunchKernelGGL((cg_step_one_kernel<T, G, true>), dim3(1), dim3(CG1_THREADS), 0, st, (int)n, x, r, p, Ap, scal, diag, no);
    else hipLaunchKernelGGL((cg_step_one_kernel<T, G, false>), dim3(1), dim3(CG1_THREADS), 0, st, (int)n, x, r, p, Ap, scal, diag, no);
}

template <typename T>
static void cg_step_T(hipStream_t st, int64_t n, T* x, T* r, T* p, T* Ap, T* scal, const T* diag, bool norm) {
    // up to 16 (fp32) / 8 (fp64) entries per thread of one workgroup (3 held vectors: 48 registers), 16-byte aligned vectors: one launch
    const bool aligned = (((uintptr_t)x | (uintptr_t)r | (uintptr_t)p | (uintptr_t)Ap | (uintptr_t)diag) & 15) == 0;
    constexpr int V = CgVec<T>::V;
    if (aligned && n <= (int64_t)1 * CG1_THREADS * V) return cg_step_one<T, 1>(st, n, x, r, p, Ap, scal, diag, norm);
    if (aligned && n <= (int64_t)2 * CG1_THREADS * V) return cg_step_one<T, 2>(st, n, x, r, p, Ap, scal, diag, norm);
    if (aligned && n <= (int64_t)4 * CG1_THREADS * V) return cg_step_one<T, 4>(st, n, x, r, p, Ap, scal, diag, norm);
    const int nb = (int)std::min<int64_t>(CG_BLOCKS, std::max<int64_t>(1, (n + 4 * CG_THREADS - 1) / (4 * CG_THREADS)));
    if (diag) hipLaunchKernelGGL((cg_dot_kernel<T, true>), dim3(nb), dim3(CG_THREADS), 0, st, n, (const T*)p, Ap, scal, diag);
    else hipLaunchKernelGGL((cg_dot_kernel<T, false>), dim3(nb), dim3(CG_THREADS), 0, st, n, (const T*)p, Ap, scal, diag);
    hipLaunchKernelGGL(cg_update_kernel<T>, dim3(nb), dim3(CG_THREADS), 0, st, n, x, r, (const T*)p, (const T*)Ap, scal, nb);
    hipLaunchKernelGGL(cg_direction_kernel<T>, dim3(nb), dim3(CG_THREADS), 0, st, n, p, (const T*)r, scal, nb, norm ? scal + 2 + 2 * CG_BLOCKS : (T*)nullptr);
}

}  // namespace covgram

using namespace covgram;

static int cg_step_impl(covgram_ctx* ctx, int64_t n, int32_t dtype, void* x, void* r, void* p, void* Ap, void* scal, const void* diag, bool norm) {
    CG_REQUIRE(ctx != nullptr, COVGRAM_EINVAL, "ctx is NULL");
    CG_REQUIRE(dtype == COVGRAM_F32 || dtype == COVGRAM_F64, COVGRAM_EINVAL, "dtype must be COVGRAM_F32 or COVGRAM_F64");
    CG_REQUIRE(n >= 0, COVGRAM_EINVAL, "n must be >= 0");
    if (n == 0) return COVGRAM_OK;
    CG_REQUIRE(x && r && p && Ap && scal, COVGRAM_EINVAL, "NULL vector");
    CG_DEVICE(ctx);
    if (dtype == COVGRAM_F32) cg_step_T<float>(ctx->stream, n, (float*)x, (float*)r, (float*)p, (float*)Ap, (float*)scal, (const float*)diag, norm);
    else cg_step_T<double>(ctx->stream, n, (double*)x, (double*)r, (double*)p, (double*)Ap, (double*)scal, (const double*)diag, norm);
    CG_CHECK_HIP(hipGetLastError());
    return COVGRAM_OK;
}

extern "C" int covgram_cg_step(covgram_ctx* ctx, int64_t n, int32_t dtype, void* x, void* r, void* p, const void* Ap, void* scal) {
    return cg_step_impl(ctx, n, dtype, x, r, p, const_cast<void*>(Ap), scal, nullptr, false);   // (Ap is only read without a shift)
}

extern "C" int covgram_cg_step_shifted(covgram_ctx* ctx, int64_t n, int32_t dtype, void* x, void* r, void* p, void* Ap, void* scal, const void* diag) {
    return cg_step_impl(ctx, n, dtype, x, r, p, Ap, scal, diag, true);
}
