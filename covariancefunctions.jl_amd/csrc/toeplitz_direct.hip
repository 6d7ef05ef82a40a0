// toeplitz_direct.hip — the direct solvers of src/toeplitz.jl on the device: Durbin (:14-27), Levinson
// (:77-98) and Trench (:57-71) for the symmetric positive definite Toeplitz matrix K = SymmetricToeplitz([1; r]) (unit diagonal).
//
// Durbin / Levinson are chains of n - 1 DEPENDENT steps, each a reversed dot product and a reversed axpy of length k
// (reverse_dot :114-122, reverse_increment! :125-145): there is no parallelism across steps, so ONE workgroup walks the chain.
// Two kernels.  n <= 16384 (levinson_reg_kernel, round 3): x, y and the r operand in registers, an LDS copy of y for the reversed
// reads, nothing in global memory inside the chain — n = 16384 fp64 in 28 ms against the reference's 0.17 s (README.md:141-142).
// Above it, up to COVGRAM_TOEPLITZ_DIRECT_MAX_N (levinson_kernel, round 2): 1024 threads, every step two strided passes over the
// first k entries (the two dot products fused into one pass, the two updates into the other: x and y are read as the pair
// (i, k-1-i) so that the in-place reversed update of y needs no temporary, exactly as the reference's x === y branch) and a
// fixed-order block reduction; the vectors live in global memory (L2-resident; a workgroup's own stores are visible to it after
// its barrier), ~8 us a step.  For large n the PCG over the FFT MVM (covgram.solve.toeplitz_solve, O(n log n) per iteration) is
// the faster path and stays the default `\`.
//
// Trench: B = inv(K) from the Durbin solution; the reference's double loop B[i,j] = B[i-1,j-1] + (v[n+1-j] v[n+1-i] -
// v[i-1] v[j-1]) / gamma is a running sum along each diagonal, so a thread owns a diagonal and walks it in the reference's
// order (bit-identical sums); consecutive lanes own consecutive diagonals -> the LOWER triangle is written coalesced, a tiled
// mirror kernel fills the upper one (the reference fills the upper triangle and wraps it in Symmetric).
#include <algorithm>
#include <type_traits>

#include "common.hpp"

namespace covgram {

constexpr int LV_THREADS = 1024;

// Sum over the 64 lanes of a wave, the same value in every lane.  Cross-lane moves inside a row of 16 are DPP modifiers on a
// register move (quad permutes, half-row and row mirrors: no LDS round trip, which is what a __shfl_xor costs — ds_bpermute, ~100+
// cycles a stage, 6 dependent stages); the four row totals are read into scalar registers.  Fixed order.
template <int CTRL> __device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ double dpp_move(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float read_lane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double read_lane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <typename T> __device__ __forceinline__ T wave_sum(T v) {
    v += dpp_move<0xB1>(v);                                        // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_move<0x4E>(v);                                        // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp_move<0x141>(v);                                       // row_half_mirror: the other quad of the 8 (all 4 lanes of a quad agree)
    v += dpp_move<0x140>(v);                                       // row_mirror: the other half of the row of 16
    return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}

// fixed-order sum of two per-thread values over the workgroup; `buf` alternates between two halves so one barrier suffices
template <typename T>
__device__ __forceinline__ void block_sum2(T& s0, T& s1, T (*red)[2][16], int parity, int nwaves = LV_THREADS / 64) {
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[parity][0][w] = s0; red[parity][1][w] = s1; }
    __syncthreads();
    T t0[2] = {0, 0}, t1[2] = {0, 0};
    for (int q = 0; q < nwaves; ++q) { t0[q & 1] += red[parity][0][q]; t1[q & 1] += red[parity][1][q]; }
    s0 = t0[0] + t0[1]; s1 = t1[0] + t1[1];
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
            if (SOLVE) sx = fma_t(ri, x[k - 1 - i], sx);
            sy = fma_t(ri, y[k - 1 - i], sy);
        }
        block_sum2(sx, sy, red, (int)(k & 1));
        const T mu = SOLVE ? (b[k] - sx) / beta : (T)0;
        const bool more = !SOLVE || k < n - 1;                     // Levinson's last step updates x only (:90-94)
        const T an = more ? -(r[k] + sy) / beta : (T)0;
        for (int64_t i = tid; 2 * i < k; i += LV_THREADS) {        // pairs (i, j = k-1-i): both updates read the OLD y
            const int64_t j = k - 1 - i;
            const T yi = y[i], yj = y[j];
            if (SOLVE) { x[i] = fma_t(mu, yj, x[i]); if (j != i) x[j] = fma_t(mu, yi, x[j]); }
            if (more) { y[i] = fma_t(an, yj, yi); if (j != i) y[j] = fma_t(an, yi, yj); }
        }
        if (tid == 0) { if (SOLVE) x[k] = mu; if (more) y[k] = an; }
        alpha = an;
        __syncthreads();                                           // the step's stores are visible to the whole workgroup
    }
}

// The same recursion with the state on chip, for n <= LR_MAX_N (the sizes the reference quotes: README.md:141-142 is n = 16384).
// The chain is latency-bound — n - 1 dependent steps of a few hundred instructions each — so what a step costs is its memory
// round trips, not its flops.  Layout:
//   * thread t OWNS the E entries e = E t .. E t + E - 1 of x and y, in registers; an LDS copy of y (16384 fp64 entries + 1 pad per
//     E = 132 KB of the CU's 160 KB; the pad makes the lane stride E + 1 entries: conflict-free) is what every thread reads its
//     PARTNER entries y[k-1-e] from.  E by size and type, see the geometry note at the kernel (n = 16384 fp64 Levinson: 512 threads
//     x 32 — x, y and the sliding r are 3 E values per thread: 192 of the 256 registers a thread has at 2 waves per SIMD);
//   * the reversed dot products are taken over the OWN entries, sum_e x[e] r[k-1-e]: the r operand slides by one entry per step,
//     so each thread keeps rs[e] = r[k-1-e] in registers and shifts it (register renames within the thread, one lane shuffle
//     for its first entry, one LDS word per wave for the wave boundary, r[k] enters at e = 0).  The sums touch no memory at all,
//     and the loop no global memory but the 64-entry blocks of r and b, loaded 64 steps ahead and read across lanes;
//   * a step: sums (registers) | barrier (the reduction's) | partner reads from the LDS copy + update of x and y in registers |
//     barrier | own y -> LDS copy.  The copy's writes of step k and its reads of step k + 1 are separated by that step's
//     reduction barrier: two barriers a step.
// Entries of the update are the reference's expressions (x[e] += mu y[k-1-e], y[e] += alpha y[k-1-e]); only the order of the
// two dot-product sums differs from the serial loop (tests: tolerance of the direct solvers).
constexpr int LR_MAX_N = 16384;
// -DLV_DIAG (tools/levinson_step_probe.hip only): thread 0 stamps the shader clock at the phase boundaries of one step
#ifdef LV_DIAG
__device__ unsigned long long lv_stamps[16];
__device__ int lv_diag_step;
#define LV_STAMP(i) do { if (k == lv_diag_step && tid == 0) lv_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define LV_STAMP(i) do { } while (0)
#endif
// Geometry: E = 2^LOG_E entries per thread, up to MAXT threads (capacity MAXT * E entries; the launch uses ceil(n / E) threads, whole
// waves).  A thread's entries are a serial stream, so E is as small as the size allows: 8 for n <= 8192 (1024 threads: n = 4096 fp64
// 5.5 -> 4.2 ms, n = 1024 1.47 -> 0.93 against E = 32); above that 16 (fp32, 1024 threads) or 32 on 512 threads (fp64: the 3 E fp64
// values per thread of Levinson need the 256 registers of two waves per SIMD; Durbin measured the same at 16 x 1024: all SIMDs are busy
// at these sizes either way).
template <typename T, bool SOLVE, int LOG_E, int MAXT>
__global__ __launch_bounds__(MAXT) void levinson_reg_kernel(const T* __restrict__ r, const T* __restrict__ b, T* __restrict__ xout,
                                                            T* __restrict__ yout, int n) {
    constexpr int LR_E = 1 << LOG_E, LR_CH = 4, LR_CAP = MAXT * LR_E;
    static_assert(LR_CAP <= LR_MAX_N && MAXT <= 1024, "geometry");
    const int nwaves = (int)(blockDim.x >> 6);
    // LDS copy of y: entry e at slot e + (e >> LOG_E), behind a prefix that the same slot formula maps the entries -1 .. -E to:
    // slot -2 (entry -1) holds 1, the others 0 — see the update
    constexpr int LR_PRE = LR_E + 2;
    __shared__ T Ybuf[LR_PRE + LR_CAP + LR_CAP / LR_E];
    T* const Y = Ybuf + LR_PRE;
    __shared__ T red[2][2][16];
    __shared__ T edge[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, e0 = tid * LR_E;
    T* const yown = Y + e0 + tid;                                  // lr_pad(e0 + q) = e0 + tid + q
    T x[LR_E], y[LR_E], rs[LR_E];
#pragma unroll
    for (int q = 0; q < LR_E; ++q) { x[q] = 0; y[q] = 0; rs[q] = 0; yown[q] = 0; }
    if (tid < LR_PRE) Ybuf[tid] = tid == LR_PRE - 2 ? (T)1 : (T)0;
    const T r0 = r[0];
    if (tid == 0) { y[0] = -r0; yown[0] = -r0; if (SOLVE) x[0] = b[0]; rs[0] = r0; }
    T alpha = -r0, beta = (T)1;
    const int rlen = SOLVE ? n - 1 : n;                            // entries of r
    // r[k], b[k]: lane l of every wave holds entry 64 (k / 64) + l of the current block of 64 and of the next one (a vector load issued
    // 64 steps ahead); the step reads its entry across lanes.  (A scalar load per step counts on the same counter as the LDS
    // operations: its full latency would be waited for at the step's first LDS wait.)
    auto block_of = [&](const T* v, int len, int kb) { const int i = kb + lane; return i < len ? v[i] : (T)0; };
    T rcur = block_of(r, rlen, 0), rnxt = block_of(r, rlen, 64), bcur = 0, bnxt = 0;
    if (SOLVE) { bcur = block_of(b, n, 0); bnxt = block_of(b, n, 64); }
    __syncthreads();
    for (int k = 1; k < n; ++k) {
        if ((k & 63) == 0) {
            rcur = rnxt; rnxt = block_of(r, rlen, k + 64);
            if (SOLVE) { bcur = bnxt; bnxt = block_of(b, n, k + 64); }
        }
        LV_STAMP(0);
        const T rk = read_lane(rcur, k & 63), bk = SOLVE ? read_lane(bcur, k & 63) : (T)0;
        beta *= ((T)1 - alpha * alpha);
        const bool live = e0 <= k;                                 // this thread owns an entry the step reads or creates
        int c = k - e0;                                            // own entry q is below k iff q < c.  Opaque: otherwise the per-entry
        asm volatile("" : "+v"(c));                                // constants (e0 + q, -1 - e0 - q) are hoisted into 2 E registers
        T sx = 0, sy = 0;
        if (live) {                                                // entries e >= k of x, y and rs are zero: no masks in the sums
            T px[4] = {0, 0, 0, 0}, py[4] = {0, 0, 0, 0};          // 4 chains a sum: the fma latency, not its issue rate, is what a step waits on
#pragma unroll
            for (int q = 0; q < LR_E; ++q) {
                if (SOLVE) px[q & 3] = fma_t(x[q], rs[q], px[q & 3]);
                py[q & 3] = fma_t(y[q], rs[q], py[q & 3]);
            }
            sx = (px[0] + px[1]) + (px[2] + px[3]); sy = (py[0] + py[1]) + (py[2] + py[3]);
            if (lane == 63) edge[wave] = rs[LR_E - 1];
        }
        const T ibeta = (T)1 / beta;                               // off the chain: beta is known before the sums are
        LV_STAMP(1);
        block_sum2<T>(sx, sy, red, k & 1, nwaves);
        LV_STAMP(2);
        // (b[k] - sx) / beta and -(r[k] + sy) / beta of :20, :88 as products with 1 / beta: a division here is ~30 dependent
        // instructions on the step's critical path
        const T mu = SOLVE ? (bk - sx) * ibeta : (T)0;
        const bool more = !SOLVE || k < n - 1;                     // Levinson's last step updates x only (:90-94)
        const T an = more ? -(rk + sy) * ibeta : (T)0;
        // the update, a chunk of partner entries at a time: x[e] += mu y[m - q], y[e] += alpha y[m - q] with m = k - 1 - e0 the partner
        // of the thread's first entry.  At the thread that owns the front of the vectors m - q runs below 0: those "entries" are the
        // prefix of the LDS copy — 1 for m - q = -1, i.e. e = k: x[k] = mu and y[k] = alpha come out of the same fma on zeros — and 0
        // beyond it: no masks, one code path for every live thread.  The E partner slots are contiguous but for at most one pad slot
        // (after the entry with (m - q) % E == 0): two base pointers a slot apart, constant offsets.
        if (live) {
            T yp[2][LR_CH];
            int m = c - 1;
            asm volatile("" : "+v"(m));
            const int ml = m & (LR_E - 1);
            const T* const pa = Y + (m + (m >> LOG_E)) - (LR_E - 1);
            const T* const pb = pa - 1;
            auto issue = [&](int ch) {
#pragma unroll
                for (int j = 0; j < LR_CH; ++j) {
                    const int q = ch * LR_CH + j;
                    yp[ch & 1][j] = (q > ml ? pb : pa)[LR_E - 1 - q];
                }
            };
            issue(0);
#pragma unroll
            for (int ch = 0; ch < LR_E / LR_CH; ++ch) {
                if (ch + 1 < LR_E / LR_CH) issue(ch + 1);          // one chunk of partner reads in flight behind the one consumed
#pragma unroll
                for (int j = 0; j < LR_CH; ++j) {
                    const int q = ch * LR_CH + j;
                    const T w = yp[ch & 1][j];
                    if (SOLVE) x[q] = fma_t(mu, w, x[q]);
                    y[q] = fma_t(an, w, y[q]);                     // an = 0 at Levinson's last step
                    if (SOLVE) asm volatile("" : "+v"(x[q]), "+v"(y[q]));   // consumed here: the chunk's partner entries die with the chunk
                    else asm volatile("" : "+v"(y[q]));
                }
                asm volatile("" ::: "memory");
            }
        }
        LV_STAMP(3);
        {                                                          // the slide: rs[e] <- rs[e - 1], r[k] enters at e = 0.  Outside the
            const T up = dpp_move<0x138>(rs[LR_E - 1]);            // wave_shr:1 — lane l takes lane l - 1's (lane 0: from the edge word)               // branch (entries beyond k are zeros sliding onto zeros): a conditional
            const T first = tid == 0 ? rk : (lane == 0 ? (live ? edge[max(wave - 1, 0)] : (T)0) : up);   // rotation costs a second copy of rs
#pragma unroll
            for (int q = LR_E - 1; q > 0; --q) rs[q] = rs[q - 1];
            rs[0] = first;
        }
        alpha = an;
        LV_STAMP(4);
        __syncthreads();                                           // every partner read of the step is done
        LV_STAMP(5);
        if (live && more) {
#pragma unroll
            for (int q = 0; q < LR_E; ++q) yown[q] = y[q];
        }
        LV_STAMP(6);
    }
#pragma unroll
    for (int q = 0; q < LR_E; ++q) {
        const int e = e0 + q;
        if (e < n) { if (SOLVE) xout[e] = x[q]; else yout[e] = y[q]; }
    }
}

// gamma = 1 / (1 + r . y), nu = gamma * reverse(y): one workgroup
template <typename T>
__global__ __launch_bounds__(LV_THREADS) void trench_head_kernel(const T* __restrict__ r, const T* __restrict__ y, int64_t m, T* __restrict__ nu,
                                                                 T* __restrict__ gamma_out) {
    __shared__ T red[2][2][16];
    T s = 0, z = 0;
    for (int64_t i = threadIdx.x; i < m; i += LV_THREADS) s = fma_t(r[i], y[i], s);
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

// the on-chip kernel at the geometry of the size (n <= LR_MAX_N)
template <typename T, bool SOLVE>
static void levinson_reg_launch(hipStream_t st, const T* r, const T* b, T* x, T* y, int64_t n) {
    auto go = [&](auto loge, auto maxt) {
        constexpr int LOG_E = decltype(loge)::value, MAXT = decltype(maxt)::value;
        const unsigned threads = (unsigned)std::min<int64_t>(MAXT, (((n + (1 << LOG_E) - 1) >> LOG_E) + 63) / 64 * 64);
        hipLaunchKernelGGL((levinson_reg_kernel<T, SOLVE, LOG_E, MAXT>), dim3(1), dim3(threads), 0, st, r, b, x, y, (int)n);
    };
    using std::integral_constant;
    if (n <= 8192) go(integral_constant<int, 3>{}, integral_constant<int, 1024>{});
    else if constexpr (sizeof(T) == 8) go(integral_constant<int, 5>{}, integral_constant<int, 512>{});
    else go(integral_constant<int, 4>{}, integral_constant<int, 1024>{});
}

template <typename T>
static int durbin_run(covgram_ctx* ctx, const T* r, int64_t n, T* y) {
    if (n <= LR_MAX_N) levinson_reg_launch<T, false>(ctx->stream, r, (const T*)nullptr, (T*)nullptr, y, n);
    else hipLaunchKernelGGL((levinson_kernel<T, false>), dim3(1), dim3(LV_THREADS), 0, ctx->stream, r, (const T*)nullptr, (T*)nullptr, y, n);
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
#define CG_LEVINSON(T)                                                                                                                \
    {                                                                                                                                  \
        if (n <= LR_MAX_N) levinson_reg_launch<T, true>(ctx->stream, (const T*)rd, (const T*)bd, (T*)xd, (T*)yw, n);                  \
        else hipLaunchKernelGGL((levinson_kernel<T, true>), dim3(1), dim3(LV_THREADS), 0, ctx->stream, (const T*)rd, (const T*)bd, (T*)xd, (T*)yw, n); \
    }
    if (dtype == COVGRAM_F32) CG_LEVINSON(float) else CG_LEVINSON(double)
#undef CG_LEVINSON
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
