// lowrank.hip — the two places where the Gramian really is a product of thin factors.
//
// Low rank: gramian(k::FiniteBasis, x, y) = LazyMatrixProduct(U, V') (src/mercer.jl:61-70) whose mul! applies the factors
// right to left (src/lazy_linear_algebra.jl:78-85): Y = α U (Vᵀ A) + β Y, U n×r, V m×r, A m×p (column-major).
//   p = 1 (and p < 8, column by column): both products are GEMVs, i.e. HBM streaming of U and V (lowrank_vta / _uz kernels).
//   p >= 8: both products are tall-skinny GEMMs on the MATRIX CORES — the one contraction of the hot path that is a true
//           Phiᵀ Phi (north_star) — with the MFMA of the data's own precision: v_mfma_f32_32x32x2_f32 (bit-for-bit an fp32
//           fmaf chain, MI355X_MICROARCH.md "FP32-input MFMA") and v_mfma_f64_16x16x4_f64, so results keep the accuracy of the
//           GEMV kernels.  Both stay HBM-bound (2 r p / (r + p) flop per streamed scalar).
//
// Factored dot product: Gramian(Dot(), x, y) (src/gramian.jl:23,150-151, src/mercer.jl:6-9) is X Yᵀ; the reference multiplies
// it entry by entry in O(n m d) through the generic loop (src/gramian.jl:78-87), mathematically X (Yᵀ a): two O((n + m) d)
// streaming passes over the point-major point sets themselves (dot_vta / dot_xz kernels), any d, any number of right-hand sides.
#include <algorithm>

#include "common.hpp"
#include "pack.hpp"

namespace covgram {

// ---- low rank: y = alpha U (V^T a) + beta y, two HBM-streaming passes --------------------------------------------------
template <typename T, int VEC> struct VecOf;
template <> struct VecOf<float, 4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct VecOf<double, 2> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct VecOf<float, 1> { typedef float type; };
template <> struct VecOf<double, 1> { typedef double type; };

template <typename T, int VEC>
__device__ __forceinline__ T vdot(typename VecOf<T, VEC>::type u, typename VecOf<T, VEC>::type w, T acc) {
    if constexpr (VEC == 1) return fma_t(u, w, acc);
    else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc = fma_t(u[e], w[e], acc);
        return acc;
    }
}

constexpr int LR_RC = 32;   // columns of V carried per pass over a row slab (accumulators per thread)

// the LAST workgroup to arrive sums the slabs' partials in lowrank_zsum_kernel's fixed order: z[k] = sum_slab zpart[slab][k], k < r (red: 8 x 32 scratch)
template <typename T>
__device__ __forceinline__ void slab_tail_sum(const T* __restrict__ zpart, T* __restrict__ z, int64_t r, T* __restrict__ red) {
    const int64_t nslab = gridDim.x;
    const int kk = threadIdx.x & 31, part = threadIdx.x >> 5;
    T* redz = red;
    for (int64_t kb = 0; kb < r; kb += 32) {
        const int64_t k = kb + kk;
        T s0 = (T)0, s1 = (T)0, s2 = (T)0, s3 = (T)0;
        if (k < r) {
            int64_t sl = part;
            // 32 uncached loads in flight per thread (round 5; 8 before): this tail is pure memory latency — 512 slabs x 32 columns took eight
            // dependent round trips of ~1.2 us behind the last slab (lowrank_vta_kernel 33.6 us against 24.1 for the equally long second pass)
            for (; sl + 248 < nslab; sl += 256) {
                T pv[32];
#pragma unroll
                for (int q = 0; q < 32; ++q) pv[q] = slab_load(zpart + (sl + 8 * q) * r + k);
#pragma unroll
                for (int q = 0; q < 32; q += 4) { s0 += pv[q]; s1 += pv[q + 1]; s2 += pv[q + 2]; s3 += pv[q + 3]; }
            }
            for (; sl + 56 < nslab; sl += 64) {
                const T p0 = slab_load(zpart + sl * r + k), p1 = slab_load(zpart + (sl + 8) * r + k), p2 = slab_load(zpart + (sl + 16) * r + k),
                        p3 = slab_load(zpart + (sl + 24) * r + k), p4 = slab_load(zpart + (sl + 32) * r + k), p5 = slab_load(zpart + (sl + 40) * r + k),
                        p6 = slab_load(zpart + (sl + 48) * r + k), p7 = slab_load(zpart + (sl + 56) * r + k);
                s0 += p0; s1 += p1; s2 += p2; s3 += p3; s0 += p4; s1 += p5; s2 += p6; s3 += p7;
            }
            for (; sl + 24 < nslab; sl += 32) {
                s0 += slab_load(zpart + sl * r + k); s1 += slab_load(zpart + (sl + 8) * r + k); s2 += slab_load(zpart + (sl + 16) * r + k);
                s3 += slab_load(zpart + (sl + 24) * r + k);
            }
            for (; sl < nslab; sl += 8) s0 += slab_load(zpart + sl * r + k);
        }
        __syncthreads();
        redz[part * 32 + kk] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (part == 0 && k < r) {
            T s = (T)0;
#pragma unroll
            for (int p = 0; p < 8; ++p) s += redz[p * 32 + kk];
            z[k] = s;
        }
    }
}

// zpart[slab][k] = sum_{j in slab} V[j + k*ldv] a[j].  One workgroup per row slab; of its four waves, wave w carries columns 8 w .. 8 w + 7 of
// each 32-column pass: a lane streams VEC rows of its 8 columns per step (16-byte loads, 8 accumulators — round 4: 32 accumulators and
// 32 loads per thread left two waves per SIMD and 4.7 TB/s; the short state runs at the occupancy of lowrank_uz_kernel), then a fixed-order
// butterfly over the wave's 64 lanes per column.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void lowrank_vta_kernel(const T* __restrict__ V, int64_t ldv, int64_t m, int64_t r, const T* __restrict__ a,
                                                          T* __restrict__ zpart, int64_t per, unsigned* __restrict__ ticket, T* __restrict__ z) {
    using VT = typename VecOf<T, VEC>::type;
    constexpr int CW = LR_RC / 4;                               // columns per wave
    const int64_t j0 = (int64_t)blockIdx.x * per, j1 = (j0 + per < m) ? (j0 + per) : m;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ T red[8 * 32];                                   // the final sum's scratch (below)
    for (int64_t c0 = 0; c0 < r; c0 += LR_RC) {
        const int64_t cb = c0 + wv * CW;                        // this wave's first column
        const int nc = (int)((r - cb < CW) ? (r - cb < 0 ? 0 : r - cb) : CW);
        T acc[CW];
#pragma unroll
        for (int c = 0; c < CW; ++c) acc[c] = (T)0;
        int64_t j = j0 + (int64_t)lane * VEC;
        // Four row groups per trip, every load issued before the first fma (round 5): a slab is 8-16 groups per wave, and one group per trip made
        // the pass as many dependent memory round trips long at one or two waves per SIMD (32 us at r = 32, n = 2^20 against 24 for the equally
        // long second pass; profiles/r05_lowrank_trace.txt).  Same sums, in the order group 0, 1, 2, 3 per accumulator.  (Giving the waves that
        // have no columns at r <= 16 a share of the rows instead was measured too: no change, not kept.)
        constexpr int UN = 4;
        for (; j + (int64_t)(UN - 1) * 64 * VEC + VEC <= j1; j += (int64_t)UN * 64 * VEC) {
            VT av[UN], vv[UN][CW];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                av[u] = *reinterpret_cast<const VT*>(a + j + (int64_t)u * 64 * VEC);
#pragma unroll
                for (int c = 0; c < CW; ++c)
                    if (c < nc) vv[u][c] = *reinterpret_cast<const VT*>(V + (cb + c) * ldv + j + (int64_t)u * 64 * VEC);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int c = 0; c < CW; ++c)
                    if (c < nc) acc[c] = vdot<T, VEC>(vv[u][c], av[u], acc[c]);
        }
        for (; j < j1; j += 64 * VEC) {
            if (j + VEC <= j1) {
                const VT av = *reinterpret_cast<const VT*>(a + j);
#pragma unroll
                for (int c = 0; c < CW; ++c)
                    if (c < nc) acc[c] = vdot<T, VEC>(*reinterpret_cast<const VT*>(V + (cb + c) * ldv + j), av, acc[c]);
            } else {                                            // ragged end of the matrix
                for (int64_t jj = j; jj < j1; ++jj)
#pragma unroll
                    for (int c = 0; c < CW; ++c)
                        if (c < nc) acc[c] = fma_t(V[(cb + c) * ldv + jj], a[jj], acc[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            T s = acc[c];
            s += __shfl_xor(s, 32); s += __shfl_xor(s, 16); s += __shfl_xor(s, 8); s += __shfl_xor(s, 4); s += __shfl_xor(s, 2); s += __shfl_xor(s, 1);
            if (lane == 0 && c < nc) slab_store(zpart + (int64_t)blockIdx.x * r + cb + c, s, ticket != nullptr);
        }
    }
    // ticket != nullptr (round 4, small r): the LAST slab's workgroup to arrive sums the partials in lowrank_zsum_kernel's fixed order — no
    // separate launch (that kernel is ONE workgroup per 32 columns walking every slab: 9.7 us of latency at r = 32, n = 2^20), eight loads in flight
    if (ticket == nullptr || !last_arrival(ticket, gridDim.x)) return;
    slab_tail_sum<T>(zpart, z, r, red);
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
                                                         T* __restrict__ y, T alpha, T beta, int32_t reverse) {
    using VT = typename VecOf<T, VEC>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* z = reinterpret_cast<T*>(smem);
    for (int64_t k = threadIdx.x; k < r; k += 256) z[k] = zg[k];
    __syncthreads();
    // reverse (U is V, round 5): the first pass has just streamed this matrix front to back, so its LAST rows are the freshest lines of the
    // Infinity Cache — the second pass walks it back to front
    const int64_t blk = reverse ? (int64_t)gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const int64_t i = (blk * 256 + threadIdx.x) * VEC;
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
            for (int64_t k = 0; k < r; ++k) s = fma_t(U[ii + k * ldu], z[k], s);
            T v = alpha * s;
            if (beta != (T)0) v = fma_t(beta, y[ii], v);
            y[ii] = v;
        }
    }
}

template <typename T, int VEC>
static void lowrank_launch(covgram_ctx* ctx, const T* U, int64_t ldu, const T* V, int64_t ldv, int64_t n, int64_t m, int64_t r, const T* a,
                           T* y, T alpha, T beta, T* zpart, T* z, int64_t nslab, int64_t per, unsigned* ticket) {
    hipLaunchKernelGGL((lowrank_vta_kernel<T, VEC>), dim3((unsigned)nslab), dim3(256), 0, ctx->stream, V, ldv, m, r, a, zpart, per, ticket, z);
    if (!ticket) hipLaunchKernelGGL(lowrank_zsum_kernel<T>, dim3((unsigned)((r + 31) / 32)), dim3(256), 0, ctx->stream, (const T*)zpart, nslab, r, z);
    const int64_t rows_per_block = 256 * VEC;
    hipLaunchKernelGGL((lowrank_uz_kernel<T, VEC>), dim3((unsigned)((n + rows_per_block - 1) / rows_per_block)), dim3(256), (size_t)r * sizeof(T),
                       ctx->stream, U, ldu, n, r, (const T*)z, y, alpha, beta,
                       (ctx->lowrank_reverse == 1 || (ctx->lowrank_reverse < 0 && (const void*)U == (const void*)V && (size_t)n * r * sizeof(T) <= ((size_t)192 << 20))) ? 1 : 0);
}

template <typename T>
static void lowrank_run(covgram_ctx* ctx, const void* U, int64_t ldu, const void* V, int64_t ldv, int64_t n, int64_t m, int64_t r, const void* a,
                        void* y, double alpha, double beta, void* zpart, void* z, int64_t nslab, int64_t per, unsigned* ticket) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const bool aligned = (((uintptr_t)U | (uintptr_t)V | (uintptr_t)a | (uintptr_t)y) % 16 == 0) && ldu % VEC == 0 && ldv % VEC == 0 && per % VEC == 0;
    if (aligned) lowrank_launch<T, VEC>(ctx, (const T*)U, ldu, (const T*)V, ldv, n, m, r, (const T*)a, (T*)y, (T)alpha, (T)beta, (T*)zpart, (T*)z, nslab, per, ticket);
    else lowrank_launch<T, 1>(ctx, (const T*)U, ldu, (const T*)V, ldv, n, m, r, (const T*)a, (T*)y, (T)alpha, (T)beta, (T*)zpart, (T*)z, nslab, per, ticket);
}


// ---- matrix right-hand sides on the matrix cores -----------------------------------------------------------------------
// Tile geometry of the MFMA that matches the data type: D (TM x TM) += A (TM x KS) B (KS x TM), one scalar of A and of B per
// lane: A[idx(l)][kk(l)], B[kk(l)][idx(l)]; D: column idx(l), rows row(v, l) for the lane's NREG accumulators
// (cdna_hip_programming.md §3: the f64 map differs from the f32 one).
template <typename T> struct Mt;
template <> struct Mt<float> {
    static constexpr int TM = 32, KS = 2, NREG = 16, PAD = 1;      // PAD: LDS row stride = rows + PAD (conflict-free operand reads)
    typedef float acc_t __attribute__((ext_vector_type(16)));
    static __device__ __forceinline__ int idx(int l) { return l & 31; }
    static __device__ __forceinline__ int kk(int l) { return l >> 5; }
    static __device__ __forceinline__ int row(int v, int l) { return (v & 3) + 8 * (v >> 2) + 4 * (l >> 5); }
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
};
template <> struct Mt<double> {
    static constexpr int TM = 16, KS = 4, NREG = 4, PAD = 2;
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ int idx(int l) { return l & 15; }
    static __device__ __forceinline__ int kk(int l) { return l >> 4; }
    static __device__ __forceinline__ int row(int v, int l) { return (l >> 4) + 4 * v; }
    static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
};

constexpr int LRM_SR = 64;        // rows of V / A staged per step (each of the 4 waves contracts 16 of them)

// zpart[slab][i + j r] = sum_{rows k in slab} V[k][c0 + i] A[k][q0 + j]  for the workgroup's (NI TM) x (NJ TM) block of Z = V' A.
// Stage: 64 rows of the block's columns of V and A go global -> LDS with 16-byte loads along the (contiguous) rows, stored
// [column][row] so that the MFMA operands — one row, 32 (16) different columns per half-wave — read conflict-free; wave w
// contracts rows 16 w .. 16 w + 15 of the stage into its own accumulators; the four waves' tiles meet in LDS at the end
// (fixed order: deterministic).  grid = (row slabs, blocks of r, blocks of p).
template <typename T, int NI, int NJ, int VEC>
__global__ __launch_bounds__(256) void lowrank_vta_mfma_kernel(const T* __restrict__ V, int64_t ldv, int64_t m, int64_t r, const T* __restrict__ A,
                                                               int64_t lda, int32_t p, T* __restrict__ zpart, int64_t per) {
    using M = Mt<T>;
    using VT = typename VecOf<T, VEC>::type;
    constexpr int TM = M::TM, KS = M::KS, NREG = M::NREG, RB = NI * TM, PB = NJ * TM, LW = LRM_SR + M::PAD;
    __shared__ T Vs[RB][LW];
    __shared__ T As[PB][LW];
    __shared__ T red[4][NREG * 64];
    const int64_t c0 = (int64_t)blockIdx.y * RB;
    const int q0 = (int)blockIdx.z * PB;
    const int64_t j0 = (int64_t)blockIdx.x * per, j1 = (j0 + per < m) ? (j0 + per) : m;
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    typename M::acc_t acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int v = 0; v < NREG; ++v) acc[i][j][v] = (T)0;
    constexpr int TPC = LRM_SR / VEC;           // threads along one staged column
    constexpr int CPP = 256 / TPC;              // columns per pass of the workgroup
    const int srow = (t % TPC) * VEC, scol = t / TPC;
    for (int64_t k0 = j0; k0 < j1; k0 += LRM_SR) {
        const int64_t gk = k0 + srow;
#pragma unroll
        for (int q = 0; q < (RB + CPP - 1) / CPP; ++q) {
            const int col = scol + q * CPP;
            if (col < RB) {
                const int64_t gc = c0 + col;
                T vals[VEC];
                if (gc < r && gk + VEC <= j1) {
                    if constexpr (VEC == 1) vals[0] = V[gk + gc * ldv];
                    else { const VT vv = *reinterpret_cast<const VT*>(V + gk + gc * ldv);
#pragma unroll
                           for (int e = 0; e < VEC; ++e) vals[e] = vv[e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) vals[e] = (gc < r && gk + e < j1) ? V[gk + e + gc * ldv] : (T)0;
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) Vs[col][srow + e] = vals[e];
            }
        }
#pragma unroll
        for (int q = 0; q < (PB + CPP - 1) / CPP; ++q) {
            const int col = scol + q * CPP;
            if (col < PB) {
                const int gc = q0 + col;
                T vals[VEC];
                if (gc < p && gk + VEC <= j1) {
                    if constexpr (VEC == 1) vals[0] = A[gk + (int64_t)gc * lda];
                    else { const VT vv = *reinterpret_cast<const VT*>(A + gk + (int64_t)gc * lda);
#pragma unroll
                           for (int e = 0; e < VEC; ++e) vals[e] = vv[e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) vals[e] = (gc < p && gk + e < j1) ? A[gk + e + (int64_t)gc * lda] : (T)0;
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) As[col][srow + e] = vals[e];
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 16 / KS; ++s) {
            const int krow = 16 * w + s * KS + M::kk(l);
            T av[NI], bv[NJ];
#pragma unroll
            for (int i = 0; i < NI; ++i) av[i] = Vs[i * TM + M::idx(l)][krow];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bv[j] = As[j * TM + M::idx(l)][krow];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = M::mfma(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
    T* __restrict__ zp = zpart + (int64_t)blockIdx.x * r * p;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int v = 0; v < NREG; ++v) red[w][v * 64 + l] = acc[i][j][v];
            __syncthreads();
            for (int e = t; e < NREG * 64; e += 256) {
                const T s = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
                const int v = e >> 6, ll = e & 63;
                const int64_t gi = c0 + i * TM + M::row(v, ll);
                const int gj = q0 + j * TM + M::idx(ll);
                if (gi < r && gj < p) zp[gi + (int64_t)gj * r] = s;
            }
            __syncthreads();
        }
}

// Y[i][q0 + j] = alpha sum_k U[i][k] Z[k][q0 + j] + beta Y[i][q0 + j], computed as the TRANSPOSED tile D(j, i) = Z'(j, k) U'(k, i):
// the B operand U'(k, i) is one coalesced row segment of a column of U per half-wave, and a lane's results are one row i,
// NREG right-hand sides -> every store instruction writes 32 (16) consecutive rows of one column of Y.  Z (r x p, column-major,
// ld = r) sits in LDS as [rhs][k] with a stride that keeps the A-operand reads conflict-free.  One row tile per wave.
template <typename T, int NJ>
__global__ __launch_bounds__(256) void lowrank_uz_mfma_kernel(const T* __restrict__ U, int64_t ldu, int64_t n, int64_t r, const T* __restrict__ Z,
                                                              int32_t p, T* __restrict__ y, int64_t ldy, T alpha, T beta, int32_t zs) {
    using M = Mt<T>;
    constexpr int TM = M::TM, KS = M::KS, NREG = M::NREG, PB = NJ * TM;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* Zs = reinterpret_cast<T*>(smem);                         // [PB][zs], zeros beyond r and beyond p
    const int q0 = (int)blockIdx.y * PB;
    for (int e = threadIdx.x; e < PB * zs; e += 256) {
        const int j = e / zs, k = e - j * zs;
        Zs[e] = (k < r && q0 + j < p) ? Z[k + (int64_t)(q0 + j) * r] : (T)0;
    }
    __syncthreads();
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i0 = ((int64_t)blockIdx.x * 4 + w) * TM;
    if (i0 >= n) return;
    const int64_t irow = i0 + M::idx(l);
    const T* __restrict__ up = U + (irow < n ? irow : n - 1);   // clamp: computed, never stored
    typename M::acc_t acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int v = 0; v < NREG; ++v) acc[j][v] = (T)0;
    const int kl = M::kk(l);
#pragma unroll 8
    for (int64_t k0 = 0; k0 < r; k0 += KS) {
        const int64_t k = k0 + kl;
        const T uv = up[(k < r ? k : r - 1) * ldu];              // unconditional load, clamped index; Zs is zero beyond r
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = M::mfma(Zs[(j * TM + M::idx(l)) * zs + k], uv, acc[j]);
    }
    if (irow >= n) return;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int v = 0; v < NREG; ++v) {
            const int jr = q0 + j * TM + M::row(v, l);
            if (jr < p) {
                T* yp = y + irow + (int64_t)jr * ldy;
                T val = alpha * acc[j][v];
                if (beta != (T)0) val = fma_t(beta, *yp, val);
                *yp = val;
            }
        }
}

template <typename T, int NI, int NJ>
static void lrm_vta_launch(covgram_ctx* ctx, bool aligned, dim3 grid, const T* V, int64_t ldv, int64_t m, int64_t r, const T* A, int64_t lda, int p,
                           T* zpart, int64_t per) {
    constexpr int VEC = 16 / (int)sizeof(T);
    if (aligned) hipLaunchKernelGGL((lowrank_vta_mfma_kernel<T, NI, NJ, VEC>), grid, dim3(256), 0, ctx->stream, V, ldv, m, r, A, lda, p, zpart, per);
    else hipLaunchKernelGGL((lowrank_vta_mfma_kernel<T, NI, NJ, 1>), grid, dim3(256), 0, ctx->stream, V, ldv, m, r, A, lda, p, zpart, per);
}

// Y = alpha U (V' A) + beta Y for p >= 8 right-hand sides (device pointers).  ws: zpart [nslab][r p] then Z [r p].
template <typename T>
static int lowrank_mfma_run(covgram_ctx* ctx, const T* U, int64_t ldu, const T* V, int64_t ldv, int64_t n, int64_t m, int64_t r, const T* A,
                            int64_t lda, T* y, int64_t ldy, int p, T alpha, T beta, T* zpart, T* Z, int64_t nslab, int64_t per) {
    using M = Mt<T>;
    constexpr int TM = M::TM, VEC = 16 / (int)sizeof(T);
    const bool aligned = (((uintptr_t)V | (uintptr_t)A) % 16 == 0) && ldv % VEC == 0 && lda % VEC == 0;   // per is a multiple of 64 rows
    // block of Z per workgroup: NI x NJ tiles, at most 8 (fp32) / 16 (fp64) accumulator tiles (<= 128 VGPRs)
    const int NJ = p <= TM ? 1 : (sizeof(T) == 8 && p > 2 * TM ? 4 : 2);
    const int NI = sizeof(T) == 4 ? (r <= 32 ? 1 : (r <= 64 ? 2 : 4)) : (r <= 32 ? 2 : 4);
    const dim3 grid((unsigned)nslab, (unsigned)((r + NI * TM - 1) / (NI * TM)), (unsigned)((p + NJ * TM - 1) / (NJ * TM)));
#define CG_LRM(I, J) lrm_vta_launch<T, I, J>(ctx, aligned, grid, V, ldv, m, r, A, lda, p, zpart, per)
    if constexpr (sizeof(T) == 4) {
        if (NJ == 1) { if (NI == 1) CG_LRM(1, 1); else if (NI == 2) CG_LRM(2, 1); else CG_LRM(4, 1); }
        else { if (NI == 1) CG_LRM(1, 2); else if (NI == 2) CG_LRM(2, 2); else CG_LRM(4, 2); }
    } else {
        if (NJ == 1) { if (NI == 2) CG_LRM(2, 1); else CG_LRM(4, 1); }
        else if (NJ == 2) { if (NI == 2) CG_LRM(2, 2); else CG_LRM(4, 2); }
        else { if (NI == 2) CG_LRM(2, 4); else CG_LRM(4, 4); }
    }
#undef CG_LRM
    const int64_t rp = r * (int64_t)p;
    hipLaunchKernelGGL(lowrank_zsum_kernel<T>, dim3((unsigned)((rp + 31) / 32)), dim3(256), 0, ctx->stream, (const T*)zpart, nslab, rp, Z);
    // second product: Z block in LDS (<= 64 KB), row stride odd (fp32) / = 2 mod 32 (fp64) and >= r rounded up to the MFMA's K
    const int kpad = (int)((r + M::KS - 1) / M::KS * M::KS);
    int zs = sizeof(T) == 4 ? (kpad | 1) : ((kpad + 31) / 32 * 32 + 2);
    int NJ2 = p <= TM ? 1 : 2;
    if ((size_t)NJ2 * TM * zs * sizeof(T) > 65536) NJ2 = 1;
    const size_t lds = (size_t)NJ2 * TM * zs * sizeof(T);
    const dim3 g2((unsigned)((n + 4 * TM - 1) / (4 * TM)), (unsigned)((p + NJ2 * TM - 1) / (NJ2 * TM)));
    if (NJ2 == 1) hipLaunchKernelGGL((lowrank_uz_mfma_kernel<T, 1>), g2, dim3(256), lds, ctx->stream, U, ldu, n, r, (const T*)Z, p, y, ldy, alpha, beta, zs);
    else hipLaunchKernelGGL((lowrank_uz_mfma_kernel<T, 2>), g2, dim3(256), lds, ctx->stream, U, ldu, n, r, (const T*)Z, p, y, ldy, alpha, beta, zs);
    return COVGRAM_OK;
}

// largest r the matrix-core path serves: one rhs tile of Z must fit 64 KB of LDS
template <typename T> static bool lowrank_mfma_fits(int64_t r) { return (size_t)Mt<T>::TM * (size_t)(r + 34) * sizeof(T) <= 65536; }

// ---- factored dot product: y = alpha X (Y' a) + beta y on the point-major point sets ------------------------------------
// zpart[slab][c + q d] = sum_{j in slab} Y[j][c] a[j + q lda]: a thread walks rows j = t, t + 256, ... of the slab and keeps
// DC x NQ partial sums (coordinate chunk x right-hand sides), then the block reduces them in fixed order.
constexpr int DOT_DC = 16, DOT_NQ = 4;
template <typename T>
__global__ __launch_bounds__(256) void dot_vta_kernel(const T* __restrict__ Y, int64_t m, int32_t d, const T* __restrict__ a, int64_t lda, int32_t nrhs,
                                                      T* __restrict__ zpart, int64_t per) {
    const int64_t j0 = (int64_t)blockIdx.x * per, j1 = (j0 + per < m) ? (j0 + per) : m;
    __shared__ T red[4][DOT_DC * DOT_NQ];
    T* __restrict__ zp = zpart + (int64_t)blockIdx.x * d * nrhs;
    for (int q0 = 0; q0 < nrhs; q0 += DOT_NQ)
        for (int c0 = 0; c0 < d; c0 += DOT_DC) {
            T acc[DOT_DC][DOT_NQ];
#pragma unroll
            for (int c = 0; c < DOT_DC; ++c)
#pragma unroll
                for (int q = 0; q < DOT_NQ; ++q) acc[c][q] = (T)0;
            for (int64_t j = j0 + threadIdx.x; j < j1; j += 256) {
                T av[DOT_NQ];
#pragma unroll
                for (int q = 0; q < DOT_NQ; ++q) av[q] = a[j + (int64_t)(q0 + q < nrhs ? q0 + q : nrhs - 1) * lda];
#pragma unroll
                for (int c = 0; c < DOT_DC; ++c) {
                    const T yv = Y[j * (int64_t)d + (c0 + c < d ? c0 + c : d - 1)];
#pragma unroll
                    for (int q = 0; q < DOT_NQ; ++q) acc[c][q] = fma_t(yv, av[q], acc[c][q]);
                }
            }
#pragma unroll
            for (int c = 0; c < DOT_DC; ++c)
#pragma unroll
                for (int q = 0; q < DOT_NQ; ++q) {
                    T s = acc[c][q];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
                    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c * DOT_NQ + q] = s;
                }
            __syncthreads();
            if (threadIdx.x < DOT_DC * DOT_NQ) {
                const int c = threadIdx.x / DOT_NQ, q = threadIdx.x % DOT_NQ;
                if (c0 + c < d && q0 + q < nrhs)
                    zp[(c0 + c) + (int64_t)(q0 + q) * d] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
            }
            __syncthreads();
        }
}

// One right-hand side (round 5): zpart[slab][c] = sum_{j in slab} Y[j][c] a[j] with DC >= min(d, 16) coordinates per pass chosen by the host (4, 8, 16:
// the generic kernel above carries 16 x 4 sums whatever d and nrhs are — 16 loads and 64 fmas per row where d = 3 needs 3 and 3, 51 us for n = 2^20,
// d = 8), four rows per thread in flight with every load issued before the first fma, and the last slab's workgroup sums the partials (no
// separate zsum launch: 9.6 us of one-workgroup latency).
template <typename T, int DC, bool VEC>
__global__ __launch_bounds__(256) void dot_vta1_kernel(const T* __restrict__ Y, int64_t m, int32_t d, const T* __restrict__ a, T* __restrict__ zpart,
                                                       int64_t per, unsigned* __restrict__ ticket, T* __restrict__ z) {
    // VEC: rows are whole 16-byte vectors (d a multiple of 4 fp32 / 2 fp64, aligned base) — a quarter / half of the load instructions
    constexpr int VW = 16 / (int)sizeof(T), NV = DC / VW;
    using VT = typename VecOf<T, VW>::type;
    constexpr int UN = (DC * (int)sizeof(T) > 64) ? 2 : 4;      // rows in flight per thread: at most 64 registers of operands
    const int64_t j0 = (int64_t)blockIdx.x * per, j1 = (j0 + per < m) ? (j0 + per) : m;
    __shared__ T red[8 * 32];
    T* __restrict__ zp = zpart + (int64_t)blockIdx.x * d;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c0 = 0; c0 < d; c0 += DC) {
        T acc[DC];
#pragma unroll
        for (int c = 0; c < DC; ++c) acc[c] = (T)0;
        int64_t j = j0 + threadIdx.x;
        if constexpr (VEC) {
            const int nv = (d - c0 < DC ? d - c0 : DC) / VW;    // whole vectors of this pass that exist (d is a multiple of VW)
            for (; j + (UN - 1) * 256 < j1; j += UN * 256) {
                T av[UN]; VT yv[UN][NV];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    av[u] = a[j + u * 256];
                    const VT* __restrict__ yr = reinterpret_cast<const VT*>(Y + (j + u * 256) * (int64_t)d + c0);
#pragma unroll
                    for (int v = 0; v < NV; ++v) yv[u][v] = yr[v < nv ? v : nv - 1];
                }
#pragma unroll
                for (int u = 0; u < UN; ++u)
#pragma unroll
                    for (int v = 0; v < NV; ++v)
#pragma unroll
                        for (int e = 0; e < VW; ++e) acc[v * VW + e] = fma_t(yv[u][v][e], av[u], acc[v * VW + e]);
            }
        } else {
            for (; j + (UN - 1) * 256 < j1; j += UN * 256) {
                T av[UN], yv[UN][DC];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    av[u] = a[j + u * 256];
                    const T* __restrict__ yr = Y + (j + u * 256) * (int64_t)d;
#pragma unroll
                    for (int c = 0; c < DC; ++c) yv[u][c] = yr[c0 + c < d ? c0 + c : d - 1];
                }
#pragma unroll
                for (int u = 0; u < UN; ++u)
#pragma unroll
                    for (int c = 0; c < DC; ++c) acc[c] = fma_t(yv[u][c], av[u], acc[c]);
            }
        }
        for (; j < j1; j += 256) {
            const T av = a[j];
            const T* __restrict__ yr = Y + j * (int64_t)d;
#pragma unroll
            for (int c = 0; c < DC; ++c) acc[c] = fma_t(yr[c0 + c < d ? c0 + c : d - 1], av, acc[c]);
        }
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            T s = acc[c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) red[wv * DC + c] = s;
        }
        __syncthreads();
        if (threadIdx.x < DC && c0 + (int)threadIdx.x < d)
            slab_store(zp + c0 + threadIdx.x, (red[threadIdx.x] + red[DC + threadIdx.x]) + (red[2 * DC + threadIdx.x] + red[3 * DC + threadIdx.x]), ticket != nullptr);
        __syncthreads();
    }
    if (ticket == nullptr || !last_arrival(ticket, gridDim.x)) return;
    slab_tail_sum<T>(zpart, z, (int64_t)d, red);
}

// Rows longer than 64 bytes (round 5): G = d / VW lanes share a row, one 16-byte vector each — a wave's load is 1 KB of consecutive memory.  With a
// thread per row such rows put every lane of a load on its own cache line, 16 lines per lane and pass: d = 32 fp64 ran at 0.4 TB/s.
// d = G VW <= 64, G a power of two; one right-hand side.
template <typename T, int G>
__global__ __launch_bounds__(256) void dot_vtag_kernel(const T* __restrict__ Y, int64_t m, const T* __restrict__ a, T* __restrict__ zpart, int64_t per,
                                                       unsigned* __restrict__ ticket, T* __restrict__ z) {
    constexpr int VW = 16 / (int)sizeof(T), RPW = 64 / G, D = G * VW, UN = 8;
    using VT = typename VecOf<T, VW>::type;
    const int64_t j0 = (int64_t)blockIdx.x * per, j1 = (j0 + per < m) ? (j0 + per) : m;
    __shared__ T red[8 * 32];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane % G, rr = lane / G;
    T acc[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) acc[e] = (T)0;
    int64_t j = j0 + wv * RPW + rr;
    for (; j + (int64_t)(UN - 1) * 4 * RPW < j1; j += (int64_t)UN * 4 * RPW) {
        T av[UN]; VT yv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) { av[u] = a[j + u * 4 * RPW]; yv[u] = *reinterpret_cast<const VT*>(Y + (j + u * 4 * RPW) * (int64_t)D + g * VW); }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int e = 0; e < VW; ++e) acc[e] = fma_t(yv[u][e], av[u], acc[e]);
    }
    for (; j < j1; j += 4 * RPW) {
        const T av = a[j]; const VT yv = *reinterpret_cast<const VT*>(Y + j * (int64_t)D + g * VW);
#pragma unroll
        for (int e = 0; e < VW; ++e) acc[e] = fma_t(yv[e], av, acc[e]);
    }
#pragma unroll
    for (int e = 0; e < VW; ++e) {
        T s = acc[e];
#pragma unroll
        for (int o = 32; o >= G; o >>= 1) s += __shfl_xor(s, o);          // the lanes that hold the same vector of other rows
        if (lane < G) red[wv * D + g * VW + e] = s;
    }
    __syncthreads();
    if (threadIdx.x < D)
        slab_store(zpart + (int64_t)blockIdx.x * D + threadIdx.x, (red[threadIdx.x] + red[D + threadIdx.x]) + (red[2 * D + threadIdx.x] + red[3 * D + threadIdx.x]), ticket != nullptr);
    __syncthreads();
    if (ticket == nullptr || !last_arrival(ticket, gridDim.x)) return;
    slab_tail_sum<T>(zpart, z, (int64_t)D, red);
}

// y[i] = alpha sum_c X[i][c] z[c] + beta y[i] in the same lane-group layout: a lane keeps its vector of z, KR rows per lane group and block
template <typename T, int G>
__global__ __launch_bounds__(256) void dot_xzg_kernel(const T* __restrict__ X, int64_t n, const T* __restrict__ zg, T* __restrict__ y, T alpha, T beta) {
    constexpr int VW = 16 / (int)sizeof(T), RPB = 256 / G, D = G * VW, KR = 8;
    using VT = typename VecOf<T, VW>::type;
    const int g = threadIdx.x % G, rr = threadIdx.x / G;
    const VT zv = *reinterpret_cast<const VT*>(zg + g * VW);
    const int64_t i0 = (int64_t)blockIdx.x * RPB * KR + rr;
    VT xv[KR];
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        const int64_t i = i0 + (int64_t)k * RPB;
        xv[k] = *reinterpret_cast<const VT*>(X + (i < n ? i : n - 1) * (int64_t)D + g * VW);
    }
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        T s = (T)0;
#pragma unroll
        for (int e = 0; e < VW; ++e) s = fma_t(xv[k][e], zv[e], s);
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const int64_t i = i0 + (int64_t)k * RPB;
        if (g == 0 && i < n) {
            T v = alpha * s;
            if (beta != (T)0) v = fma_t(beta, y[i], v);
            y[i] = v;
        }
    }
}

// y[i + q ldy] = alpha sum_c X[i][c] z[c + q d] + beta y[i + q ldy]: a thread per row, z (d x nrhs) in LDS
template <typename T>
__global__ __launch_bounds__(256) void dot_xz_kernel(const T* __restrict__ X, int64_t n, int32_t d, const T* __restrict__ zg, int32_t nrhs,
                                                     T* __restrict__ y, int64_t ldy, T alpha, T beta) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* z = reinterpret_cast<T*>(smem);
    for (int e = threadIdx.x; e < d * nrhs; e += 256) z[e] = zg[e];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const T* __restrict__ xr = X + i * (int64_t)d;
    for (int q0 = 0; q0 < nrhs; q0 += DOT_NQ) {
        T s[DOT_NQ];
#pragma unroll
        for (int q = 0; q < DOT_NQ; ++q) s[q] = (T)0;
        for (int c = 0; c < d; ++c) {
            const T xv = xr[c];
#pragma unroll
            for (int q = 0; q < DOT_NQ; ++q) s[q] = fma_t(xv, z[c + (q0 + q < nrhs ? q0 + q : nrhs - 1) * d], s[q]);
        }
#pragma unroll
        for (int q = 0; q < DOT_NQ; ++q)
            if (q0 + q < nrhs) {
                T* yp = y + i + (int64_t)(q0 + q) * ldy;
                T v = alpha * s[q];
                if (beta != (T)0) v = fma_t(beta, *yp, v);
                *yp = v;
            }
    }
}

template <typename T>
static int dot_factored_run(covgram_ctx* ctx, const covgram_points* X, const covgram_points* Y, const T* a, int64_t lda, T* y, int64_t ldy, int nrhs,
                            T alpha, T beta) {
    const int64_t n = X->n, m = Y->n;
    const int d = X->d;
    const int64_t slabs_per_cu = nrhs == 1 ? 2 : 4;            // one rhs: few long slabs (four rows per thread in flight; the last workgroup adds one partial per slab)
    int64_t per = (m + (int64_t)ctx->num_cus * slabs_per_cu - 1) / ((int64_t)ctx->num_cus * slabs_per_cu);
    per = std::max<int64_t>(1024, ((per + 1023) / 1024) * 1024);
    const int64_t nslab = (m + per - 1) / per;
    const int64_t dq = (int64_t)d * nrhs;
    const size_t zoff = ((size_t)nslab * dq + 63) & ~(size_t)63;
    void* w;
    int rc = ws_reserve(ctx, 1, (zoff + (size_t)dq) * sizeof(T), &w);
    if (rc) return rc;
    T* zpart = (T*)w;
    T* z = zpart + zoff;
    CG_REQUIRE((size_t)dq * sizeof(T) <= 65536, COVGRAM_EUNSUPPORTED, "factored dot product: d * nrhs = %lld exceeds the 64 KB of LDS", (long long)dq);
    constexpr int VW = 16 / (int)sizeof(T);
    const bool vecY = d % VW == 0 && ((uintptr_t)Y->dptr % 16) == 0, vecX = d % VW == 0 && ((uintptr_t)X->dptr % 16) == 0;
    const int G = d / VW;                                         // lanes per row of the lane-group kernels: rows of more than 64 bytes, d <= 64
    const bool grp = nrhs == 1 && d % VW == 0 && (G & (G - 1)) == 0 && G >= 2 && d <= 64;
    bool xdone = false;
    if (nrhs == 1 && d <= 128) {
        unsigned* ticket = nullptr;
        rc = tickets_reserve(ctx, 1, &ticket);
        if (rc) return rc;
#define CG_DOTG(GV) do { if (grp && vecY && G == GV) { hipLaunchKernelGGL((dot_vtag_kernel<T, GV>), dim3((unsigned)nslab), dim3(256), 0, ctx->stream, (const T*)Y->dptr, m, a, zpart, per, ticket, z); ydone = true; } } while (0)
#define CG_DOT1(DCV) do { if (vecY) hipLaunchKernelGGL((dot_vta1_kernel<T, DCV, true>), dim3((unsigned)nslab), dim3(256), 0, ctx->stream, (const T*)Y->dptr, m, d, a, zpart, per, ticket, z); \
                          else hipLaunchKernelGGL((dot_vta1_kernel<T, DCV, false>), dim3((unsigned)nslab), dim3(256), 0, ctx->stream, (const T*)Y->dptr, m, d, a, zpart, per, ticket, z); } while (0)
        bool ydone = false;
        CG_DOTG(2); CG_DOTG(4); CG_DOTG(8); CG_DOTG(16); CG_DOTG(32);
        if (!ydone) { if (d <= 4) CG_DOT1(4); else if (d <= 8) CG_DOT1(8); else CG_DOT1(16); }
#undef CG_DOT1
#undef CG_DOTG
#define CG_XZG(GV) do { if (grp && vecX && G == GV) { hipLaunchKernelGGL((dot_xzg_kernel<T, GV>), dim3((unsigned)((n + (256 / GV) * 8 - 1) / ((256 / GV) * 8))), dim3(256), 0, ctx->stream, (const T*)X->dptr, n, (const T*)z, y, alpha, beta); xdone = true; } } while (0)
        CG_XZG(2); CG_XZG(4); CG_XZG(8); CG_XZG(16); CG_XZG(32);
#undef CG_XZG
    } else {
        hipLaunchKernelGGL(dot_vta_kernel<T>, dim3((unsigned)nslab), dim3(256), 0, ctx->stream, (const T*)Y->dptr, m, d, a, lda, nrhs, zpart, per);
        hipLaunchKernelGGL(lowrank_zsum_kernel<T>, dim3((unsigned)((dq + 31) / 32)), dim3(256), 0, ctx->stream, (const T*)zpart, nslab, dq, z);
    }
    if (!xdone)
    hipLaunchKernelGGL(dot_xz_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), (size_t)dq * sizeof(T), ctx->stream, (const T*)X->dptr, n, d,
                       (const T*)z, nrhs, y, ldy, alpha, beta);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("factored dot-product kernels failed: %s", hipGetErrorString(e)); return COVGRAM_EHIP; }
    return COVGRAM_OK;
}

// y <- alpha * scale * X (Y' a) + beta * y for Gramian(Dot(), x, y) (device pointers, column-major a / y)
int mvm_dot_factored(covgram_ctx* ctx, const HostKernel& hk, const covgram_points* X, const covgram_points* Y, const void* a, int64_t lda, void* y,
                     int64_t ldy, int32_t nrhs, double alpha, double beta) {
    const double al = alpha * hk.kp.scale;
    if (X->dtype == COVGRAM_F32) return dot_factored_run<float>(ctx, X, Y, (const float*)a, lda, (float*)y, ldy, nrhs, (float)al, (float)beta);
    return dot_factored_run<double>(ctx, X, Y, (const double*)a, lda, (double*)y, ldy, nrhs, al, beta);
}

}  // namespace covgram

using namespace covgram;

extern "C" {

int covgram_lowrank_mvm(covgram_ctx* ctx, const void* U, int64_t ldu, const void* V, int64_t ldv, int64_t n, int64_t m, int64_t r,
                        int32_t dtype, const void* a, int64_t lda, void* y, int64_t ldy, int32_t nrhs, double alpha, double beta,
                        int32_t loc) {
    CG_REQUIRE(ctx && U && V && a && y, COVGRAM_EINVAL, "NULL argument");
    CG_REQUIRE(n >= 1 && m >= 1 && r >= 1 && ldu >= n && ldv >= m, COVGRAM_EINVAL, "lowrank: bad shape");
    CG_REQUIRE(nrhs >= 1 && lda >= m && ldy >= n, COVGRAM_EINVAL, "DimensionMismatch: lowrank: nrhs=%d lda=%lld (m=%lld) ldy=%lld (n=%lld)",
               nrhs, (long long)lda, (long long)m, (long long)ldy, (long long)n);
    CG_REQUIRE(dtype == COVGRAM_F32 || dtype == COVGRAM_F64, COVGRAM_EINVAL, "unknown dtype %d", dtype);
    CG_REQUIRE(r <= 8192, COVGRAM_EUNSUPPORTED, "lowrank: r = %lld exceeds 8192", (long long)r);
    const size_t ts = dtype_size(dtype);
    CG_DEVICE(ctx);
    const bool mfma = nrhs >= 8 && (dtype == COVGRAM_F32 ? lowrank_mfma_fits<float>(r) : lowrank_mfma_fits<double>(r));
    const int pz = mfma ? nrhs : 1;                              // columns of Z held at once
    // row slabs of V: ~4 workgroups per CU (GEMV form: ~2, i.e. at least two sweeps per slab at n = 2^20 — a slab of ONE sweep spends as long in
    // its 32-column LDS reduction as on its loads: 28.4 us at 4.7 TB/s for 134 MB), slab length a multiple of one sweep of the block (256 threads x 16 bytes)
    const int64_t sweep = 256 * (16 / (int64_t)ts);
    const int64_t wg_per_cu = ctx->lowrank_wgs > 0 ? ctx->lowrank_wgs : (mfma ? 4 : 1);      // GEMV form, round 5 (four row groups in flight per wave; tools/lowrank_wg_ab.py, r = 32 fp32): 1: 48.9 us, 2: 52.1, 4: 60.1 — the last workgroup adds one partial per slab and column
    int64_t per = (m + (int64_t)ctx->num_cus * wg_per_cu - 1) / ((int64_t)ctx->num_cus * wg_per_cu);
    per = std::max<int64_t>(sweep, ((per + sweep - 1) / sweep) * sweep);
    const int64_t nslab = (m + per - 1) / per;
    size_t need = ((size_t)nslab * r * pz + (size_t)r * pz) * ts + 512;
    if (loc == COVGRAM_HOST) need += ((size_t)n * r + (size_t)m * r + ((size_t)m + n) * nrhs) * ts + 64;
    void* w; int rc = ws_reserve(ctx, 1, need, &w); if (rc) return rc;
    char* p = (char*)w;
    void* zpart = p; p += (((size_t)nslab * r * pz * ts) + 255) & ~(size_t)255;
    void* z = p; p += (((size_t)r * pz * ts) + 255) & ~(size_t)255;
    const void *Ud = U, *Vd = V, *ad = a; void* yd = y; int64_t ldud = ldu, ldvd = ldv, ldad = lda, ldyd = ldy;
    if (loc == COVGRAM_HOST) {
        CG_CHECK_HIP(hipMemcpy2DAsync(p, (size_t)n * ts, U, (size_t)ldu * ts, (size_t)n * ts, r, hipMemcpyHostToDevice, ctx->stream)); Ud = p; ldud = n; p += (size_t)n * r * ts;
        CG_CHECK_HIP(hipMemcpy2DAsync(p, (size_t)m * ts, V, (size_t)ldv * ts, (size_t)m * ts, r, hipMemcpyHostToDevice, ctx->stream)); Vd = p; ldvd = m; p += (size_t)m * r * ts;
        CG_CHECK_HIP(hipMemcpy2DAsync(p, (size_t)m * ts, a, (size_t)lda * ts, (size_t)m * ts, nrhs, hipMemcpyHostToDevice, ctx->stream)); ad = p; ldad = m; p += (size_t)m * nrhs * ts;
        if (beta != 0.0) CG_CHECK_HIP(hipMemcpy2DAsync(p, (size_t)n * ts, y, (size_t)ldy * ts, (size_t)n * ts, nrhs, hipMemcpyHostToDevice, ctx->stream));
        yd = p; ldyd = n;
    }
    if (mfma) {
        if (dtype == COVGRAM_F32)
            rc = lowrank_mfma_run<float>(ctx, (const float*)Ud, ldud, (const float*)Vd, ldvd, n, m, r, (const float*)ad, ldad, (float*)yd, ldyd, nrhs,
                                         (float)alpha, (float)beta, (float*)zpart, (float*)z, nslab, per);
        else
            rc = lowrank_mfma_run<double>(ctx, (const double*)Ud, ldud, (const double*)Vd, ldvd, n, m, r, (const double*)ad, ldad, (double*)yd, ldyd,
                                          nrhs, alpha, beta, (double*)zpart, (double*)z, nslab, per);
        if (rc) return rc;
    } else {
        // small r: the last slab's workgroup sums the partials inside lowrank_vta_kernel (one counter, zero between launches); a long z keeps
        // the separate one-workgroup-per-32-columns kernel
        unsigned* ticket = nullptr;
        if (r <= 128) { rc = tickets_reserve(ctx, 1, &ticket); if (rc) return rc; }
        for (int c = 0; c < nrhs; ++c) {                         // GEMV pair per column
            const char* ac = (const char*)ad + (size_t)c * ldad * ts;
            char* yc = (char*)yd + (size_t)c * ldyd * ts;
            if (dtype == COVGRAM_F32) lowrank_run<float>(ctx, Ud, ldud, Vd, ldvd, n, m, r, ac, yc, alpha, beta, zpart, z, nslab, per, ticket);
            else lowrank_run<double>(ctx, Ud, ldud, Vd, ldvd, n, m, r, ac, yc, alpha, beta, zpart, z, nslab, per, ticket);
        }
    }
    CG_CHECK_HIP(hipGetLastError());
    if (loc == COVGRAM_HOST) {
        CG_CHECK_HIP(hipMemcpy2DAsync(y, (size_t)ldy * ts, yd, (size_t)n * ts, (size_t)n * ts, nrhs, hipMemcpyDeviceToHost, ctx->stream));
        CG_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return COVGRAM_OK;
}

}  // extern "C"
