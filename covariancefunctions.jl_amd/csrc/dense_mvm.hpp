// dense_mvm.hpp — the north-star kernel: y <- alpha * G(k; X, Y) * A + beta * y for a lazily
// represented n×m Gramian (reference hot loop: src/gramian.jl:78-99, entry src/gramian.jl:37-40,
// squared distance by DIRECT differences src/util.jl:40-47, dot product src/mercer.jl:3).
//
// MI355X mapping:
//   * one LANE owns R output rows (x_i lives in 3R..D*R VGPRs, NRHS*R accumulators): no cross-lane
//     reduction exists anywhere in the pair loop;
//   * the column stream P[j] = (gamma*y_j[0..D), a_j[0..NRHS)) is wave-uniform.  Variant 0 reads it
//     through the scalar data cache (s_load_dwordxN -> SGPR operands of the VALU ops: zero VGPR/LDS
//     cost, the CDNA-native broadcast); variant 1 stages 16 KiB tiles in LDS and reads them back as
//     same-address (broadcast) ds_read_b128;
//   * the grid is (row blocks) × (J splits): every workgroup owns a 256*R-row × jchunk-column
//     rectangle, partial sums go to a [jsplit][NRHS][npad] slab and a tiny second kernel applies
//     alpha/beta (deterministic; no float atomics);
//   * fp32 accumulation is two-level (512-column inner chunks) so the error against the fp64
//     oracle stays ~1e-6 at m = 2^17 (SURVEY §7 "fp32 accumulation").
#pragma once
#include "profiles.hpp"

namespace covgram {

constexpr int DENSE_THREADS = 256;
constexpr int DENSE_INNER = 512;  // inner accumulation chunk (columns)

template <int D, int NR>
constexpr int stride_of() { return D + NR; }

// Dimensions are consumed in chunks of one 64-byte scalar load (16 floats / 8 doubles); for D larger
// than a chunk a scheduling barrier after each chunk keeps hipcc from hoisting every s_load of a column
// to the top, which would overflow the ~100 usable SGPRs and spill them through v_writelane.
template <typename T> constexpr int dim_chunk() { return 64 / (int)sizeof(T); }

template <typename T, int FAM, int D, int NR, int R, bool POW, bool ISO, int JU>
struct DenseBody {
    // One column j against R rows.  p points at the (uniform) packed record of column j.
    template <typename PT>
    static __device__ __forceinline__ void step(const PT* __restrict__ p, const T (&x)[R][D], T (&acc)[R][NR],
                                                const KParams<T>& kp) {
        constexpr int DC = dim_chunk<T>();
        T s[R];
#pragma unroll
        for (int c0 = 0; c0 < D; c0 += DC) {
#pragma unroll
            for (int l = c0; l < ((c0 + DC < D) ? c0 + DC : D); ++l) {
                const T yl = p[l];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if constexpr (ISO) {
                        const T dl = x[r][l] - yl;
                        s[r] = (l == 0) ? dl * dl : cg_fma(dl, dl, s[r]);
                    } else {
                        s[r] = (l == 0) ? x[r][l] * yl : cg_fma(x[r][l], yl, s[r]);
                    }
                }
            }
            if constexpr (D > DC) __builtin_amdgcn_sched_barrier(0);
        }
        T aj[NR];
#pragma unroll
        for (int c = 0; c < NR; ++c) aj[c] = p[D + c];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const T kv = phi_value<FAM, T, (FAM == COVGRAM_EQ), POW>(s[r], kp);
#pragma unroll
            for (int c = 0; c < NR; ++c) acc[r][c] = cg_fma(aj[c], kv, acc[r][c]);
        }
    }
};

// VARIANT 0: scalar-cache stream.  VARIANT 1: LDS-staged tiles.
template <typename T, int FAM, int D, int NR, int R, bool POW, int VARIANT>
__global__ __launch_bounds__(DENSE_THREADS) void dense_mvm_kernel(
    const T* __restrict__ X, int64_t n, int32_t d, const T* __restrict__ P, int64_t m, T* __restrict__ out,
    int64_t npad, int64_t ldy, int32_t nrhs, int64_t jchunk, T alpha, T beta, int32_t final_store,
    const KParams<T> kp) {
    constexpr bool ISO = (FAM != COVGRAM_DOT && FAM != COVGRAM_EXPDOT);
    constexpr int S = D + NR;
    constexpr int W = D * (int)sizeof(T) / 4;   // row width in dwords
    constexpr int JU = (W <= 4) ? 8 : ((W <= 16) ? 4 : ((W <= 32) ? 2 : 1));
    using Body = DenseBody<T, FAM, D, NR, R, POW, ISO, JU>;

    const int tid = threadIdx.x;
    const int64_t row_base = (int64_t)blockIdx.x * (DENSE_THREADS * R);
    const int64_t j0 = (int64_t)blockIdx.y * jchunk;
    const int64_t j1 = (j0 + jchunk < m) ? (j0 + jchunk) : m;

    // rows of this lane: row_base + r*256 + tid  (coalesced across the wave for every r)
    T x[R][D];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t row = row_base + (int64_t)r * DENSE_THREADS + tid;
        if (row >= n) row = n - 1;  // clamp: computed but never stored
        const T* xr = X + row * (int64_t)d;
#pragma unroll
        for (int l = 0; l < D; ++l) x[r][l] = (l < d) ? xr[l] * kp.gamma : (T)0;
    }

    T tot[R][NR];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < NR; ++c) tot[r][c] = (T)0;

    if constexpr (VARIANT == 0) {
        for (int64_t jb = j0; jb < j1; jb += DENSE_INNER) {
            const int64_t je = (jb + DENSE_INNER < j1) ? (jb + DENSE_INNER) : j1;
            T acc[R][NR];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < NR; ++c) acc[r][c] = (T)0;
            const int cnt = (int)(je - jb);
            const T* __restrict__ p = P + jb * S;       // uniform address -> s_load_dwordxN
            int j = 0;
            for (; j + JU <= cnt; j += JU, p += JU * S) {
#pragma unroll
                for (int u = 0; u < JU; ++u) Body::step(p + u * S, x, acc, kp);
            }
            for (; j < cnt; ++j, p += S) Body::step(p, x, acc, kp);
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < NR; ++c) tot[r][c] += acc[r][c];
        }
    } else {
        // LDS-staged: tiles of DENSE_INNER columns, double-buffered, one barrier per tile.
        // TJ columns per tile: 16 KiB per buffer (two buffers), a multiple of JU.
        constexpr int TJ = ((16384 / (S * (int)sizeof(T))) / JU) * JU;
        __shared__ __attribute__((aligned(16))) T tile[2][TJ * S];
        const int64_t ntile = (j1 - j0 + TJ - 1) / TJ;
        auto stage = [&](int buf, int64_t t) {
            const int64_t jb = j0 + t * TJ;
            const int64_t cnt = ((jb + TJ < j1) ? TJ : (j1 - jb)) * S;
            const T* __restrict__ src = P + jb * S;
            for (int64_t e = tid; e < cnt; e += DENSE_THREADS) tile[buf][e] = src[e];
        };
        if (ntile > 0) stage(0, 0);
        __syncthreads();
        for (int64_t t = 0; t < ntile; ++t) {
            const int buf = (int)(t & 1);
            if (t + 1 < ntile) stage(buf ^ 1, t + 1);
            const int64_t jb = j0 + t * TJ;
            const int cnt = (int)((jb + TJ < j1) ? TJ : (j1 - jb));
            T acc[R][NR];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < NR; ++c) acc[r][c] = (T)0;
            const T* tp = tile[buf];
            int j = 0;
            for (; j + JU <= cnt; j += JU) {
#pragma unroll
                for (int u = 0; u < JU; ++u) Body::step(tp + (j + u) * S, x, acc, kp);
            }
            for (; j < cnt; ++j) Body::step(tp + j * S, x, acc, kp);
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < NR; ++c) tot[r][c] += acc[r][c];
            __syncthreads();
        }
    }

    // epilogue ------------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = row_base + (int64_t)r * DENSE_THREADS + tid;
        if (row >= n) continue;
        if (final_store) {   // jsplit == 1: apply alpha/beta here (beta == 0 never reads y)
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                if (c < nrhs) {
                    T* yp = out + row + (int64_t)c * ldy;
                    T v = alpha * tot[r][c];
                    if (beta != (T)0) v = cg_fma(beta, *yp, v);
                    *yp = v;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < NR; ++c)
                out[((int64_t)blockIdx.y * NR + c) * npad + row] = tot[r][c];
        }
    }
}

// y[i + c*ldy] = alpha * sum_s partial[s][c][i] + beta * y   (fixed summation order: deterministic)
template <typename T>
__global__ __launch_bounds__(256) void dense_reduce_kernel(const T* __restrict__ partial, int64_t npad, int32_t NRpad,
                                                           int32_t jsplit, T* __restrict__ y, int64_t n, int64_t ldy,
                                                           int32_t nrhs, T alpha, T beta) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (i >= n || c >= nrhs) return;
    T s = (T)0;
    for (int sp = 0; sp < jsplit; ++sp) s += partial[((int64_t)sp * NRpad + c) * npad + i];
    T* yp = y + i + (int64_t)c * ldy;
    T v = alpha * s;
    if (beta != (T)0) v = cg_fma(beta, *yp, v);
    *yp = v;
}

// P[j][0..D) = gamma * Y[j][0..d) (zero padded), P[j][D..D+NR) = A[j + c*lda] (zero padded)
template <typename T>
__global__ __launch_bounds__(256) void dense_pack_kernel(const T* __restrict__ Y, int64_t m, int32_t d, const T* __restrict__ A,
                                                         int64_t lda, int32_t nrhs, int32_t c0, T* __restrict__ P, int32_t D,
                                                         int32_t NR, T gamma) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    T* p = P + j * (int64_t)(D + NR);
    for (int l = 0; l < D; ++l) p[l] = (l < d) ? Y[j * (int64_t)d + l] * gamma : (T)0;
    for (int c = 0; c < NR; ++c) p[D + c] = (c0 + c < nrhs) ? A[j + (int64_t)(c0 + c) * lda] : (T)0;
}

// -------------------------------------------------------------------------------------------------
// launcher for one family (instantiated per translation unit, see dense_fam.hip)
// -------------------------------------------------------------------------------------------------
template <typename T, int FAM, int D, int NR, int R, bool POW>
static int launch_dense_one(const DenseArgs& a) {
    const KParams<T> kp = cast_params<T>(a.hk->kp);
    const int64_t rows_per_wg = (int64_t)DENSE_THREADS * R;
    dim3 grid((unsigned)((a.n + rows_per_wg - 1) / rows_per_wg), (unsigned)a.jsplit);
    const int final_store = (a.jsplit == 1) ? 1 : 0;
    // the LDS-staged variant is compiled only where it is an A/B candidate (vector RHS, D <= 8)
    constexpr bool HAS_LDS = (NR == 1 && !POW && D <= 8);
    bool launched = false;
    if constexpr (HAS_LDS) {
        if (a.variant == 1) {
            hipLaunchKernelGGL((dense_mvm_kernel<T, FAM, D, NR, R, POW, 1>), grid, dim3(DENSE_THREADS), 0, a.stream,
                               (const T*)a.X, a.n, a.d, (const T*)a.P, a.m, (T*)a.out, a.npad, a.ldy, a.nrhs,
                               a.jchunk, (T)a.alpha, (T)a.beta, final_store, kp);
            launched = true;
        }
    }
    if (!launched)
        hipLaunchKernelGGL((dense_mvm_kernel<T, FAM, D, NR, R, POW, 0>), grid, dim3(DENSE_THREADS), 0, a.stream,
                           (const T*)a.X, a.n, a.d, (const T*)a.P, a.m, (T*)a.out, a.npad, a.ldy, a.nrhs, a.jchunk,
                           (T)a.alpha, (T)a.beta, final_store, kp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dense_mvm launch failed: %s", hipGetErrorString(e)); return COVGRAM_EHIP; }
    return COVGRAM_OK;
}

// rows per lane compiled for each D (register budget: R*(D+NR) VGPRs of state)
template <typename T, int D> struct RowsFor {
    static constexpr int W = D * (int)sizeof(T) / 4;
    static constexpr int value = (W <= 4) ? 4 : ((W <= 16) ? 2 : 1);
};

template <typename T, int FAM, int D, int NR>
static int launch_dense_D(const DenseArgs& a) {
    constexpr int R = RowsFor<T, D>::value;
    const bool pow = a.hk->k.power != 1;
    if (pow) return launch_dense_one<T, FAM, D, NR, R, true>(a);
    return launch_dense_one<T, FAM, D, NR, R, false>(a);
}

template <typename T, int FAM, int NR>
static int launch_dense_NR(const DenseArgs& a) {
    switch (a.Dpad) {
        case 1: return launch_dense_D<T, FAM, 1, NR>(a);
        case 2: return launch_dense_D<T, FAM, 2, NR>(a);
        case 3: return launch_dense_D<T, FAM, 3, NR>(a);
        case 4: return launch_dense_D<T, FAM, 4, NR>(a);
        case 6: return launch_dense_D<T, FAM, 6, NR>(a);
        case 8: return launch_dense_D<T, FAM, 8, NR>(a);
        case 12: return launch_dense_D<T, FAM, 12, NR>(a);
        case 16: return launch_dense_D<T, FAM, 16, NR>(a);
        case 24: return launch_dense_D<T, FAM, 24, NR>(a);
        case 32: return launch_dense_D<T, FAM, 32, NR>(a);
        case 48: return launch_dense_D<T, FAM, 48, NR>(a);
        case 64: return launch_dense_D<T, FAM, 64, NR>(a);
        default: set_error("dense_mvm: padded dimension %d not compiled", a.Dpad); return COVGRAM_EUNSUPPORTED;
    }
}

template <int FAM>
int launch_dense_family(const DenseArgs& a, int dtype) {
    if (dtype == COVGRAM_F32) {
        if (a.NRpad == 1) return launch_dense_NR<float, FAM, 1>(a);
        if (a.NRpad == 4) return launch_dense_NR<float, FAM, 4>(a);
    } else {
        if (a.NRpad == 1) return launch_dense_NR<double, FAM, 1>(a);
        if (a.NRpad == 4) return launch_dense_NR<double, FAM, 4>(a);
    }
    set_error("dense_mvm: nrhs pad %d not compiled", a.NRpad);
    return COVGRAM_EUNSUPPORTED;
}

inline int rows_per_lane_for(int Dpad, int dtype) {
    const int W = Dpad * (dtype == COVGRAM_F64 ? 2 : 1);
    return (W <= 4) ? 4 : ((W <= 16) ? 2 : 1);
}

}  // namespace covgram
