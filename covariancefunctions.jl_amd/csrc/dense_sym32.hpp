// dense_sym32.hpp — gramian(k, x) * a in fp32 on the DIRECT-DIFFERENCE path: the upper triangle once.
//
// The reference's mul! follows the data's element type for every kernel (src/gramian.jl:27-33, 78-87): Float32 points give a Float32
// Gramian also for the profiles the matrix-core path refuses — Exponential (src/stationary.jl:60), gamma-exponential (:71), MaternP(0) —
// and for every cloud whose radius fails that path's gate (dense_mfma.hip).  Those ran all n^2 entries on dense_mvm_kernel.  This is the
// fp32 twin of dense_sym_kernel (dense_mvm.hpp), with two changes the packed fp32 arithmetic asks for:
//   * a lane owns R rows (R = 4 for d <= 8, 2 up to 32): the pair body of two packed columns is ~9 instructions, so a cross-lane
//     reduction per 64-row block and column would cost more than the evaluation it saves.  With R rows per lane the column term is
//     first summed inside the lane (one packed fma per evaluated pair of columns and row: c += a_i k_ij) and reduced across the wave
//     once per 64 R rows and column;
//   * EIGHT columns are reduced together: four v_permlane32_swap + adds leave (column 2q in the lower, 2q + 1 in the upper half
//     wave), two v_permlane16_swap + adds leave one column per row of 16 lanes in each of two registers, and four DPP row_shr adds per
//     register finish them (a float add takes the DPP modifier itself): 20 instructions per 8 columns and 64 R rows, lanes 15 / 31 / 47 / 63
//     store.  Fixed order, no float atomics.
// The diagonal block (64 R x 64 R) is evaluated in full for its row sums only (512 / n of the work at R = 4).  Column sums go to
// colslab[local row block][column] (n^2 / (16 R) bytes), dense_sym32_reduce_kernel adds per output row the split-J partials of its row
// block and the column sums of the row blocks above it, accumulating in fp64.  rb_first / rb_stride: covgram_mvm_sym_partial's cyclic
// row blocks of rank r of P.
#pragma once
#include "dense_mvm.hpp"

namespace covgram {

// rows per lane of the fp32 symmetric kernel (host and device agree through this one function)
constexpr int dense_sym32_rows(int D) { return D <= 8 ? 4 : (D <= 32 ? 2 : 1); }

__device__ __forceinline__ float swap32_add_f32(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap16_add_f32(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float row_shr_add_f32(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true);   // bound_ctrl: lanes without a source read 0
    return v + __int_as_float(t);
}
// wave totals of eight columns' per-lane terms c[q] = (column 2q, column 2q + 1): on return lane 15 / 31 / 47 / 63 holds in
// lo the totals of columns 0 / 2 / 1 / 3 and in hi those of columns 4 / 6 / 5 / 7
__device__ __forceinline__ void wave_sum8_f32(const v2f (&c)[4], float& lo, float& hi) {
    float a = swap16_add_f32(swap32_add_f32(c[0].x, c[0].y), swap32_add_f32(c[1].x, c[1].y));
    float b = swap16_add_f32(swap32_add_f32(c[2].x, c[2].y), swap32_add_f32(c[3].x, c[3].y));
    a = row_shr_add_f32<0x111>(a); b = row_shr_add_f32<0x111>(b);
    a = row_shr_add_f32<0x112>(a); b = row_shr_add_f32<0x112>(b);
    a = row_shr_add_f32<0x114>(a); b = row_shr_add_f32<0x114>(b);
    a = row_shr_add_f32<0x118>(a); b = row_shr_add_f32<0x118>(b);
    lo = a; hi = b;
}

template <int FAM, int D, int R>
__global__ __launch_bounds__(DENSE_THREADS) void dense_sym32_kernel(
    const float* __restrict__ X, int64_t n, int32_t d, const v2f* __restrict__ P, float* __restrict__ out,
    float* __restrict__ colslab, int64_t npad, int64_t jchunk, const float* __restrict__ Cn,
    const typename ParamsOf<FAM, float>::type kp0, int32_t rb_first, int32_t rb_stride) {
    using T = float;
    using V = v2f;
    using PK = Pk<float>;
    constexpr bool ISO = fam_is_iso<FAM>;
    using Body = DenseBody<T, FAM, D, 1, R, false, ISO>;
    constexpr int S = D + 1;                               // stream elements (column pairs) per group
    constexpr int BR = DENSE_THREADS * R;                  // rows of a block
    const int lane = threadIdx.x;
    const int64_t row_lo = ((int64_t)rb_first + (int64_t)blockIdx.x * rb_stride) * BR;
    const int64_t n8 = (n + 7) & ~(int64_t)7;              // the stream is padded to whole groups of eight columns (weight 0)
    const int64_t j0 = (int64_t)blockIdx.y * jchunk;       // multiples of 512 (and of BR)
    const int64_t j1 = (j0 + jchunk < n8) ? (j0 + jchunk) : n8;
    if (j1 <= row_lo) {                                    // the whole chunk lies left of the diagonal block
#pragma unroll
        for (int r = 0; r < R; ++r) out[(int64_t)blockIdx.y * npad + row_lo + r * DENSE_THREADS + lane] = 0.f;
        return;
    }
    T x[R][D];
    T ai[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = row_lo + r * DENSE_THREADS + lane;
        const int64_t rowc = (row < n) ? row : n - 1;      // clamp: computed, weighted 0 in the column sums, never used as a row
        const T* xr = X + rowc * (int64_t)d;
#pragma unroll
        for (int l = 0; l < D; ++l) x[r][l] = (l < d) ? (ISO ? xr[l] - Cn[l] : xr[l]) * kp0.gamma : 0.f;
        // the row's own weight: the stream holds (a_2g, a_2g+1) behind the D coordinate pairs of group g
        ai[r] = (row < n) ? ((const float*)(P + (rowc >> 1) * S + D))[rowc & 1] : 0.f;
    }
    T tot[R];
#pragma unroll
    for (int r = 0; r < R; ++r) tot[r] = 0.f;
    const int cmap = ((lane >> 4) & 1) * 2 + (lane >> 5);  // which of four columns this lane's row of 16 ends up holding
    float* __restrict__ cdst = colslab + (int64_t)blockIdx.x * npad;

    // the column sweep as a lambda over the parameter block: MaternP orders p <= 3 run it on a copy whose p the compiler can bound
    // (dense_mvm_kernel does the same), which folds the profile's per-pair "fixed degree or looped Horner" branch away
    auto sweep = [&](const typename ParamsOf<FAM, float>::type& kp) {
        int64_t jb = (j0 > row_lo) ? j0 : row_lo;              // both multiples of BR
        while (jb < j1) {
            int64_t je = (jb / DENSE_INNER + 1) * DENSE_INNER; // two-level accumulation: 512-column inner blocks (dense_mvm.hpp)
            if (je > j1) je = j1;
            V acc[R][1];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r][0] = PK::splat(0.f);
            int64_t j = jb;
            if (j == row_lo) {                                 // diagonal block: all entries, row sums only
                const int64_t jd = (row_lo + BR < je) ? row_lo + BR : je;
                const V* __restrict__ p = P + (j >> 1) * S;
                for (; j < jd; j += 2, p += S) Body::step(p, x, acc, kp);
            }
            const V* __restrict__ p = P + (j >> 1) * S;        // uniform address -> s_load
            for (; j < je; j += 8, p += 4 * S) {               // je - j is a multiple of 8
                V c[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    V s[R];
                    Body::dist(p + q * S, x, s);
                    const V aj = p[q * S + D];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const V kv = PK::map(s[r], [&](T sv) { return dense_phi<FAM, T, false>(sv, kp); });
                        acc[r][0] = PK::fma(aj, kv, acc[r][0]);
                        c[q] = (r == 0) ? PK::splat(ai[0]) * kv : PK::fma(PK::splat(ai[r]), kv, c[q]);
                    }
                }
                float lo, hi;
                wave_sum8_f32(c, lo, hi);
                if ((lane & 15) == 15) { cdst[j + cmap] = lo; cdst[j + 4 + cmap] = hi; }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) tot[r] += PK::hsum(acc[r][0]);
            jb = je;
        }
    };
    if constexpr (FAM == COVGRAM_MATERNP) {
        if (kp0.p == 1 || kp0.p == 2) {
            typename ParamsOf<FAM, T>::type kq = kp0;
            if (kp0.p == 1) { kq.p = 1; sweep(kq); } else { kq.p = 2; sweep(kq); }
        } else if (kp0.p <= 3) {
            typename ParamsOf<FAM, T>::type kq = kp0;
            kq.p = kp0.p & 3;
            sweep(kq);
        } else {
            sweep(kp0);
        }
    } else {
        sweep(kp0);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) out[(int64_t)blockIdx.y * npad + row_lo + r * DENSE_THREADS + lane] = tot[r];
}

// y[i] = alpha * (sum_sp out[sp][i] + sum_{evaluated row blocks rb above block(i)} colslab[rb][i]) + beta * y[i], fp64 accumulators,
// fixed order.  blk_rows = 64 R; slab row x belongs to row block first + x stride (dense_sym_reduce_kernel's layout).
template <typename T /* float */>
__global__ __launch_bounds__(1024) void dense_sym32_reduce_kernel(const float* __restrict__ out, const float* __restrict__ colslab,
                                                                  int64_t npad, int32_t jsplit, float* __restrict__ y, int64_t n,
                                                                  float alpha, float beta, int32_t rb_first, int32_t rb_stride, int32_t blk_rows) {
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    __shared__ double red[16][64];
    double s = 0.0;
    if (i < n) {
        const int64_t B = ((int64_t)blockIdx.x * 64) / blk_rows;
        if (B >= rb_first && (B - rb_first) % rb_stride == 0)
            for (int sp = part; sp < jsplit; sp += 16) s += (double)out[(int64_t)sp * npad + i];
        const int64_t nb = B > rb_first ? (B - rb_first + rb_stride - 1) / rb_stride : 0;
        for (int64_t rb = part; rb < nb; rb += 16) s += (double)colslab[rb * npad + i];
    }
    red[part][lane] = s;
    __syncthreads();
    if (part != 0 || i >= n) return;
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; q += 4) t += (red[q][lane] + red[q + 1][lane]) + (red[q + 2][lane] + red[q + 3][lane]);
    float v = alpha * (float)t;
    if (beta != 0.f) v = __builtin_fmaf(beta, y[i], v);
    y[i] = v;
}

template <int FAM, int D>
static int launch_dense_sym32_one(const DenseArgs& a) {
    static_assert(DENSE_THREADS == 64, "dense_sym32_kernel: one wave per workgroup");
    constexpr int R = dense_sym32_rows(D);
    const typename ParamsOf<FAM, float>::type kp = make_params<FAM, float>(*a.hk);
    const int64_t blocks = (a.n + 64 * R - 1) / (64 * R);
    const int64_t mine = a.sym_first < blocks ? (blocks - a.sym_first + a.sym_stride - 1) / a.sym_stride : 0;
    if (mine == 0) return COVGRAM_OK;
    dim3 grid((unsigned)mine, (unsigned)a.jsplit);
    hipLaunchKernelGGL((dense_sym32_kernel<FAM, D, R>), grid, dim3(DENSE_THREADS), 0, a.stream, (const float*)a.X, a.n, a.d,
                       (const v2f*)a.P, (float*)a.out, (float*)a.colslab, a.npad, a.jchunk, (const float*)a.C, kp, a.sym_first, a.sym_stride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dense_sym32 launch failed: %s", hipGetErrorString(e)); return COVGRAM_EHIP; }
    return COVGRAM_OK;
}

}  // namespace covgram
