// dense_wide.hpp — the dense Gramian MVM for point dimensions beyond the register-resident set (d > 64; any d works).
//
// Same mapping as dense_mvm.hpp — one lane owns one output row, the column stream arrives through the scalar data cache
// as SGPR operands, fp32 packs two columns per instruction — but x_i no longer fits in VGPRs, so the dimension is walked
// in chunks of 32 coordinates and a BLOCK of 2*JG columns is carried through all chunks at once:
//     for each column block:   s[g] = 0                                     (JG packed accumulators in VGPRs)
//         for each chunk c:    load x_i[c] (32 coordinates, this lane's own row: 128 contiguous bytes)
//             for each group g: s[g] += sum_l (x_il - y_gl)^2  /  x_il * y_gl   (y from SGPRs, 16 dims per s_load batch)
//         for each group g:    acc += a_g * phi(s[g])
// Per pair this is the same d subtractions + d FMAs (direct differences, src/util.jl:40-47) as the reference; the row is
// re-read once per column block (from L2 / Infinity Cache: X is n*d*sizeof(T), e.g. 64 MB at n = 16384, d = 1024).
#pragma once
#include "dense_mvm.hpp"

namespace covgram {

constexpr int WIDE_CH = 32;    // coordinates of x_i held in registers at a time
constexpr int WIDE_SB = 16;    // coordinates per scalar-load batch (bounds SGPR live ranges)

template <typename T> constexpr int wide_jg() { return 32; }   // column groups per block (fp32: 64 columns, fp64: 32)

// Blocked column stream for the wide kernel.  Column block b holds JG groups (2*JG columns for fp32) and is stored as
//     [chunk ch][group g][coordinate l]  (nch * JG * 32 packed values)   followed by   [group g][rhs c]  (JG * NR weights)
// so that inside a (block, chunk) every operand sits at a COMPILE-TIME offset from one scalar base address.  Blocks are
// always full: columns past m repeat the last point with weight 0 (they add exactly 0 * phi(finite)).
template <typename T>
__global__ __launch_bounds__(256) void dense_wide_pack_kernel(const T* __restrict__ Y, int64_t m, int32_t d, int32_t dpad,
                                                              const T* __restrict__ A, int64_t lda, int32_t nrhs, int32_t c0,
                                                              T* __restrict__ P, int32_t NR, int32_t PKN, int32_t JG, T gamma,
                                                              const T* __restrict__ Cn) {
    const int64_t cols_per_blk = (int64_t)JG * PKN;
    const int64_t nblk = (m + cols_per_blk - 1) / cols_per_blk;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // (padded column, coordinate-or-weight slot)
    const int64_t slots = (int64_t)dpad + NR;
    if (e >= nblk * cols_per_blk * slots) return;
    const int64_t jj = e / slots;
    const int sl = (int)(e - jj * slots);
    const bool pad = jj >= m;
    const int64_t j = pad ? (m - 1) : jj;
    const int64_t blk = jj / cols_per_blk;
    const int within = (int)(jj - blk * cols_per_blk);
    const int g = within / PKN, h = within - g * PKN;
    T* base = P + blk * (int64_t)JG * slots * PKN;
    if (sl < dpad) {
        const int ch = sl / WIDE_CH, ll = sl - ch * WIDE_CH;
        base[((int64_t)(ch * JG + g) * WIDE_CH + ll) * PKN + h] = (sl < d) ? (Y[j * (int64_t)d + sl] - (Cn ? Cn[sl] : (T)0)) * gamma : (T)0;
    } else {
        const int c = sl - dpad;
        base[((int64_t)dpad * JG + (int64_t)g * NR + c) * PKN + h] = (!pad && c0 + c < nrhs) ? A[j + (int64_t)(c0 + c) * lda] : (T)0;
    }
}

template <typename T, int FAM, int NR, bool POW>
__global__ __launch_bounds__(DENSE_THREADS) void dense_wide_kernel(const T* __restrict__ X, int64_t n, int32_t d, int32_t dpad,
                                                                   const typename Pk<T>::V* __restrict__ P, int64_t m,
                                                                   T* __restrict__ out, int64_t npad, int64_t ldy, int32_t nrhs,
                                                                   int64_t jchunk, T alpha, T beta, int32_t final_store,
                                                                   const T* __restrict__ Cn, const typename ParamsOf<FAM, T>::type kp) {
    constexpr bool ISO = fam_is_iso<FAM>;
    using PK = Pk<T>;
    using V = typename PK::V;
    constexpr int JG = wide_jg<T>();
    constexpr int BC = JG * PK::N;                          // columns per block
    const int64_t BS = (int64_t)JG * (dpad + NR);           // stream elements (V) per block
    const int tid = threadIdx.x;
    int64_t row = (int64_t)blockIdx.x * DENSE_THREADS + tid;
    const bool live = row < n;
    if (!live) row = n - 1;
    const T* __restrict__ xr = X + row * (int64_t)d;
    const int64_t j0 = (int64_t)blockIdx.y * jchunk;        // a multiple of 64 >= BC
    const int64_t j1 = (j0 + jchunk < m) ? (j0 + jchunk) : m;
    const int64_t b0 = j0 / BC, b1 = (j1 + BC - 1) / BC;
    const int nch = dpad / WIDE_CH;

    T tot[NR];
#pragma unroll
    for (int c = 0; c < NR; ++c) tot[c] = (T)0;

    for (int64_t b = b0; b < b1; ++b) {
        const V* __restrict__ pb = P + b * BS;              // uniform
        V s[JG];
#pragma unroll
        for (int g = 0; g < JG; ++g) s[g] = PK::splat((T)0);
        for (int ch = 0; ch < nch; ++ch) {
            T x[WIDE_CH];
            const int l0 = ch * WIDE_CH;
            if (l0 + WIDE_CH <= d) {
#pragma unroll
                for (int l = 0; l < WIDE_CH; ++l) x[l] = (ISO ? xr[l0 + l] - Cn[l0 + l] : xr[l0 + l]) * kp.gamma;
            } else {
#pragma unroll
                for (int l = 0; l < WIDE_CH; ++l) x[l] = (l0 + l < d) ? (ISO ? xr[l0 + l] - Cn[l0 + l] : xr[l0 + l]) * kp.gamma : (T)0;
            }
            const V* __restrict__ pc = pb + (int64_t)ch * (JG * WIDE_CH);
#pragma unroll
            for (int g = 0; g < JG; ++g) {
                V sg = s[g];
#pragma unroll
                for (int q0 = 0; q0 < WIDE_CH; q0 += WIDE_SB) {
#pragma unroll
                    for (int l = q0; l < q0 + WIDE_SB; ++l) {
                        const V xl = PK::splat(x[l]);
                        const V yl = pc[g * WIDE_CH + l];   // compile-time offset from the uniform base
                        if constexpr (ISO) {
                            const V dl = xl - yl;
                            sg = PK::fma(dl, dl, sg);
                        } else {
                            sg = PK::fma(xl, yl, sg);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                s[g] = sg;
            }
        }
        // profile + accumulate (one block = 2*JG columns: the two-level accumulation comes for free)
        V acc[NR];
#pragma unroll
        for (int c = 0; c < NR; ++c) acc[c] = PK::splat((T)0);
        const V* __restrict__ pa = pb + (int64_t)dpad * JG;
#pragma unroll
        for (int g = 0; g < JG; ++g) {
            const V kv = PK::map(s[g], [&](T sv) { return phi_value<FAM, T, dense_folded<FAM, T>, POW>(sv, kp); });
#pragma unroll
            for (int c = 0; c < NR; ++c) acc[c] = PK::fma(pa[g * NR + c], kv, acc[c]);
        }
#pragma unroll
        for (int c = 0; c < NR; ++c) tot[c] += PK::hsum(acc[c]);
    }

    if (!live) return;
    if (final_store) {
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            if (c < nrhs) {
                T* yp = out + row + (int64_t)c * ldy;
                T v = alpha * tot[c];
                if (beta != (T)0) v = cg_fma(beta, *yp, v);
                *yp = v;
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < NR; ++c) out[((int64_t)blockIdx.y * NR + c) * npad + row] = tot[c];
    }
}

template <typename T, int FAM, int NR>
static int launch_dense_wide_NR(const DenseArgs& a) {
    const typename ParamsOf<FAM, T>::type kp = make_params<FAM, T>(*a.hk);
    dim3 grid((unsigned)((a.n + DENSE_THREADS - 1) / DENSE_THREADS), (unsigned)a.jsplit);
    const int final_store = (a.jsplit == 1) ? 1 : 0;
    const bool pow = a.hk->k.power != 1;
#define CG_WIDE_LAUNCH(POWV)                                                                                                        \
    hipLaunchKernelGGL((dense_wide_kernel<T, FAM, NR, POWV>), grid, dim3(DENSE_THREADS), 0, a.stream, (const T*)a.X, a.n, a.d, a.Dpad, \
                       (const typename Pk<T>::V*)a.P, a.m, (T*)a.out, a.npad, a.ldy, a.nrhs, a.jchunk, (T)a.alpha, (T)a.beta,        \
                       final_store, (const T*)a.C, kp)
    if constexpr (fam_is_expr<FAM>) CG_WIDE_LAUNCH(false);
    else { if (pow) CG_WIDE_LAUNCH(true); else CG_WIDE_LAUNCH(false); }
#undef CG_WIDE_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dense_wide launch failed: %s", hipGetErrorString(e)); return COVGRAM_EHIP; }
    return COVGRAM_OK;
}

template <int FAM>
int launch_dense_wide_family(const DenseArgs& a, int dtype) {
    if (dtype == COVGRAM_F32) {
        if (a.NRpad == 1) return launch_dense_wide_NR<float, FAM, 1>(a);
        if (a.NRpad == 4) return launch_dense_wide_NR<float, FAM, 4>(a);
    } else {
        if (a.NRpad == 1) return launch_dense_wide_NR<double, FAM, 1>(a);
        if (a.NRpad == 4) return launch_dense_wide_NR<double, FAM, 4>(a);
    }
    set_error("dense_wide: nrhs pad %d not compiled", a.NRpad);
    return COVGRAM_EUNSUPPORTED;
}

}  // namespace covgram
