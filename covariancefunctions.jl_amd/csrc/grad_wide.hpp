// grad_wide.hpp — GradientKernel Gramian MVM for point dimensions beyond the register-resident set of grad_mvm.hpp
// (the reference's flagship gradient example has d = 1024, README.md:231-245).  Same block algebra
// (src/gradient.jl:86-92, 109-115), same lane-per-row / scalar-stream mapping, but neither x_i nor the d-vector
// accumulator b_i fits in VGPRs, so the MVM runs per column PANEL as two kernels:
//   grad_wide_coef_kernel   (row block × column block)  walks the dimension in chunks of 32 and accumulates, for a block of
//                           2*GJG columns at once, s = |r|^2 (or x.y) and t = r.a (or x.a); then stores the two scalars of
//                           every block of the Gramian, c1 = phi'(s) and c2 = 2 phi''(s) t (dot product: phi''(s) t), as
//                           [column group][row] slabs (coalesced over the lanes);
//   grad_wide_apply_kernel  (row block × dimension chunk)  keeps 32 coordinates of b_i in registers over ALL columns of the
//                           panel:  b_l += c1 a_jl + c2 (x_il - y_jl)   (dot product: c1 a_jl + c2 y_jl), and writes the
//                           chunk once with alpha/beta (first panel) or accumulating (later panels).
// 6 flops per dimension and block instead of 4-5, in exchange for unbounded d and (row × chunk) parallelism at small n.
// The coefficient slab is 2 * npad * panel columns scalars; panels keep it under 256 MB.
#pragma once
#include "dense_wide.hpp"

namespace covgram {

constexpr int GJG = 16;   // column groups per block (fp32: 32 columns, fp64: 16)

// Blocked panel stream: block b, chunk ch:  [y: GJG x 32][a: GJG x 32]  packed values, every operand of a (block, chunk) at a
// compile-time offset from one scalar base.  Columns past m repeat the last point with a = 0 (c2 = 0, c1 * a = 0).
template <typename T>
__global__ __launch_bounds__(256) void grad_wide_pack_kernel(const T* __restrict__ Y, int64_t m, int32_t d, int32_t dpad,
                                                             const T* __restrict__ A, int64_t col0, int64_t pcols,
                                                             T* __restrict__ P, int32_t PKN, T gamma, int32_t vg, T* __restrict__ A0P,
                                                             const T* __restrict__ Cn) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (panel column, coordinate)
    if (e >= pcols * (int64_t)dpad) return;
    const int64_t jp = e / dpad;
    const int sl = (int)(e - jp * dpad);
    const int64_t jj = col0 + jp;
    const bool pad = jj >= m;
    const int64_t j = pad ? (m - 1) : jj;
    const int64_t bc = (int64_t)GJG * PKN;
    const int64_t blk = jp / bc;
    const int within = (int)(jp - blk * bc);
    const int g = within / PKN, h = within - g * PKN;
    const int ch = sl / WIDE_CH, ll = sl - ch * WIDE_CH;
    const int nch = dpad / WIDE_CH;
    T* base = P + blk * (int64_t)nch * 2 * GJG * WIDE_CH * PKN;
    const bool real = sl < d;
    base[(((int64_t)(ch * 2 + 0) * GJG + g) * WIDE_CH + ll) * PKN + h] = real ? (Y[j * (int64_t)d + sl] - (Cn ? Cn[sl] : (T)0)) * gamma : (T)0;
    base[(((int64_t)(ch * 2 + 1) * GJG + g) * WIDE_CH + ll) * PKN + h] = (real && !pad) ? A[j * (int64_t)(d + vg) + vg + sl] : (T)0;
    // value-gradient blocks: the value weight of each column, [block][group][packed column]
    if (vg && sl == 0) A0P[(blk * GJG + g) * PKN + h] = pad ? (T)0 : A[j * (int64_t)(d + 1)];
}

// VG: ValueGradientKernel blocks (grad_mvm.hpp): c2 gets + vg_c k1 a0, and the value row's partial sum of this column
// block, sum_j (k0 a0 + vg_b k1 t), goes to C0[block][row] for the apply kernel to add up.
template <typename T, int FAM, bool POW, bool VG>
__global__ __launch_bounds__(64) void grad_wide_coef_kernel(const T* __restrict__ X, int64_t n, int32_t d, int32_t dpad,
                                                            const typename Pk<T>::V* __restrict__ P,
                                                            typename Pk<T>::V* __restrict__ C1, typename Pk<T>::V* __restrict__ C2,
                                                            int64_t npad, const typename Pk<T>::V* __restrict__ A0P, T* __restrict__ C0,
                                                            T vg_c, T vg_b, const T* __restrict__ Cn,
                                                            const typename ParamsOf<FAM, T>::type kp) {
    constexpr bool ISO = fam_is_iso<FAM>;
    using PK = Pk<T>;
    using V = typename PK::V;
    int64_t row = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t srow = row;
    if (row >= n) row = n - 1;
    const T* __restrict__ xr = X + row * (int64_t)d;
    const int nch = dpad / WIDE_CH;
    const int64_t b = blockIdx.y;
    const V* __restrict__ pb = P + b * (int64_t)nch * 2 * GJG * WIDE_CH;
    V s[GJG], t[GJG];
#pragma unroll
    for (int g = 0; g < GJG; ++g) { s[g] = PK::splat((T)0); t[g] = PK::splat((T)0); }
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
        const V* __restrict__ py = pb + (int64_t)ch * (2 * GJG * WIDE_CH);
        const V* __restrict__ pa = py + GJG * WIDE_CH;
#pragma unroll
        for (int g = 0; g < GJG; ++g) {
            V sg = s[g], tg = t[g];
#pragma unroll
            for (int q0 = 0; q0 < WIDE_CH; q0 += WIDE_SB) {
#pragma unroll
                for (int l = q0; l < q0 + WIDE_SB; ++l) {
                    const V xl = PK::splat(x[l]);
                    if constexpr (ISO) {
                        const V rl = xl - py[g * WIDE_CH + l];
                        sg = PK::fma(rl, rl, sg);
                        tg = PK::fma(rl, pa[g * WIDE_CH + l], tg);
                    } else {
                        sg = PK::fma(xl, py[g * WIDE_CH + l], sg);
                        tg = PK::fma(xl, pa[g * WIDE_CH + l], tg);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            s[g] = sg; t[g] = tg;
        }
    }
    if (srow >= n) return;
    T b0 = (T)0;
    const T f = ISO ? (T)2 : (T)1;
    if constexpr (fam_is_expr<FAM>) {
        // composites: the jets of 4 column groups at a time, factor-outer (profiles.hpp: expr_jet_block)
        constexpr int BG = 4;
#pragma unroll
        for (int g0 = 0; g0 < GJG; g0 += BG) {
            V sb[BG], k0[BG], k1[BG], k2[BG];
#pragma unroll
            for (int g = 0; g < BG; ++g) sb[g] = s[g0 + g];
            expr_jet_block<T, ISO, BG>(sb, kp, k0, k1, k2);
#pragma unroll
            for (int g = 0; g < BG; ++g) {
                V c2 = PK::splat(f) * k2[g] * t[g0 + g];
                if constexpr (VG) {
                    const V a0 = A0P[b * GJG + g0 + g];
                    c2 = PK::fma(PK::splat(vg_c) * k1[g], a0, c2);
                    b0 += PK::hsum(PK::fma(k0[g], a0, PK::splat(vg_b) * k1[g] * t[g0 + g]));
                }
                C1[(b * GJG + g0 + g) * npad + srow] = k1[g];
                C2[(b * GJG + g0 + g) * npad + srow] = c2;
            }
        }
    } else {
#pragma unroll
    for (int g = 0; g < GJG; ++g) {
        V c1, c2;
        if constexpr (PK::N == 2) {
            T k0x, k0y, k1x, k2x, k1y, k2y;
            phi_jet<FAM, T, POW>(s[g].x, kp, k0x, k1x, k2x);
            phi_jet<FAM, T, POW>(s[g].y, kp, k0y, k1y, k2y);
            c1 = V{k1x, k1y};
            c2 = V{f * k2x * t[g].x, f * k2y * t[g].y};
            if constexpr (VG) {
                const V a0 = A0P[b * GJG + g];
                c2 = V{cg_fma(vg_c * k1x, a0.x, c2.x), cg_fma(vg_c * k1y, a0.y, c2.y)};
                b0 += cg_fma(k0x, a0.x, vg_b * k1x * t[g].x) + cg_fma(k0y, a0.y, vg_b * k1y * t[g].y);
            }
        } else {
            T k0, k1, k2;
            phi_jet<FAM, T, POW>(s[g], kp, k0, k1, k2);
            c1 = k1;
            c2 = f * k2 * t[g];
            if constexpr (VG) {
                const V a0 = A0P[b * GJG + g];
                c2 = cg_fma(vg_c * k1, a0, c2);
                b0 += cg_fma(k0, a0, vg_b * k1 * t[g]);
            }
        }
        C1[(b * GJG + g) * npad + srow] = c1;
        C2[(b * GJG + g) * npad + srow] = c2;
    }
    }
    if constexpr (VG) C0[b * npad + srow] = b0;
}

template <typename T, bool ISO>
__global__ __launch_bounds__(64) void grad_wide_apply_kernel(const T* __restrict__ X, int64_t n, int32_t d, int32_t dpad,
                                                             const typename Pk<T>::V* __restrict__ P,
                                                             const typename Pk<T>::V* __restrict__ C1,
                                                             const typename Pk<T>::V* __restrict__ C2, int64_t npad,
                                                             int64_t nblocks, T* __restrict__ y, T alpha, T beta, int32_t accumulate,
                                                             T gamma, int32_t vg, const T* __restrict__ C0, T alpha0, int64_t zstride,
                                                             const T* __restrict__ Cn) {
    using PK = Pk<T>;
    using V = typename PK::V;
    const int64_t row = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (row >= n) return;                                   // no barriers, no cross-lane traffic below
    // gridDim.z > 1: the panel's column blocks are split over z and each slice accumulates RAW sums into its own output
    // slab (y + z * zstride); grad_wide_reduce_kernel adds the slices in fixed order and applies alpha / beta.  This is what
    // fills the chip when (row blocks x dimension chunks) alone are too few waves (n = 16384, d = 128: 1024).
    const int64_t b_begin = nblocks * blockIdx.z / gridDim.z, b_end = nblocks * (blockIdx.z + 1) / gridDim.z;
    y += (int64_t)blockIdx.z * zstride;
    const int ch = blockIdx.y;
    const int l0 = ch * WIDE_CH;
    const int nch = dpad / WIDE_CH;
    const T* __restrict__ xr = X + row * (int64_t)d;
    T x[WIDE_CH];
    V bv[WIDE_CH];
#pragma unroll
    for (int l = 0; l < WIDE_CH; ++l) {
        x[l] = (l0 + l < d) ? (ISO ? xr[l0 + l] - Cn[l0 + l] : xr[l0 + l]) * gamma : (T)0;
        bv[l] = PK::splat((T)0);
    }
    for (int64_t b = b_begin; b < b_end; ++b) {
        const V* __restrict__ py = P + (b * nch + ch) * (int64_t)(2 * GJG * WIDE_CH);
        const V* __restrict__ pa = py + GJG * WIDE_CH;
#pragma unroll 1
        for (int g = 0; g < GJG; ++g) {
            const V c1 = C1[(b * GJG + g) * npad + row];
            const V c2 = C2[(b * GJG + g) * npad + row];
#pragma unroll
            for (int q0 = 0; q0 < WIDE_CH; q0 += WIDE_SB) {
#pragma unroll
                for (int l = q0; l < q0 + WIDE_SB; ++l) {
                    V v;
                    if constexpr (ISO) v = PK::splat(x[l]) - py[g * WIDE_CH + l];
                    else v = py[g * WIDE_CH + l];
                    bv[l] = PK::fma(c2, v, PK::fma(c1, pa[g * WIDE_CH + l], bv[l]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (vg && ch == 0) {                                    // value row: add up the column blocks' partial sums
        T b0 = (T)0;
        for (int64_t b = b_begin; b < b_end; ++b) b0 += C0[b * npad + row];
        T* y0 = y + row * (int64_t)(d + 1);
        T v = alpha0 * b0;
        if (accumulate) v += *y0;
        else if (beta != (T)0) v = cg_fma(beta, *y0, v);
        *y0 = v;
    }
    T* yp = y + row * (int64_t)(d + vg) + vg + l0;
#pragma unroll
    for (int l = 0; l < WIDE_CH; ++l) {
        if (l0 + l < d) {
            T v = alpha * PK::hsum(bv[l]);
            if (accumulate) v += yp[l];
            else if (beta != (T)0) v = cg_fma(beta, yp[l], v);
            yp[l] = v;
        }
    }
}

// y[e] = alpha(e) * sum_z slab[z][e] + beta * y[e];  alpha(e) = alpha0 for the value entry of a value-gradient block
// One thread per (row, entry of the block): grid = (rows / 256, bd); blockIdx.y == 0 is the value entry when vg.
// (A flat index with `e % bd == 0` selecting the scale was miscompiled by hipcc 7.2 at -O3 — every entry took alpha0; the
// kernel printed the right values as soon as a printf was added.  The 2-D form has no modulo to get wrong.)
template <typename T>
__global__ __launch_bounds__(256) void grad_wide_reduce_kernel(const T* __restrict__ slab, int32_t zs, int64_t total, int64_t n,
                                                               T* __restrict__ y, T alpha, T beta, int32_t vg, int32_t bd, T alpha0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (i >= n) return;
    const int64_t e = i * bd + k;
    T s = (T)0;
    for (int z = 0; z < zs; ++z) s += slab[(int64_t)z * total + e];
    const T sc = (vg != 0 && k == 0) ? alpha0 : alpha;
    T v = sc * s;
    if (beta != (T)0) v = cg_fma(beta, y[e], v);
    y[e] = v;
}

struct GradWideArgs {
    const void* X; int64_t n; int32_t d; int32_t dpad;
    const void* P; void* C1; void* C2; int64_t npad; int64_t nblocks;
    void* y; double alpha, beta; int32_t accumulate;
    int32_t vg = 0; const void* A0P = nullptr; void* C0 = nullptr; double alpha0 = 0, vg_c = 0, vg_b = 0;
    int32_t zs = 1; int64_t zstride = 0;   // column split of the apply kernel (y then points at the slice slab)
    const void* Cn = nullptr;              // common centre of isotropic kernels (dense_mvm.hpp)
    const HostKernel* hk;
    hipStream_t stream;
};

template <typename T, int FAM>
static int launch_grad_wide_T(const GradWideArgs& a) {
    using V = typename Pk<T>::V;
    constexpr bool ISO = fam_is_iso<FAM>;
    const typename ParamsOf<FAM, T>::type kp = make_params<FAM, T>(*a.hk);
    const unsigned rb = (unsigned)((a.n + 63) / 64);
    const bool pow = !fam_is_expr<FAM> && a.hk->k.power != 1;
    constexpr bool POWT = !fam_is_expr<FAM>;
#define CG_COEF_LAUNCH(POWV, VGV)                                                                                                       \
    hipLaunchKernelGGL((grad_wide_coef_kernel<T, FAM, POWV, VGV>), dim3(rb, (unsigned)a.nblocks), dim3(64), 0, a.stream, (const T*)a.X, a.n, \
                       a.d, a.dpad, (const V*)a.P, (V*)a.C1, (V*)a.C2, a.npad, (const V*)a.A0P, (T*)a.C0, (T)a.vg_c, (T)a.vg_b, (const T*)a.Cn, kp)
    if (a.vg) { if (pow) CG_COEF_LAUNCH(POWT, true); else CG_COEF_LAUNCH(false, true); }
    else { if (pow) CG_COEF_LAUNCH(POWT, false); else CG_COEF_LAUNCH(false, false); }
#undef CG_COEF_LAUNCH
    hipLaunchKernelGGL((grad_wide_apply_kernel<T, ISO>), dim3(rb, (unsigned)(a.dpad / WIDE_CH), (unsigned)a.zs), dim3(64), 0, a.stream,
                       (const T*)a.X, a.n, a.d, a.dpad, (const V*)a.P, (const V*)a.C1, (const V*)a.C2, a.npad, a.nblocks, (T*)a.y, (T)a.alpha,
                       (T)a.beta, a.accumulate, kp.gamma, a.vg, (const T*)a.C0, (T)a.alpha0, a.zstride, (const T*)a.Cn);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("grad_wide launch failed: %s", hipGetErrorString(e)); return COVGRAM_EHIP; }
    return COVGRAM_OK;
}

template <int FAM>
int launch_grad_wide_family(const GradWideArgs& a, int dtype) {
    if (dtype == COVGRAM_F32) return launch_grad_wide_T<float, FAM>(a);
    return launch_grad_wide_T<double, FAM>(a);
}

typedef int (*grad_wide_launch_fn)(const GradWideArgs&, int dtype);
grad_wide_launch_fn grad_wide_launcher(int family);

}  // namespace covgram
