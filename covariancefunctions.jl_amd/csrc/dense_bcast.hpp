// dense_bcast.hpp — fp64 dense Gramian MVM for WIDE points (d >= 8): expanded distance, column records in vector registers (round 4).
//
// dense_mvm_kernel evaluates |x_i - y_j|^2 by direct differences (src/util.jl:40-47): a v_add_f64 and a v_fma_f64 per dimension and pair,
// the column record streamed through SGPRs.  From d = 16 on that stream — 8 d bytes per column through out-of-order scalar loads, one
// 64-byte chunk in flight per wave — stalls the loop as it did the gradient kernel's (measured, profiles/r04_fp64_dense_d_sweep.txt, fp64
// issue slots per pair against 2 d + 23 instructions: d = 16 78 for 55, d = 32 133 for 87, d = 48 204 for 119, d = 64 874).
// This kernel is the dense twin of grad_bcast_kernel (grad_bcast.hpp): with the norms |x'_i|^2, |y'_j|^2 known,
//     s_ij = |x'_i|^2 + |y'_j|^2 - 2 x'_i . y'_j,      b_i += a_j phi(s_ij)
// costs ONE v_fmac_f64 per dimension and pair, its column operand taken by DPP broadcast (row_newbcast) from a record held in
// d / 16 VGPR pairs that counted vector loads keep one block of four columns ahead; (|y'_j|^2, a_j) ride beside it as scalars.
// s is a difference of O(|x'|^2) terms: absolute error ~1e-16 R^2 (R = radius of the pre-scaled cloud about the common centre), so the
// form runs under the gradient kernel's radius gate (gamma^2 R^2 <= GRAD_EXPAND_GATE = 1000) and only for profiles that are smooth in s at 0
// (EQ, RQ, Cauchy, IMQ, MaternP(p >= 1)); everything else keeps the direct differences.  fp64, one right-hand side, no Power wrapper.
#pragma once
#include "dense_mvm.hpp"
#include "grad_bcast.hpp"

namespace covgram {

template <int D> constexpr int dense_bcast_pairs = (D + 15) / 16;
constexpr int DENSE_BCAST_CB = 4;                 // columns per block (one s_load_dwordx16 of their scalars, 4 D / 16 record loads)
constexpr bool dense_bcast_ok(int D) { return D >= 8 && D <= 64; }
// registers: x (2 D) + two blocks of records + the profile's temporaries for FOUR columns evaluated side by side (MaternP / RQ keep their
// polynomial and table state per column: with 48 registers of temporaries those instances spilled 400-600 bytes of scratch inside the loop)
template <int FAM, int D> constexpr int dense_bcast_waves() {
    const int temps = (FAM == COVGRAM_MATERNP || FAM == COVGRAM_RQ) ? 120 : 48;
    const int regs = 2 * D + 4 * DENSE_BCAST_CB * dense_bcast_pairs<D> + temps;
    const int w = 512 / regs;
    return w < 1 ? 1 : (w > 4 ? 4 : w);
}

// dot += sum_l bcast(rec, l) * x[l], four partial sums (the DP ALU's accumulate latency under one wave's issue)
template <int L, int D, int NP>
__device__ __forceinline__ void bcast_dot(const double (&rec)[NP], const double (&x)[D], double (&p)[4]) {
    if constexpr (L < D) {
        fmac_rec<L, NP>(p[L & 3], rec, x[L]);
        bcast_dot<L + 1, D, NP>(rec, x, p);
    }
}

// P: [mpad][D] pre-scaled, centred column points (zero rows beyond m); Ex: [mpad][2] = (|y'_j|^2, a_j), weight 0 beyond m
template <int FAM, int D, int WAVES>
__global__ __launch_bounds__(64 * WAVES, (dense_bcast_waves<FAM, D>())) void dense_bcast_kernel(
    const double* __restrict__ X, int64_t n, int32_t d, const double* __restrict__ P, const double* __restrict__ Ex, int64_t m,
    double* __restrict__ out, int64_t npad, int64_t jchunk, double alpha, double beta, int32_t final_store, const double* __restrict__ Cn,
    const typename ParamsOf<FAM, double>::type kp0) {
    using T = double;
    static_assert(fam_is_iso<FAM> && !fam_is_expr<FAM>, "expanded distance: isotropic single profiles");
    if constexpr (dense_lds_tab<FAM>) exp_tab_lds_fill();
    if constexpr (FAM == COVGRAM_RQ || FAM == COVGRAM_GAMMAEXP) log_tab_lds_fill();
    constexpr int NP = dense_bcast_pairs<D>, CB = DENSE_BCAST_CB;
    const int tid = threadIdx.x;
    int64_t row = (int64_t)blockIdx.x * blockDim.x + tid;
    const bool live = row < n;
    if (!live) row = n - 1;
    const int64_t j0 = (int64_t)blockIdx.y * jchunk;                // multiples of CB
    const int64_t j1 = (j0 + jchunk < m) ? (j0 + jchunk) : m;
    const int nblk = (int)((j1 - j0 + CB - 1) / CB);                // the stream is padded to whole blocks (+ one prefetch-only block)

    T x[D];
    {
        const T* xr = X + row * (int64_t)d;
        if (d == D) {
#pragma unroll
            for (int l = 0; l < D; ++l) x[l] = (xr[l] - Cn[l]) * kp0.gamma;
        } else {
#pragma unroll
            for (int l = 0; l < D; ++l) x[l] = (l < d) ? (xr[l] - Cn[l]) * kp0.gamma : (T)0;
        }
    }
    T nx = (T)0;
#pragma unroll
    for (int l = 0; l < D; ++l) nx = cg_fma(x[l], x[l], nx);
    const T hnx = (T)-0.5 * nx;

    const double* __restrict__ pl = P + j0 * D + (tid & 15);
    const double* __restrict__ exq = Ex + 2 * j0;
    auto load_blk = [&](double (&rec)[CB][NP], int bi) {
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int k = 0; k < NP; ++k) rec[c][k] = pl[((int64_t)bi * CB + c) * D + 16 * k];
    };
    T tot = (T)0;
    auto sweep = [&](const typename ParamsOf<FAM, T>::type& kp) {
        auto block = [&](const double (&rec)[CB][NP], const T (&sc)[2 * CB]) {
            T kv[CB];
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                // the partial sums START from -(|x'|^2 + |y'|^2) / 2: |x' - y'|^2 = -2 (p0 + p1 + p2 + p3)
                T p[4] = {cg_fma((T)-0.5, sc[2 * c], hnx), (T)0, (T)0, (T)0};
                bcast_dot<0, D, NP>(rec[c], x, p);
                T s = (T)-2 * ((p[0] + p[1]) + (p[2] + p[3]));
                s = (s < (T)0) ? (T)0 : s;                           // rounding may take s a few ulp below zero; NaN stays NaN
                if constexpr (dense_s0<FAM, T> != (T)0) s += dense_s0<FAM, T>;   // MaternP: its square root takes no zero test (dense_mvm.hpp)
                kv[c] = dense_phi<FAM, T, false>(s, kp);
            }
#pragma unroll
            for (int c = 0; c < CB; ++c) tot = cg_fma(sc[2 * c + 1], kv[c], tot);
        };
        double ra[CB][NP], rb[CB][NP];
        T sa[2 * CB], sb[2 * CB];
        load_blk(ra, 0);
#pragma unroll
        for (int q = 0; q < 2 * CB; ++q) sa[q] = exq[q];
        int bi = 0;
        for (; bi + 2 <= nblk; bi += 2) {
            load_blk(rb, bi + 1);
#pragma unroll
            for (int q = 0; q < 2 * CB; ++q) sb[q] = exq[2 * CB * (bi + 1) + q];
            __builtin_amdgcn_sched_barrier(0);
            block(ra, sa);
            __builtin_amdgcn_sched_barrier(0);
            load_blk(ra, bi + 2);
#pragma unroll
            for (int q = 0; q < 2 * CB; ++q) sa[q] = exq[2 * CB * (bi + 2) + q];
            __builtin_amdgcn_sched_barrier(0);
            block(rb, sb);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (bi < nblk) block(ra, sa);
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
    if (!live) return;
    if (final_store) {
        T v = alpha * tot;
        if (beta != (T)0) v = cg_fma(beta, out[row], v);
        out[row] = v;
    } else {
        out[(int64_t)blockIdx.y * npad + row] = tot;
    }
}

// gramian(k, x) on one point set: the upper triangle once (dense_sym_kernel's scheme, dense_mvm.hpp) with the broadcast distance.  A workgroup
// is one wave = one 64-row block; it walks the column blocks of four from its own diagonal block to the right: inside the diagonal block
// row sums only, right of it every evaluated k_ij also feeds the column sum b_j += a_i k_ij — the block's four columns are exactly what
// wave_sum4_f64 reduces at once (lane-swap adds + a row scan), lanes 15 / 31 / 47 / 63 store the four totals to colslab[rb][j].
// dense_sym_reduce_kernel (unchanged) adds, per output row, the split-J partials of its row block and the column sums of the blocks above.
template <int FAM, int D>
__global__ __launch_bounds__(64, (dense_bcast_waves<FAM, D>())) void dense_bcast_sym_kernel(
    const double* __restrict__ X, int64_t n, int32_t d, const double* __restrict__ P, const double* __restrict__ Ex, double* __restrict__ out,
    double* __restrict__ colslab, int64_t npad, int64_t jchunk, const double* __restrict__ Cn, const typename ParamsOf<FAM, double>::type kp0,
    int32_t rb_first, int32_t rb_stride) {
    using T = double;
    if constexpr (dense_lds_tab<FAM>) exp_tab_lds_fill();
    if constexpr (FAM == COVGRAM_RQ || FAM == COVGRAM_GAMMAEXP) log_tab_lds_fill();
    constexpr int NP = dense_bcast_pairs<D>, CB = DENSE_BCAST_CB;
    static_assert(CB == 4, "the column-sum reduction takes four columns at a time");
    const int lane = threadIdx.x;
    const int64_t row_lo = ((int64_t)rb_first + (int64_t)blockIdx.x * rb_stride) * 64;
    const int64_t n4 = (n + CB - 1) / CB * CB;
    const int64_t j0 = (int64_t)blockIdx.y * jchunk;                // multiples of 64
    const int64_t j1 = (j0 + jchunk < n4) ? (j0 + jchunk) : n4;
    const int64_t row = row_lo + lane;
    if (j1 <= row_lo) {                                               // the whole chunk lies left of the diagonal block
        out[(int64_t)blockIdx.y * npad + row] = 0.0;
        return;
    }
    const int64_t rowc = (row < n) ? row : n - 1;                     // clamp: computed, weighted 0 in the column sums, never stored as a row
    T x[D];
    {
        const T* xr = X + rowc * (int64_t)d;
#pragma unroll
        for (int l = 0; l < D; ++l) x[l] = (l < d) ? (xr[l] - Cn[l]) * kp0.gamma : (T)0;
    }
    T nx = (T)0;
#pragma unroll
    for (int l = 0; l < D; ++l) nx = cg_fma(x[l], x[l], nx);
    const T hnx = (T)-0.5 * nx;
    const T ai = (row < n) ? Ex[2 * rowc + 1] : (T)0;                 // the row's own weight
    const int cmap = ((lane >> 4) & 1) * 2 + (lane >> 5);             // wave_sum4_f64: lanes 15 / 31 / 47 / 63 hold columns 0 / 2 / 1 / 3
    const int64_t jstart = (j0 > row_lo) ? j0 : row_lo;               // both multiples of 64
    const int nblk = (int)((j1 - jstart) / CB);
    const double* __restrict__ pl = P + jstart * D + (lane & 15);
    const double* __restrict__ exq = Ex + 2 * jstart;
    T* __restrict__ cdst = colslab + (int64_t)blockIdx.x * npad + jstart;
    const int ndiag = (jstart == row_lo) ? 64 / CB : 0;               // the first 16 blocks are the diagonal block when the chunk starts on it
    auto load_blk = [&](double (&rec)[CB][NP], int bi) {
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int k = 0; k < NP; ++k) rec[c][k] = pl[((int64_t)bi * CB + c) * D + 16 * k];
    };
    T tot = (T)0;
    auto sweep = [&](const typename ParamsOf<FAM, T>::type& kp) {
        auto block = [&](const double (&rec)[CB][NP], const T (&sc)[2 * CB], int bi) {
            T kv[CB];
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                T p[4] = {cg_fma((T)-0.5, sc[2 * c], hnx), (T)0, (T)0, (T)0};
                bcast_dot<0, D, NP>(rec[c], x, p);
                T s = (T)-2 * ((p[0] + p[1]) + (p[2] + p[3]));
                s = (s < (T)0) ? (T)0 : s;
                if constexpr (dense_s0<FAM, T> != (T)0) s += dense_s0<FAM, T>;
                kv[c] = dense_phi<FAM, T, false>(s, kp);
            }
#pragma unroll
            for (int c = 0; c < CB; ++c) tot = cg_fma(sc[2 * c + 1], kv[c], tot);
            if (bi >= ndiag) {                                        // wave-uniform: right of the diagonal block
                const T tsum = wave_sum4_f64(ai * kv[0], ai * kv[1], ai * kv[2], ai * kv[3]);
                if ((lane & 15) == 15) cdst[bi * CB + cmap] = tsum;
            }
        };
        double ra[CB][NP], rb[CB][NP];
        T sa[2 * CB], sb[2 * CB];
        load_blk(ra, 0);
#pragma unroll
        for (int q = 0; q < 2 * CB; ++q) sa[q] = exq[q];
        int bi = 0;
        for (; bi + 2 <= nblk; bi += 2) {
            load_blk(rb, bi + 1);
#pragma unroll
            for (int q = 0; q < 2 * CB; ++q) sb[q] = exq[2 * CB * (bi + 1) + q];
            __builtin_amdgcn_sched_barrier(0);
            block(ra, sa, bi);
            __builtin_amdgcn_sched_barrier(0);
            load_blk(ra, bi + 2);
#pragma unroll
            for (int q = 0; q < 2 * CB; ++q) sa[q] = exq[2 * CB * (bi + 2) + q];
            __builtin_amdgcn_sched_barrier(0);
            block(rb, sb, bi + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (bi < nblk) block(ra, sa, bi);
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
    out[(int64_t)blockIdx.y * npad + row] = tot;
}

// P[j][l] = gamma (Y[j][l] - c[l]) (zero padded in l and for j >= m), Ex[2 j] = |P[j]|^2, Ex[2 j + 1] = a_j (0 for j >= m); mpad rows
template <typename T /* double */>
__global__ __launch_bounds__(256) void dense_bcast_pack_kernel(const double* __restrict__ Y, int64_t m, int32_t d, const double* __restrict__ A,
                                                               double* __restrict__ P, double* __restrict__ Ex, int32_t D, double gamma,
                                                               const double* __restrict__ Cn, int64_t mpad) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= mpad) return;
    double ny = 0.0;
    double* p = P + j * (int64_t)D;
    for (int l = 0; l < D; ++l) {
        const double v = (j < m && l < d) ? (Y[j * (int64_t)d + l] - Cn[l]) * gamma : 0.0;
        p[l] = v;
        ny = __builtin_fma(v, v, ny);
    }
    Ex[2 * j] = ny;
    Ex[2 * j + 1] = (j < m) ? A[j] : 0.0;
}

// The same with DL = D (a power of two: 16, 32, 64) lanes per column, one coordinate each: consecutive lanes read and write consecutive memory (round 5;
// a thread per column put every lane of a load on its own cache line: 14.4 us for 4 MB at d = 32, n = 16384)
template <int DL>
__global__ __launch_bounds__(256) void dense_bcast_pack_lanes_kernel(const double* __restrict__ Y, int64_t m, int32_t d, const double* __restrict__ A,
                                                                     double* __restrict__ P, double* __restrict__ Ex, double gamma,
                                                                     const double* __restrict__ Cn, int64_t mpad) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j = g / DL;
    const int l = (int)(g % DL);
    if (j >= mpad) return;                                       // (whole lane groups: DL divides the block)
    const double v = (j < m && l < d) ? (Y[j * (int64_t)d + l] - Cn[l]) * gamma : 0.0;
    P[j * (int64_t)DL + l] = v;
    double ny = v * v;
#pragma unroll
    for (int o = DL / 2; o > 0; o >>= 1) ny += __shfl_xor(ny, o);
    if (l == 0) { Ex[2 * j] = ny; Ex[2 * j + 1] = (j < m) ? A[j] : 0.0; }
}

template <int FAM, int D>
static int launch_dense_bcast_one(const DenseArgs& a) {
    if constexpr (dense_bcast_ok(D) && fam_is_iso<FAM> && !fam_is_expr<FAM> && FAM != COVGRAM_MATERN && FAM != COVGRAM_EXP && FAM != COVGRAM_GAMMAEXP) {
        const typename ParamsOf<FAM, double>::type kp = make_params<FAM, double>(*a.hk);
        if (a.sym) {   // gramian(k, x): the upper triangle once, 64-row blocks first, first + stride, ... (all of them: 0, 1)
            const int64_t blocks = (a.n + 63) / 64;
            const int64_t mine = a.sym_first < blocks ? (blocks - a.sym_first + a.sym_stride - 1) / a.sym_stride : 0;
            if (mine == 0) return COVGRAM_OK;
            hipLaunchKernelGGL((dense_bcast_sym_kernel<FAM, D>), dim3((unsigned)mine, (unsigned)a.jsplit), dim3(64), 0, a.stream, (const double*)a.X, a.n, a.d,
                               (const double*)a.P, (const double*)a.Ex, (double*)a.out, (double*)a.colslab, a.npad, a.jchunk, (const double*)a.C, kp, a.sym_first,
                               a.sym_stride);
        } else {
        const int final_store = (a.jsplit == 1) ? 1 : 0;
        dim3 grid((unsigned)((a.n + 255) / 256), (unsigned)a.jsplit);
        hipLaunchKernelGGL((dense_bcast_kernel<FAM, D, 4>), grid, dim3(256), 0, a.stream, (const double*)a.X, a.n, a.d, (const double*)a.P,
                           (const double*)a.Ex, a.m, (double*)a.out, a.npad, a.jchunk, a.alpha, a.beta, final_store, (const double*)a.C, kp);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error("dense_bcast launch failed: %s", hipGetErrorString(e)); return COVGRAM_EHIP; }
        return COVGRAM_OK;
    } else {
        set_error("dense_bcast: no instance for this family / dimension");
        return COVGRAM_EUNSUPPORTED;
    }
}

}  // namespace covgram
