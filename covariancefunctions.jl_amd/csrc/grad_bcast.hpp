// grad_bcast.hpp — the fp64 expanded-form gradient MVM with the column records in VECTOR registers (round 4, C4).
//
// grad_mvm_kernel streams the record (y'_j, a_j) of a column through SGPRs: free operands, but scalar loads return out of order, so a
// wave can have ONE 64-byte chunk in flight (lgkmcnt(0) is the only wait there is) and at the 3 waves per SIMD that 2 d-vectors of
// fp64 state allow, the kernel sat in s_waitcnt half of its life (VALU busy 0.70: profiles/r02_c4_grad_expanded_pmc.txt,
// r02_c4_smem_pmc.txt).  The two ways out that rounds 2-3 left open are both closed by measurement this round:
//   * the fp64 matrix pipe does not run beside the fp64 VALU (profiles/r04_mfma64_interleave_probe.txt: interleaved streams ADD), and
//   * LDS-staged records would have to come back through ds_read — 32 KB of return data per wave and column, 1.8x the CU's LDS rate.
// What gfx950 does have is a DPP broadcast on the DP ALU's accumulate form:  v_fmac_f64_dpp acc, src0, src1 row_newbcast:k  takes src0
// from lane k of each row of 16 lanes (tools/dpp64_probe.hip: semantics and rate).  So a record lives in 2 D / 16 VGPR pairs — lane l
// of every row holds element 16 k + (l & 15) of the record, loaded by ONE coalesced global_load_dwordx2 per pair — and every fma of the
// block takes its column operand by broadcast:
//     sweep 1:  s += y'_j[l] (bcast) * x'_i[l],   t += a_j[l] (bcast) * x'_i[l]
//     sweep 2:  b_i[l] += a_j[l] (bcast) * k1,    b_i[l] += y'_j[l] (bcast) * (-c2)
// all four of the accumulate form.  Vector loads are counted (vmcnt), so the NEXT column's record is in flight under the whole current
// column — no scalar chunks, no scheduling barriers, no SGPR pressure — at the price of 2 x 2 D / 8 staging registers.
// Same arithmetic as grad_mvm_kernel<..., EXPD> (sweep 1 runs two partial sums per reduction instead of one), same slab / reduce
// kernel, same pack kernels (records of 2 D scalars, the (|y'|^2, y'.a) pairs beside them); fp64, isotropic single profiles, one
// right-hand side, D in {8, ..., 48}.
#pragma once
#include "grad_mvm.hpp"

namespace covgram {

// acc += bcast_K(rec) * x, K = the lane of each row of 16 whose value every lane of the row reads
template <int K>
__device__ __forceinline__ void fmac_bc(double& acc, const double& rec, const double& x) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(rec), "v"(x), "n"(K));
}
// element E of a record held as NP register pairs: pair E / 16, lane E % 16
template <int E, int NP>
__device__ __forceinline__ void fmac_rec(double& acc, const double (&rec)[NP], const double& x) {
    fmac_bc<E % 16>(acc, rec[E / 16], x);
}

// the two sweeps over the dimensions, unrolled at compile time (the broadcast lane is an immediate of each instruction)
template <int L, int D, int NP>
__device__ __forceinline__ void bcast_sweep1(const double (&rec)[NP], const double (&x)[D], double& s0, double& s1, double& t0, double& t1) {
    if constexpr (L < D) {
        if constexpr (L % 2 == 0) { fmac_rec<L, NP>(s0, rec, x[L]); fmac_rec<D + L, NP>(t0, rec, x[L]); }
        else { fmac_rec<L, NP>(s1, rec, x[L]); fmac_rec<D + L, NP>(t1, rec, x[L]); }
        bcast_sweep1<L + 1, D, NP>(rec, x, s0, s1, t0, t1);
    }
}
template <int L, int D, int NP>
__device__ __forceinline__ void bcast_sweep2(const double (&rec)[NP], double (&b)[D], const double& k1, const double& mc2) {
    if constexpr (L < D) {
        fmac_rec<D + L, NP>(b[L], rec, k1);
        fmac_rec<L, NP>(b[L], rec, mc2);
        bcast_sweep2<L + 1, D, NP>(rec, b, k1, mc2);
    }
}

template <int D> constexpr int grad_bcast_pairs = (2 * D + 15) / 16;
// 2 d-vectors of state + two records of staging + ~44 registers of temporaries: waves per SIMD the allocator is asked for
template <int D> constexpr int grad_bcast_waves() {
    const int regs = 4 * D + 4 * grad_bcast_pairs<D> + 44;
    const int w = 512 / regs;
    return w < 1 ? 1 : (w > 4 ? 4 : w);
}
constexpr bool grad_bcast_ok(int D) { return D >= 8 && D <= 48; }

// VG: the ValueGradientKernel Gramian (blocks of d + 1, value component first; grad_mvm.hpp): one more scalar per column (the value
// weight a0, streamed beside the column scalars), one more accumulator (b0) and one more fma on c2 — the sweeps are the same.
template <int FAM, int D, bool POW, int WAVES, bool VG = false>
__global__ __launch_bounds__(64 * WAVES, (grad_bcast_waves<D>())) void grad_bcast_kernel(
    const double* __restrict__ X, int64_t n, int32_t d, const double* __restrict__ P, int64_t m, double* __restrict__ out, int64_t npad,
    int64_t jchunk, double alpha, double beta, int32_t final_store, const double* __restrict__ Cn,
    const typename ParamsOf<FAM, double>::type kp, const double* __restrict__ Ex, const double* __restrict__ A0, double alpha0, double vg_c,
    double vg_b) {
    using T = double;
    static_assert(fam_is_iso<FAM> && !fam_is_expr<FAM>, "expanded form: isotropic single profiles");
    // EQ: exp(-s / 2) on the LDS table (15 instructions + one ds_read against the polynomial's 22: now that the column operands no
    // longer stall the loop, the jet's instructions are a fifth of it)
    constexpr bool EQTAB = (FAM == COVGRAM_EQ) && !POW;
    if constexpr (grad_lds_tab<FAM, T> || EQTAB) { exp_tab_lds_fill(); if constexpr (FAM == COVGRAM_RQ || FAM == COVGRAM_GAMMAEXP) log_tab_lds_fill(); }
    constexpr int RS = 2 * D;                                  // scalars per column record (grad_pack_kernel, nr = 1)
    constexpr int NP = grad_bcast_pairs<D>;
    const int tid = threadIdx.x;
    int64_t row = (int64_t)blockIdx.x * blockDim.x + tid;
    const bool live = row < n;
    if (!live) row = n - 1;
    const int64_t j0 = (int64_t)blockIdx.y * jchunk;
    const int64_t j1 = (j0 + jchunk < m) ? (j0 + jchunk) : m;
    const int cnt = (int)(j1 - j0);

    T x[D], b[D];
    {
        const T* xr = X + row * (int64_t)d;
        if (d == D) {   // common case: no padding, straight vector loads
#pragma unroll
            for (int l = 0; l < D; ++l) x[l] = (xr[l] - Cn[l]) * kp.gamma;
        } else {
#pragma unroll
            for (int l = 0; l < D; ++l) x[l] = (l < d) ? (xr[l] - Cn[l]) * kp.gamma : (T)0;
        }
#pragma unroll
        for (int l = 0; l < D; ++l) b[l] = (T)0;
    }
    T nx = (T)0, csum = (T)0;
    [[maybe_unused]] T b0 = (T)0;
#pragma unroll
    for (int l = 0; l < D; ++l) nx = cg_fma(x[l], x[l], nx);
    const T hnx = (T)-0.5 * nx;

    // lane l of every row of 16 holds element 16 k + (l & 15) of the record: one 128-byte segment per pair, four rows reading the same
    const double* __restrict__ pl = P + j0 * RS + (tid & 15);
    const double* __restrict__ exq = Ex + 2 * j0;              // (|y'_j|^2, y'_j . a_j) per column, entry m is zero (prefetch only)
    const double* __restrict__ a0q = VG ? A0 + j0 : nullptr;   // value weights, entry m is zero (prefetch only)
    auto load_rec = [&](double (&rec)[NP], int jj) {
#pragma unroll
        for (int k = 0; k < NP; ++k) rec[k] = pl[(int64_t)jj * RS + 16 * k];   // past the chunk: the next record / the stream's zero pad
    };

    auto columns = [&](auto pfix_) {
        constexpr int PFIX = decltype(pfix_)::value;
        auto column = [&](const double (&rec)[NP], T eny, T eya, [[maybe_unused]] T a0) {
            // ---- sweep 1: two partial sums per reduction (the DP ALU's accumulate latency under one wave's issue); the sums START from
            // the column scalars: s0 from -(|x'|^2 + |y'|^2) / 2, so that |x' - y'|^2 = -2 (s0 + s1), and t0 from -y'.a
            T s0 = cg_fma((T)-0.5, eny, hnx), s1 = (T)0, t0 = -eya, t1 = (T)0;
            bcast_sweep1<0, D, NP>(rec, x, s0, s1, t0, t1);
            T s = (T)-2 * (s0 + s1);                            // x'.y' -> |x' - y'|^2
            s = (s < (T)0) ? (T)0 : s;                          // rounding may take s a few ulp below zero; NaN stays NaN
            const T t = t0 + t1;                                // x'.a -> r'.a
            T k1, k2, v_;
            if constexpr (EQTAB) { v_ = exp_neg_half_lds(s); k1 = (T)-0.5 * v_; k2 = (T)0.25 * v_; }
            else grad_jet<FAM, T, POW, PFIX>(s, kp, v_, k1, k2);
            T c2;
            if constexpr (VG) {
                c2 = cg_fma(vg_c * k1, a0, (T)2 * k2 * t);
                b0 = cg_fma(v_, a0, cg_fma(vg_b * k1, t, b0));
            } else if constexpr (FAM == COVGRAM_EQ && !POW) c2 = -k1 * t;   // EQ: 2 k2 = -k1
            else c2 = (T)2 * k2 * t;
            csum += c2;
            const T mc2 = -c2;
            // ---- sweep 2: b += k1 a - c2 y'   (+ c2 x' through csum, once per row)
            bcast_sweep2<0, D, NP>(rec, b, k1, mc2);
        };
        double ra[NP], rb[NP];
        load_rec(ra, 0);
        T enya = exq[0], eyaa = exq[1];
        [[maybe_unused]] T a0a = (T)0, a0b = (T)0;
        if constexpr (VG) a0a = a0q[0];
        int jj = 0;
        for (; jj + 2 <= cnt; jj += 2) {
            // the next column's record goes in flight BEFORE this column's work (pinned: the scheduler otherwise sinks the loads to
            // where their registers are needed, half a column later)
            load_rec(rb, jj + 1);
            const T enyb = exq[2 * (jj + 1)], eyab = exq[2 * (jj + 1) + 1];
            if constexpr (VG) a0b = a0q[jj + 1];
            __builtin_amdgcn_sched_barrier(0);
            column(ra, enya, eyaa, a0a);
            __builtin_amdgcn_sched_barrier(0);
            load_rec(ra, jj + 2);
            enya = exq[2 * (jj + 2)]; eyaa = exq[2 * (jj + 2) + 1];
            if constexpr (VG) a0a = a0q[jj + 2];
            __builtin_amdgcn_sched_barrier(0);
            column(rb, enyb, eyab, a0b);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (jj < cnt) column(ra, enya, eyaa, a0a);
    };
    if constexpr (FAM == COVGRAM_MATERNP) {
        if (kp.p == 2) columns(std::integral_constant<int, 2>());
        else columns(std::integral_constant<int, -1>());
    } else {
        columns(std::integral_constant<int, -1>());
    }
#pragma unroll
    for (int l = 0; l < D; ++l) b[l] = cg_fma(x[l], csum, b[l]);

    if (!live) return;
    constexpr int VGI = VG ? 1 : 0;
    if (final_store) {
        T* yp = out + row * (int64_t)(d + VGI);
        if constexpr (VG) {
            T v = alpha0 * b0;
            if (beta != (T)0) v = cg_fma(beta, yp[0], v);
            yp[0] = v;
        }
#pragma unroll
        for (int l = 0; l < D; ++l) {
            if (l < d) {
                T v = alpha * b[l];
                if (beta != (T)0) v = cg_fma(beta, yp[VGI + l], v);
                yp[VGI + l] = v;
            }
        }
    } else {
        T* op = out + (int64_t)blockIdx.y * (D + VGI) * npad + row;    // partial slab [jsplit][D (+ 1: the value row)][npad], as grad_mvm_kernel
#pragma unroll
        for (int l = 0; l < D; ++l) op[(int64_t)l * npad] = b[l];
        if constexpr (VG) op[(int64_t)D * npad] = b0;
    }
}

// launches the kernel when (FAM, D) has an instance; false = not compiled for this shape (the caller runs grad_mvm_kernel)
template <int FAM, int D>
static bool launch_grad_bcast(const GradArgs& a) {
    if constexpr (grad_bcast_ok(D) && fam_is_iso<FAM> && !fam_is_expr<FAM> && FAM != COVGRAM_MATERN) {
        const typename ParamsOf<FAM, double>::type kp = make_params<FAM, double>(*a.hk);
        const int final_store = (a.jsplit == 1) ? 1 : 0;
#define CG_BCAST_LAUNCH(W, VGV)                                                                                                              \
        hipLaunchKernelGGL((grad_bcast_kernel<FAM, D, false, W, VGV>), dim3((unsigned)((a.n + 64 * W - 1) / (64 * W)), (unsigned)a.jsplit), dim3(64 * W), 0, \
                           a.stream, (const double*)a.X, a.n, a.d, (const double*)a.P, a.m, (double*)a.out, a.npad, a.jchunk, a.alpha, a.beta,         \
                           final_store, (const double*)a.C, kp, (const double*)a.Ex, (const double*)a.A0, a.alpha0, a.vg_c, a.vg_b)
        if (a.bcast == 4) { if (a.vg) CG_BCAST_LAUNCH(4, true); else CG_BCAST_LAUNCH(4, false); }
        else { if (a.vg) CG_BCAST_LAUNCH(1, true); else CG_BCAST_LAUNCH(1, false); }
#undef CG_BCAST_LAUNCH
        return true;
    } else {
        return false;
    }
}

}  // namespace covgram
