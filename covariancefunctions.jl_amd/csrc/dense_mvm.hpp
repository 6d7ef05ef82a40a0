// dense_mvm.hpp — the north-star kernel: y <- alpha * G(k; X, Y) * A + beta * y for a lazily
// represented n×m Gramian (reference hot loop: src/gramian.jl:78-99, entry src/gramian.jl:37-40,
// squared distance by DIRECT differences src/util.jl:40-47, dot product src/mercer.jl:3).
//
// MI355X mapping:
//   * one LANE owns R output rows (x_i lives in VGPRs, NRHS*R accumulators): there is no cross-lane
//     reduction anywhere in the pair loop;
//   * the column stream is wave-uniform and is read through the scalar data cache (s_load_dwordx8/x16),
//     so y_j and a_j are SGPR operands of the VALU instructions: no LDS, no VGPR copies, no barriers.
//     (An LDS-staged variant — 16 KiB double-buffered tiles read back as broadcast ds_read_b128 — was
//     built and measured 14 % slower on MI355X, profiles/r01_quickbench_variant_ab.txt, and removed.)
//   * fp32 processes TWO columns per instruction with packed math (v_pk_add/mul/fma_f32): the stream is
//     stored pair-interleaved, P[g] = (y_{2g,l}, y_{2g+1,l})_l, (a_{2g,c}, a_{2g+1,c})_c, so each 64-bit
//     SGPR pair is one packed operand and x_il is splat.  Measured issue cost per 64 pairs and SIMD
//     (tools/microbench.hip, profiles/r01_microbench_valu_rates.txt): 23.2 cycles packed vs 30.1 scalar;
//     v_exp_f32 (8.2 cycles, quarter rate) is the single most expensive instruction of the pair body;
//   * the grid is (row blocks) × (J splits): every workgroup owns a 256*R-row × jchunk-column rectangle,
//     partial sums go to a [jsplit][NRHS][npad] slab and a tiny second kernel applies alpha/beta
//     (deterministic; no float atomics);
//   * accumulation is two-level (512-column inner chunks, then per-split partials) so the fp32 error
//     against the fp64 oracle stays ~3e-7 at m = 2^17 (a sequential fp32 sum drifts to ~6e-6).
#pragma once
#include "profiles.hpp"

namespace covgram {

#ifndef CG_DENSE_THREADS
#define CG_DENSE_THREADS 64
#endif
// one wave per workgroup: the kernel uses no LDS and no barriers, so small workgroups only improve tail balance and
// shrink the partial slab (jsplit ~ waves * 64 R / n)
constexpr int DENSE_THREADS = CG_DENSE_THREADS;
constexpr int DENSE_INNER = 512;  // inner accumulation chunk (columns); even, so column pairs never straddle chunks

// Dimensions are consumed in chunks of one 64-byte scalar load; for rows wider than a chunk a scheduling
// barrier after each chunk keeps hipcc from hoisting every s_load of a column group to the top, which would
// overflow the ~100 usable SGPRs and spill them through v_writelane.
// The profile of the dense fp64 kernels: EQ / MaternP / Exponential read the exponential's table from the LDS copy their kernels
// fill first (profiles.hpp: exp_tab_lds); everything else is phi_value.
template <int FAM> constexpr bool dense_lds_tab = (FAM == COVGRAM_EQ || FAM == COVGRAM_MATERNP || FAM == COVGRAM_EXP || FAM == COVGRAM_RQ ||
                                                   FAM == COVGRAM_GAMMAEXP);
// what the sum of squares starts from: 2^-1000 for the fp64 MaternP profile (its square root then needs no zero test: profiles.hpp), else 0
template <int FAM, typename T> constexpr T dense_s0 = (FAM == COVGRAM_MATERNP && sizeof(T) == 8) ? (T)0x1p-1000 : (T)0;
template <int FAM, typename T, bool POW>
__device__ __forceinline__ T dense_phi(T s, const typename ParamsOf<FAM, T>::type& kp) {
    if constexpr (sizeof(T) == 8 && dense_lds_tab<FAM>) {
        T v;
        if constexpr (FAM == COVGRAM_EQ) v = exp2_neg_tab(s, exp_tab_lds());
        else if constexpr (FAM == COVGRAM_MATERNP) v = Phi<COVGRAM_MATERNP, T, true>::template eval_tab<true>(s, kp, exp_tab_lds());   // s starts from 2^-1000: dense_s0
        else if constexpr (FAM == COVGRAM_EXP) v = exp_neg_tab(cg_sqrt(s), exp_tab_lds());
        else if constexpr (FAM == COVGRAM_RQ) {                    // (1 + s / (2 alpha))^(-alpha), as Phi<RQ> / rq_pow with the table exponential
            const T u = cg_fma(s, kp.c0, (T)1);
            const T w = exp2_neg_prod_lds(log2_lds(u), kp.param);
            v = u <= 1.7e308 ? w : (u > 1.7e308 ? (T)0 : u);       // u = inf: 0; NaN: NaN
        } else {                                                   // exp(-s^(gamma/2) / 2), as Phi<GAMMAEXP> with the outer exponential on the table
            const T w = exp2_prod_lds(log2_lds(s), kp.param);
            const T t0 = (s > (T)0 && s <= 1.7e308) ? w : (s == (T)0 ? (T)0 : s);   // pow_pos: 0 -> 0, inf -> inf, NaN -> NaN
            const T t = (kp.param == (T)0) ? (T)1 : t0;
            v = exp_neg_half_lds(t);
        }
        if constexpr (POW) v = ipow(v, kp.power);
        return v;
    } else {
        return phi_value<FAM, T, dense_folded<FAM, T>, POW>(s, kp);
    }
}

template <typename T, int FAM, int D, int NR, int R, bool POW, bool ISO>
struct DenseBody {
    using PK = Pk<T>;
    using V = typename PK::V;
    static constexpr int DC = 64 / (int)sizeof(V);

    // s[r] = |x_r - y|^2 (or x_r . y) for one column group
    static __device__ __forceinline__ void dist(const V* __restrict__ p, const T (&x)[R][D], V (&s)[R]) {
#pragma unroll
        for (int c0 = 0; c0 < D; c0 += DC) {
#pragma unroll
            for (int l = c0; l < ((c0 + DC < D) ? c0 + DC : D); ++l) {
                const V yl = p[l];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const V xl = PK::splat(x[r][l]);
                    if constexpr (ISO) {
                        const V dl = xl - yl;
                        if constexpr (dense_s0<FAM, T> != (T)0) s[r] = PK::fma(dl, dl, (l == 0) ? PK::splat(dense_s0<FAM, T>) : s[r]);
                        else s[r] = (l == 0) ? dl * dl : PK::fma(dl, dl, s[r]);
                    } else {
                        s[r] = (l == 0) ? xl * yl : PK::fma(xl, yl, s[r]);
                    }
                }
            }
            if constexpr (D > DC) __builtin_amdgcn_sched_barrier(0);
        }
    }

    // Composite kernels: BG column groups at once, so that the interpreter's scalar loads and branches (one set per
    // factor) are paid per block instead of per pair (profiles.hpp: expr_value_block).
    template <int BG>
    static __device__ __forceinline__ void step_block(const V* __restrict__ p, const T (&x)[R][D], V (&acc)[R][NR],
                                                      const typename ParamsOf<FAM, T>::type& kp) {
        static_assert(R == 1, "block evaluation is written for one row per lane");
        V s[BG];
#pragma unroll
        for (int g = 0; g < BG; ++g) {
            V sg[R];
            dist(p + g * (D + NR), x, sg);
            s[g] = sg[0];
        }
        expr_accumulate_block<T, ISO, BG, NR>(s, kp, p + D, D + NR, acc[0]);
    }

    // One column group (PK::N columns) against R rows.  p points at the (uniform) packed record of the group.
    static __device__ __forceinline__ void step(const V* __restrict__ p, const T (&x)[R][D], V (&acc)[R][NR],
                                                const typename ParamsOf<FAM, T>::type& kp) {
        V s[R];
#pragma unroll
        for (int c0 = 0; c0 < D; c0 += DC) {
#pragma unroll
            for (int l = c0; l < ((c0 + DC < D) ? c0 + DC : D); ++l) {
                const V yl = p[l];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const V xl = PK::splat(x[r][l]);
                    if constexpr (ISO) {
                        const V dl = xl - yl;
                        if constexpr (dense_s0<FAM, T> != (T)0) s[r] = PK::fma(dl, dl, (l == 0) ? PK::splat(dense_s0<FAM, T>) : s[r]);
                        else s[r] = (l == 0) ? dl * dl : PK::fma(dl, dl, s[r]);
                    } else {
                        s[r] = (l == 0) ? xl * yl : PK::fma(xl, yl, s[r]);
                    }
                }
            }
            if constexpr (D > DC) __builtin_amdgcn_sched_barrier(0);
        }
        V aj[NR];
#pragma unroll
        for (int c = 0; c < NR; ++c) aj[c] = p[D + c];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const V kv = PK::map(s[r], [&](T sv) { return dense_phi<FAM, T, POW>(sv, kp); });
#pragma unroll
            for (int c = 0; c < NR; ++c) acc[r][c] = PK::fma(aj[c], kv, acc[r][c]);
        }
    }
};

template <typename T, int FAM, int D, int NR, int R, bool POW>
__global__ __launch_bounds__(DENSE_THREADS) void dense_mvm_kernel(
    const T* __restrict__ X, int64_t n, int32_t d, const typename Pk<T>::V* __restrict__ P, int64_t m,
    T* __restrict__ out, int64_t npad, int64_t ldy, int32_t nrhs, int64_t jchunk, T alpha, T beta,
    int32_t final_store, const T* __restrict__ Cn, const typename ParamsOf<FAM, T>::type kp, unsigned* __restrict__ tickets,
    T* __restrict__ yfinal) {
    // Cn: the common centre c (d scalars, the column point set's reference point) that isotropic kernels subtract from BOTH
    // sides before the pre-scale: (x - c) gamma - (y - c) gamma keeps the rounding of the scaled coordinates relative to the
    // cloud's extent, not to its distance from the origin (the reference subtracts first and scales after, src/util.jl:40-47).
    constexpr bool ISO = fam_is_iso<FAM>;
    if constexpr (sizeof(T) == 8 && dense_lds_tab<FAM>) exp_tab_lds_fill();
    if constexpr (sizeof(T) == 8 && (FAM == COVGRAM_RQ || FAM == COVGRAM_GAMMAEXP)) log_tab_lds_fill();
    using Body = DenseBody<T, FAM, D, NR, R, POW, ISO>;
    using PK = Pk<T>;
    using V = typename PK::V;
    constexpr int S = D + NR;                               // stream elements per column group
    constexpr int W = D * (int)sizeof(V) / 4;               // SGPRs per column group (coordinates)
    constexpr int GU = (W <= 8) ? 4 : ((W <= 32) ? 2 : 1);  // groups per unrolled step
    constexpr int GINNER = DENSE_INNER / PK::N;

    const int tid = threadIdx.x;
    const int64_t row_base = (int64_t)blockIdx.x * (DENSE_THREADS * R);
    const int64_t j0 = (int64_t)blockIdx.y * jchunk;
    const int64_t j1 = (j0 + jchunk < m) ? (j0 + jchunk) : m;
    const int64_t g0 = j0 / PK::N;                          // jchunk is a multiple of DENSE_INNER (even)
    const int64_t g1 = (j1 + PK::N - 1) / PK::N;            // the stream is padded to whole groups (a = 0)

    // rows of this lane: row_base + r*256 + tid  (coalesced across the wave for every r)
    T x[R][D];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t row = row_base + (int64_t)r * DENSE_THREADS + tid;
        if (row >= n) row = n - 1;  // clamp: computed but never stored
        const T* xr = X + row * (int64_t)d;
        if (d == D) {   // common case: no padding, straight loads
#pragma unroll
            for (int l = 0; l < D; ++l) x[r][l] = (ISO ? xr[l] - Cn[l] : xr[l]) * kp.gamma;
        } else {
#pragma unroll
            for (int l = 0; l < D; ++l) x[r][l] = (l < d) ? (ISO ? xr[l] - Cn[l] : xr[l]) * kp.gamma : (T)0;
        }
    }

    T tot[R][NR];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < NR; ++c) tot[r][c] = (T)0;

    // The column sweep as a lambda over the parameter block: MaternP orders p <= 3 (every order the reference's users reach in
    // practice) run it on a copy whose p the compiler can bound (p & 3), which folds the profile's per-pair "fixed-degree or looped
    // Horner" branch away — two scalar branches per pair in the fp64 loop otherwise.
    auto sweep = [&](const typename ParamsOf<FAM, T>::type& kp) {
    for (int64_t gb = g0; gb < g1; gb += GINNER) {
        const int cnt = (int)(((gb + GINNER < g1) ? (gb + GINNER) : g1) - gb);
        V acc[R][NR];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < NR; ++c) acc[r][c] = PK::splat((T)0);
        const V* __restrict__ p = P + gb * S;               // uniform address -> s_load_dwordxN
        int g = 0;
        if constexpr (fam_is_expr<FAM>) {
            constexpr int BG = 4;
            for (; g + BG <= cnt; g += BG, p += BG * S) Body::template step_block<BG>(p, x, acc, kp);
        } else {
            for (; g + GU <= cnt; g += GU, p += GU * S) {
#pragma unroll
                for (int u = 0; u < GU; ++u) Body::step(p + u * S, x, acc, kp);
            }
        }
        for (; g < cnt; ++g, p += S) Body::step(p, x, acc, kp);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < NR; ++c) tot[r][c] += PK::hsum(acc[r][c]);
    }
    };
    if constexpr (FAM == COVGRAM_MATERNP) {
        if (kp.p == 1 || kp.p == 2) {                        // the common orders: their own copies (the polynomial at its own degree)
            typename ParamsOf<FAM, T>::type kq = kp;
            if (kp.p == 1) { kq.p = 1; sweep(kq); } else { kq.p = 2; sweep(kq); }
        } else if (kp.p <= 3) {
            typename ParamsOf<FAM, T>::type kq = kp;
            kq.p = kp.p & 3;
            sweep(kq);
        } else {
            sweep(kp);
        }
    } else {
        sweep(kp);
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
                slab_store(out + ((int64_t)blockIdx.y * NR + c) * npad + row, tot[r][c], tickets != nullptr);
        }
    }
    // jsplit > 1 with tickets: the LAST workgroup of this row block to arrive adds the block's partials in dense_reduce_kernel's order
    // and applies alpha / beta (pack.hpp: last_arrival) — the separate reduce launch and its dependent-launch gap are gone
    if (!final_store && tickets != nullptr) {
        if (!last_arrival(tickets + blockIdx.x, gridDim.y)) return;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = row_base + (int64_t)r * DENSE_THREADS + tid;
            if (row >= n) continue;
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                if (c < nrhs) {
                    const T sum = ordered_split_sum<T>(out + (int64_t)c * npad + row, (int64_t)NR * npad, (int)gridDim.y);
                    T* yp = yfinal + row + (int64_t)c * ldy;
                    T v = alpha * sum;
                    if (beta != (T)0) v = cg_fma(beta, *yp, v);
                    *yp = v;
                }
            }
        }
    }
}

// y[i + c*ldy] = alpha * sum_s partial[s][c][i] + beta * y   (fixed summation order: deterministic)
template <typename T>
__global__ __launch_bounds__(256) void dense_reduce_kernel(const T* __restrict__ partial, int64_t npad, int32_t NRpad,
                                                           int32_t jsplit, T* __restrict__ y, int64_t n, int64_t ldy,
                                                           int32_t nrhs, T alpha, T beta) {
    // 64 rows per workgroup, the split index strided over the 4 waves (4 independent chains per row, 4x the workgroups of a
    // row-per-thread layout: a 16384-row shard at jsplit = 64 took 16.8 us with 64 workgroups of serial 64-long chains)
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    const int c = blockIdx.y;
    __shared__ T red[4][64];
    T s = (T)0;
    if (i < n && c < nrhs)
        for (int sp = part; sp < jsplit; sp += 4) s += partial[((int64_t)sp * NRpad + c) * npad + i];
    red[part][lane] = s;
    __syncthreads();
    if (part != 0 || i >= n || c >= nrhs) return;
    s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    T* yp = y + i + (int64_t)c * ldy;
    T v = alpha * s;
    if (beta != (T)0) v = cg_fma(beta, *yp, v);
    *yp = v;
}

// -------------------------------------------------------------------------------------------------
// gramian(k, x) in fp64 (the reference's default element type): the Gramian is symmetric and the reference does not use that
// (src/gramian.jl:78-87 loops over all n*m entries).  dense_sym_kernel keeps the lane-per-row layout and the scalar column stream,
// but a workgroup (one wave = one 64-row block rb) only walks the columns j >= 64 rb of its chunk: its own diagonal block in full
// (row sums only), and to the right of it every entry k_ij ONCE for both the row sum b_i += a_j k_ij and the column sum
// b_j += a_i k_ij.  A column sum is a reduction over the wave's 64 rows; done column by column it costs as much as the pair body
// (6 dependent DPP steps of two dword moves + an add: ~30 instructions against 27...46), so four columns are reduced TOGETHER by
// lane swaps (wave_sum4_f64 below: 21 instructions per four columns) and four lanes store the four totals to colslab[rb][j].  dense_sym_reduce_kernel then adds, per output row, the split-J partials of its row block and the
// column sums of all row blocks above it — fixed order, no float atomics.  Half the profile evaluations of dense_mvm_kernel.
// -------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);   // lanes without a source (or in masked rows) add 0.0
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
    return v + __hiloint2double(hi2, lo2);
}
// sum over the 64 lanes, delivered in lane 63 (the other lanes hold partial prefixes): an inclusive scan inside each row of 16
// lanes (row_shr 1, 2, 4, 8), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3 — 6 x (2 v_mov_dpp + v_add_f64),
// no v_readlane, no select.  Fixed order: deterministic.
__device__ __forceinline__ double wave_sum_lane63_f64(double v) {
    v = dpp_add_f64<0x111, 0xF>(v);
    v = dpp_add_f64<0x112, 0xF>(v);
    v = dpp_add_f64<0x114, 0xF>(v);
    v = dpp_add_f64<0x118, 0xF>(v);
    v = dpp_add_f64<0x142, 0xA>(v);
    v = dpp_add_f64<0x143, 0xC>(v);
    return v;
}

// Four columns at once (gfx950's lane-swap instructions; semantics checked on the device with tools/swap_probe.hip):
// v_permlane32_swap(a, b) leaves {a[0:31], b[0:31]} and {a[32:63], b[32:63]}, so their sum holds a's lane pairs (l, l + 32) in the lower
// half of the wave and b's in the upper half — ONE add reduces two columns by a factor two; v_permlane16_swap does the same between
// odd and even rows of 16 lanes.  After both, row r of the wave holds column (c0, c2, c1, c3)[r] summed over the four rows, and the
// in-row scan finishes all four columns together: 21 instructions per FOUR columns (6 swaps of two dwords... + 3 adds + 4 x 3 DPP).
__device__ __forceinline__ double swap32_add_f64(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double swap16_add_f64(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
template <int CTRL>
__device__ __forceinline__ double row_shr_add_f64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);        // bound_ctrl: lanes without a source read 0
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return v + __hiloint2double(hi2, lo2);
}
// totals of four columns' per-lane terms: lane 15 / 31 / 47 / 63 returns the wave total of c0 / c2 / c1 / c3
__device__ __forceinline__ double wave_sum4_f64(double c0, double c1, double c2, double c3) {
    double m = swap16_add_f64(swap32_add_f64(c0, c1), swap32_add_f64(c2, c3));
    m = row_shr_add_f64<0x111>(m);
    m = row_shr_add_f64<0x112>(m);
    m = row_shr_add_f64<0x114>(m);
    m = row_shr_add_f64<0x118>(m);
    return m;
}

template <int FAM, int D>
__global__ __launch_bounds__(DENSE_THREADS) void dense_sym_kernel(
    const double* __restrict__ X, int64_t n, int32_t d, const double* __restrict__ P, double* __restrict__ out,
    double* __restrict__ colslab, int64_t npad, int64_t jchunk, const double* __restrict__ Cn,
    const typename ParamsOf<FAM, double>::type kp0, int32_t rb_first, int32_t rb_stride) {
    using T = double;
    constexpr bool ISO = fam_is_iso<FAM>;
    if constexpr (dense_lds_tab<FAM>) exp_tab_lds_fill();          // before the early return below: every thread reaches the barrier
    if constexpr (FAM == COVGRAM_RQ || FAM == COVGRAM_GAMMAEXP) log_tab_lds_fill();
    using Body = DenseBody<T, FAM, D, 1, 1, false, ISO>;
    constexpr int S = D + 1;
    const int lane = threadIdx.x;
    // workgroup x takes the 64-row block rb_first + x rb_stride (all blocks: 0, 1; rank r of P in the multi-GPU form: r, P — a cyclic
    // assignment gives every rank the same share of the triangle); its column sums go to row x of the launch's own slab
    const int64_t row_lo = ((int64_t)rb_first + (int64_t)blockIdx.x * rb_stride) * 64;
    const int64_t j0 = (int64_t)blockIdx.y * jchunk;
    const int64_t j1 = (j0 + jchunk < n) ? (j0 + jchunk) : n;
    const int64_t row = row_lo + lane;
    if (j1 <= row_lo) {                                    // the whole chunk lies left of the diagonal block: another row block's column sums
        out[(int64_t)blockIdx.y * npad + row] = 0.0;
        return;
    }
    const int64_t rowc = (row < n) ? row : n - 1;          // clamp: computed, weighted 0 in the column sums, never stored as a row
    T x[1][D];
    {
        const T* xr = X + rowc * (int64_t)d;
#pragma unroll
        for (int l = 0; l < D; ++l) x[0][l] = (l < d) ? (ISO ? xr[l] - Cn[l] : xr[l]) * kp0.gamma : (T)0;
    }
    const T ai = (row < n) ? P[rowc * S + D] : (T)0;       // the row's own weight (the stream holds a_j beside the scaled y_j)
    T tot = (T)0;

    auto sweep = [&](const typename ParamsOf<FAM, T>::type& kp) {
        const int64_t jstart = (j0 > row_lo) ? j0 : row_lo;          // both multiples of 64
        for (int64_t jb = jstart; jb < j1; jb += 64) {
            const int cnt = (int)((jb + 64 < j1) ? 64 : (j1 - jb));
            const T* __restrict__ p = P + jb * S;                    // uniform address -> s_load
            T acc = (T)0;
            if (jb == row_lo) {                                      // diagonal block: all 64 x cnt entries, row sums only
                for (int u = 0; u < cnt; ++u, p += S) {
                    T sv[1];
                    Body::dist(p, x, sv);
                    const T kv = dense_phi<FAM, T, false>(sv[0], kp);
                    acc = cg_fma(p[D], kv, acc);
                }
            } else {
                T* __restrict__ cdst = colslab + (int64_t)blockIdx.x * npad + jb;
                auto column = [&](const T* __restrict__ pc, int u) {
                    T sv[1];
                    Body::dist(pc, x, sv);
                    const T kv = dense_phi<FAM, T, false>(sv[0], kp);
                    acc = cg_fma(pc[D], kv, acc);
                    const T c = wave_sum_lane63_f64(ai * kv);
                    if (lane == 63) cdst[u] = c;                     // one 8-byte store per column; the 64 of a block merge in L2
                };
                if (cnt == 64) {
                    const int cmap = ((lane >> 4) & 1) * 2 + (lane >> 5);       // which of the four columns this lane's row ends up holding
                    for (int u = 0; u < 64; u += 4, p += 4 * S) {
                        T c[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            T sv[1];
                            Body::dist(p + q * S, x, sv);
                            const T kv = dense_phi<FAM, T, false>(sv[0], kp);
                            acc = cg_fma(p[q * S + D], kv, acc);
                            c[q] = ai * kv;
                        }
                        const T tsum = wave_sum4_f64(c[0], c[1], c[2], c[3]);
                        if ((lane & 15) == 15) cdst[u + cmap] = tsum;
                    }
                } else {
                    for (int u = 0; u < cnt; ++u, p += S) column(p, u);
                }
            }
            tot += acc;
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
    out[(int64_t)blockIdx.y * npad + row] = tot;
}

// y[i] = alpha * (sum_sp out[sp][i] + sum_{rb < i / 64} colslab[rb][i]) + beta * y[i]   (fixed order: deterministic)
// Partial form (rb_stride > 1): only the row blocks first, first + stride, ... were evaluated by this launch — rows of other blocks
// get their column-sum terms only, and slab row x belongs to row block first + x stride.
template <typename T /* double */>
__global__ __launch_bounds__(1024) void dense_sym_reduce_kernel(const double* __restrict__ out, const double* __restrict__ colslab,
                                                                int64_t npad, int32_t jsplit, double* __restrict__ y, int64_t n,
                                                                double alpha, double beta, int32_t rb_first, int32_t rb_stride) {
    // 64 rows per workgroup = one row block; TWO consecutive rows per lane (16-byte loads — round 5; 8-byte loads read the slabs at 1.5 TB/s: 11.7 us of the
    // README case's 224), the terms strided over 32 half-waves (the last row blocks add n / 64 column-sum rows each: many independent loads in flight).
    // Per row: terms part, part + 32, ... in order, then the 32 parts in a fixed tree — deterministic.  (npad is a multiple of 64; slabs are 256-byte aligned.)
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int64_t i = (int64_t)blockIdx.x * 64 + 2 * lane;
    __shared__ d2 red[32][32];
    d2 s = {0.0, 0.0};
    if (i < n) {
        const int64_t B = blockIdx.x;
        if (B >= rb_first && (B - rb_first) % rb_stride == 0)          // this launch evaluated the block's own rows
            for (int sp = part; sp < jsplit; sp += 32) s += *reinterpret_cast<const d2*>(out + (int64_t)sp * npad + i);
        // evaluated row blocks above this one hold column sums for these rows: first + x stride < B
        const int64_t nb = B > rb_first ? (B - rb_first + rb_stride - 1) / rb_stride : 0;
#pragma unroll 4
        for (int64_t rb = part; rb < nb; rb += 32) s += *reinterpret_cast<const d2*>(colslab + rb * npad + i);
    }
    red[part][lane] = s;
    __syncthreads();
    if (part != 0 || i >= n) return;
    d2 t = {0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 32; q += 4) t += (red[q][lane] + red[q + 1][lane]) + (red[q + 2][lane] + red[q + 3][lane]);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (i + e >= n) break;
        double v = alpha * t[e];
        if (beta != 0.0) v = cg_fma(beta, y[i + e], v);
        y[i + e] = v;
    }
}

template <int FAM, int D>
static int launch_dense_sym_one(const DenseArgs& a) {
    // the kernel, its reduce kernel and covgram_mvm's slab sizes are written for 64-row blocks and one wave per workgroup
    static_assert(DENSE_THREADS == 64, "dense_sym_kernel: 64-row blocks, one wave per workgroup (CG_DENSE_THREADS must stay 64)");
    const typename ParamsOf<FAM, double>::type kp = make_params<FAM, double>(*a.hk);
    const int64_t blocks = (a.n + 63) / 64;
    const int64_t mine = a.sym_first < blocks ? (blocks - a.sym_first + a.sym_stride - 1) / a.sym_stride : 0;
    if (mine == 0) return COVGRAM_OK;
    dim3 grid((unsigned)mine, (unsigned)a.jsplit);
    hipLaunchKernelGGL((dense_sym_kernel<FAM, D>), grid, dim3(DENSE_THREADS), 0, a.stream, (const double*)a.X, a.n, a.d,
                       (const double*)a.P, (double*)a.out, (double*)a.colslab, a.npad, a.jchunk, (const double*)a.C, kp, a.sym_first, a.sym_stride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dense_sym launch failed: %s", hipGetErrorString(e)); return COVGRAM_EHIP; }
    return COVGRAM_OK;
}

// Column stream, PK columns per group g:  P[(g*(D+NR) + l)*PK + h] = gamma * Y[g*PK+h][l]   (l < D, zero padded in l)
//                                         P[(g*(D+NR) + D + c)*PK + h] = A[g*PK+h + (c0+c)*lda]
// A column beyond m (odd m, fp32) duplicates the last point with weight 0, so it adds exactly 0 * phi(finite).
template <typename T>
__global__ __launch_bounds__(256) void dense_pack_kernel(const T* __restrict__ Y, int64_t m, int32_t d, const T* __restrict__ A,
                                                         int64_t lda, int32_t nrhs, int32_t c0, T* __restrict__ P, int32_t D,
                                                         int32_t NR, int32_t PKN, T gamma, const T* __restrict__ Cn, int32_t PADTO) {
    // PADTO: a multiple of PKN — the stream ends on a whole group of PADTO columns (dense_sym32_kernel: 8)
    const int64_t jj = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // padded column index
    const int64_t mp = ((m + PADTO - 1) / PADTO) * PADTO;
    if (jj >= mp) return;
    const bool pad = jj >= m;
    const int64_t j = pad ? (m - 1) : jj;
    const int64_t g = jj / PKN;
    const int h = (int)(jj - g * PKN);
    T* p = P + g * (int64_t)(D + NR) * PKN + h;
    for (int l = 0; l < D; ++l) p[(int64_t)l * PKN] = (l < d) ? (Y[j * (int64_t)d + l] - (Cn ? Cn[l] : (T)0)) * gamma : (T)0;
    for (int c = 0; c < NR; ++c) p[(int64_t)(D + c) * PKN] = (!pad && c0 + c < nrhs) ? A[j + (int64_t)(c0 + c) * lda] : (T)0;
}

// -------------------------------------------------------------------------------------------------
// launcher for one family (instantiated per translation unit, see dense_fam.hip)
// -------------------------------------------------------------------------------------------------
template <typename T, int FAM, int D, int NR, int R, bool POW>
static int launch_dense_one(const DenseArgs& a) {
    const typename ParamsOf<FAM, T>::type kp = make_params<FAM, T>(*a.hk);
    const int64_t rows_per_wg = (int64_t)DENSE_THREADS * R;
    dim3 grid((unsigned)((a.n + rows_per_wg - 1) / rows_per_wg), (unsigned)a.jsplit);
    const int final_store = (a.jsplit == 1) ? 1 : 0;
    hipLaunchKernelGGL((dense_mvm_kernel<T, FAM, D, NR, R, POW>), grid, dim3(DENSE_THREADS), (size_t)a.lds_pad, a.stream, (const T*)a.X, a.n,
                       a.d, (const typename Pk<T>::V*)a.P, a.m, (T*)a.out, a.npad, a.ldy, a.nrhs, a.jchunk, (T)a.alpha,
                       (T)a.beta, final_store, (const T*)a.C, kp, a.tickets, (T*)a.yfinal);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dense_mvm launch failed: %s", hipGetErrorString(e)); return COVGRAM_EHIP; }
    return COVGRAM_OK;
}

// Rows per lane.  R = 1, 2 and 4 were measured on C2 at equal wave counts (profiles/r01_quickbench_rows_per_lane.txt):
// 2.53-2.62 ms for all three — SGPR operands make the column stream free to re-read, so more rows per lane buy nothing,
// while R = 1 gives 4x more row blocks, i.e. a 4x smaller split-J partial slab (8.4 MB instead of 33.5 MB on C2).
template <int D> struct RowsFor { static constexpr int value = 1; };
inline int rows_per_lane_for(int) { return 1; }

template <int FAM, int D> static int launch_dense_sym32_one(const DenseArgs& a);   // dense_sym32.hpp
template <int FAM, int D> static int launch_dense_bcast_one(const DenseArgs& a);   // dense_bcast.hpp

template <typename T, int FAM, int D, int NR>
static int launch_dense_D(const DenseArgs& a) {
    constexpr int R = RowsFor<D>::value;
    if constexpr (sizeof(T) == 8 && NR == 1 && !fam_is_expr<FAM>) {
        if (a.bcast) return launch_dense_bcast_one<FAM, D>(a);
        if (a.sym) return launch_dense_sym_one<FAM, D>(a);
    }
    if constexpr (sizeof(T) == 4 && NR == 1 && !fam_is_expr<FAM>)
        if (a.sym) return launch_dense_sym32_one<FAM, D>(a);
    const bool pow = a.hk->k.power != 1;
    if constexpr (!fam_is_expr<FAM>)   // composites apply Power per factor
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

}  // namespace covgram
