// grad_mvm.hpp — O(n m d) MVM with the Gramian of a GradientKernel (reference: blockmul!
// src/gramian.jl:241-253, block mul! src/gradient.jl:86-92 (IsotropicInput) and :109-115
// (DotProductInput), phi', phi'' src/gradient.jl:584-600 as closed forms).
//
//   isotropic:    b_i += alpha * -2 (k1 a_j + 2 k2 r (r.a_j)),   r = x_i - y_j, (k1,k2) = (phi', phi'')(|r|^2)
//   dot product:  b_i += alpha *    (k1 a_j +   k2 y_j (x_i.a_j)),              (k1,k2) = (phi', phi'')(x_i.y_j)
//
// MI355X mapping: one lane owns one block row (x_i and the d-vector accumulator b_i live in VGPRs);
// the column stream P[j] = (gamma*y_j[0..D), a_j[0..D)) is wave-uniform and arrives through the
// scalar data cache as SGPR operands, so the 4-5 d flops per block contain no cross-lane traffic, no
// LDS and no VGPR copies.  With gamma = 1/l the chain rule gives
//   b = -2 gamma^2 (psi' a + 2 psi'' r' (r'.a)),  r' = gamma r,  psi(s') = phi(s'/gamma^2),
// so the host folds -2 gamma^2 * scale into alpha.  Grid = (row blocks) × (J splits) with a
// deterministic second-pass reduction, as in dense_mvm.hpp.
//
// Expanded form (EXPD; fp64 isotropic Gramians inside a radius gate).  The reference's block needs r = x_i - y_j twice: 6 fp64
// VALU instructions per dimension and pair (sub + 2 fma in each sweep).  With the norms |x'_i|^2, |y'_j|^2 and the column scalar
// y'_j . a_j known (one pass per point set / per MVM), the same block is
//     s = |x'|^2 + |y'|^2 - 2 x'.y',   t = x'.a_j - y'_j.a_j,   b_i += k1 a_j - c2 y'_j   (+ x'_i * sum_j c2_j once per row):
// 4 fma per dimension and pair, a third fewer instructions on a kernel that is fp64-VALU bound.  s and c2 (x' - y') are then
// differences of O(|x'|^2) terms instead of direct differences (src/util.jl:40-47): the absolute error is ~1e-16 R^2 (R = the
// radius of the pre-scaled cloud about the common centre), so the form is used while R^2 <= GRAD_EXPAND_GATE = 1000 — C4:
// R^2 ~ 90, measured rel-err 3e-15.  fp32 (round 5): the same form while R^2 <= GRAD_EXPAND_GATE_F32 = 128 — s then carries a few fp32 roundings of R^2, as the fp32
// matrix-core dense kernels do inside their gate; a third fewer VALU instructions on a kernel that is 87 % VALU-busy (profiles/r05_grad32_pmc.txt).
// Option "grad_expand": -1 these rules, 0 never, 1 always (tests).
//
// Software pipeline of the scalar stream.  The state (x_i, b_i [, r]) costs 2-3 d-vectors of VGPRs, so wide fp64 rows
// run at 2 waves per SIMD and cannot hide scalar-load latency by occupancy: rocprofv3 showed 61 % of the wave cycles of
// the first version parked in s_waitcnt (profiles/r01_pmc_counters_v1.txt).  SMEM returns out of order, so only
// lgkmcnt(0) exists: a load can overlap compute only if it is ISSUED AFTER the wait that validates the current chunk.
// Each 64-byte chunk step is therefore pinned (scheduling barriers) as
//     touch the current chunk (the compiler's s_waitcnt lands here)  ->  issue the s_load of the NEXT chunk  ->
//     the rest of the current chunk's VALU work
// across both sweeps of a column and across columns (the last chunk of column j prefetches the first of j+1; the
// stream is padded by one record so the final prefetch stays in bounds).
#pragma once
#include "profiles.hpp"

namespace covgram {

constexpr int GRAD_THREADS = 256;   // largest workgroup; rows are padded to this
// Waves per workgroup: the waves of a workgroup take consecutive 64-row blocks and walk the SAME column chunk from the same
// start, so their record lines meet in the scalar cache (no LDS, no barrier).  Four waves per workgroup measured 2-3 % faster
// than one wherever the kernel runs at >= 3 waves per SIMD (C4 1.829 -> 1.789 ms, d = 48 -3 %, value-gradient -2.7 %; d <= 8
// neutral) and 2 % slower on the two-waves-per-SIMD kernels (wide fp64 RQ / gamma-exponential), which keep one
// (tools/c4_lib_ab.py, profiles/r02_c4_exp_ab.txt).
inline int grad_waves_per_simd(int elem_size, int D, int fam) {
    const int state = 2 * D * (elem_size / 4);
    const int temps = (elem_size == 8 && (fam == COVGRAM_RQ || fam == COVGRAM_GAMMAEXP || fam >= COVGRAM_NFAMILY)) ? 110 : 40;
    const int w = 512 / (state + temps);
    return w < 1 ? 1 : (w > 8 ? 8 : w);
}
inline int grad_block_threads(int elem_size, int D, int fam) {
    const int state = 2 * D * (elem_size / 4);
    const int temps = (elem_size == 8 && (fam == COVGRAM_RQ || fam == COVGRAM_GAMMAEXP || fam >= COVGRAM_NFAMILY)) ? 110 : 40;
    return (512 / (state + temps) >= 3) ? GRAD_THREADS : 64;
}

// Two right-hand sides per pass (grad_mvm_kernel<..., NR = 2>): x and two accumulators are 3 d-vectors of VGPRs — compiled while they leave
// room for the temporaries at two waves per SIMD (fp64 d <= 32, fp32 d <= 64), for the single profiles (the composite interpreter keeps its
// jets live and stays at one right-hand side).  Host and device agree through this one function.
constexpr bool grad_two_rhs_ok(size_t elem_size, int D, int fam) {
    return fam < COVGRAM_NFAMILY && 3 * D * (int)(elem_size / 4) + (elem_size == 8 && (fam == COVGRAM_RQ || fam == COVGRAM_GAMMAEXP) ? 110 : 40) <= 256;
}

template <typename T, int DC, int NR = 1>
struct GradChunk {
    T y[DC];
    T a[NR][DC];
};

// Register budget: the state is 2 (or 3 with KEEP_R) d-vectors; ask the allocator for the occupancy that state allows
// with ~40 VGPRs of temporaries (e.g. fp64 d = 32 without r: 128 + 40 = 168 -> 3 waves per SIMD instead of 2).
// fp64 profiles built on the library's log / pow (RQ, gamma-exponential) keep ~20 polynomial coefficients live: with 40
// registers of temporaries they spilled to scratch inside the column loop (RQ at the C4 shape: 6.0 ms, 40 VGPRs spilled,
// every reload a serial vmcnt(0) wait) — those families get 110, i.e. one wave per SIMD less and no spills.
template <typename T, int FAM> constexpr int grad_temp_regs() {
    return (sizeof(T) == 8 && (FAM == COVGRAM_RQ || FAM == COVGRAM_GAMMAEXP || fam_is_expr<FAM>)) ? 110 : 40;
}
template <typename T, int D, bool KEEP_R, int FAM = COVGRAM_EQ, int NR = 1>
constexpr int grad_min_waves() {
    const int state = ((KEEP_R ? 2 : 1) + NR) * D * (int)(sizeof(T) / 4);
    const int w = 512 / (state + grad_temp_regs<T, FAM>());
    return w < 1 ? 1 : (w > 8 ? 8 : w);
}

// The jets of the gradient kernel.  fp64 rational-quadratic and gamma-exponential: their log2 / exp2 on the LDS tables the kernel fills
// first (profiles.hpp: log2_lds, exp2_neg_prod_lds, exp2_prod_lds, exp_neg_half_lds — 15 + 13 instructions for the 28 + 23 of the
// polynomial forms; these profiles' arithmetic, not the scalar stream, is what separates their MVM from EQ's); everything else: phi_jet.
template <int FAM, typename T> constexpr bool grad_lds_tab = sizeof(T) == 8 && (FAM == COVGRAM_RQ || FAM == COVGRAM_GAMMAEXP || FAM == COVGRAM_MATERNP);
template <int FAM, typename T, bool POW, int PFIX>
__device__ __forceinline__ void grad_jet(T s, const typename ParamsOf<FAM, T>::type& kp, T& v, T& d1, T& d2) {
    if constexpr (grad_lds_tab<FAM, T> && FAM == COVGRAM_RQ) {                   // DPhi<RQ> with the table power
        const T a = kp.param;
        const T u = cg_fma(s, kp.c0, (T)1);
        const T iu = cg_rcp(u);
        const T w = exp2_neg_prod_lds(log2_lds(u), a);
        v = u <= 1.7e308 ? w : (u > 1.7e308 ? (T)0 : u);                       // u = inf: 0; NaN: NaN (rq_pow)
        d1 = (T)-0.5 * v * iu;
        d2 = (a + (T)1) * ((T)0.5 * kp.c0) * v * iu * iu;
        if constexpr (POW) power_jet(kp.power, v, d1, d2);
    } else if constexpr (grad_lds_tab<FAM, T> && FAM == COVGRAM_GAMMAEXP) {      // DPhi<GAMMAEXP> with the table power and exponential
        const T g = kp.param;
        const T w = exp2_prod_lds(log2_lds(s), g);
        const T p0 = (s > (T)0 && s <= 1.7e308) ? w : (s == (T)0 ? (T)0 : s);   // pow_pos: 0 -> 0, inf -> inf, NaN -> NaN
        const T sg = (g == (T)0) ? (T)1 : p0;
        const T is = cg_rcp(s);
        v = exp_neg_half_lds(sg);
        const T hg = (T)0.5 * g;
        d1 = -hg * sg * is * v;
        d2 = v * (hg * hg * sg * sg * is * is - hg * (g - (T)1) * sg * is * is);
        if constexpr (POW) power_jet(kp.power, v, d1, d2);
    } else if constexpr (grad_lds_tab<FAM, T> && FAM == COVGRAM_MATERNP) {         // DPhi<MATERNP> with exp(-r) on the table
        DPhi<FAM, T>::template eval<PFIX, true>(s, kp, v, d1, d2);
        if constexpr (POW) power_jet(kp.power, v, d1, d2);
    } else {
        phi_jet<FAM, T, POW, PFIX>(s, kp, v, d1, d2);
    }
}

// VG = true is the ValueGradientKernel Gramian (src/gradient.jl:400-474, block mul! :319-351): blocks of d+1 with a value
// row/column.  In the pre-scaled coordinates (a_j = (a0, av), t = r'.av or x'.av):
//   isotropic:    bv = -2 gamma^2 (psi' av + (2 psi'' t - psi' a0 / gamma) r'),   b0 = psi a0 - 2 gamma psi' t
//   dot product:  bv =    gamma^2 (phi' av + (  phi'' t + phi' a0 / gamma) y'),   b0 = phi a0 +   gamma phi' t
// i.e. one extra FMA on c2 and one scalar accumulator; a0 streams from A0 (one scalar load per column, prefetched with
// the column's first chunk), vg_c = -+1/gamma, vg_b = -2 gamma | gamma, and b0 is scaled by alpha0 = alpha * scale.
// NR right-hand sides at once (src/gramian.jl:241-257 takes vectors of matrices, the block mul! of src/gradient.jl:86-92 broadcasts over the
// columns): r, s, phi', phi'' are evaluated ONCE per pair and feed NR accumulators b_i; the record of column j is (y_j, a_j^(0), ..., a_j^(NR-1)),
// the value weights / expanded-form scalars are NR (1 + NR) per column, the outputs NR vectors ldy apart (partial slabs: [split][NR][D (+1)][npad]).
template <typename T, int FAM, int D, bool KEEP_R, bool POW, bool VG, bool EXPD = false, int NR = 1>
__global__ __launch_bounds__(GRAD_THREADS, (grad_min_waves<T, D, KEEP_R, FAM, NR>())) void grad_mvm_kernel(const T* __restrict__ X, int64_t n, int32_t d,
                                                                const T* __restrict__ P, const T* __restrict__ P2,
                                                                int64_t m, T* __restrict__ out, int64_t npad,
                                                                int64_t jchunk, T alpha, T beta, int32_t final_store,
                                                                const T* __restrict__ A0, T alpha0, T vg_c, T vg_b,
                                                                const T* __restrict__ Cn, const typename ParamsOf<FAM, T>::type kp,
                                                                const T* __restrict__ Ex, int64_t ldy) {
    constexpr bool ISO = fam_is_iso<FAM>;
    if constexpr (grad_lds_tab<FAM, T>) { exp_tab_lds_fill(); if constexpr (FAM != COVGRAM_MATERNP) log_tab_lds_fill(); }
    constexpr int RS = (1 + NR) * D;                                             // scalars per column record
    static_assert(!EXPD || (ISO && !KEEP_R), "the expanded form is an isotropic variant that keeps no r");
    // dims per chunk: one 64-byte s_load per operand; 32-byte loads with two right-hand sides — a chunk is (1 + NR) operands and two chunks
    // are live (current + prefetched): 2 x 3 x 16 dwords would not fit the 102 SGPRs
    constexpr int DCB = (NR > 1 ? 32 : 64) / (int)sizeof(T);
    constexpr int DC = (DCB < D) ? DCB : D;
    constexpr int NC = (D + DC - 1) / DC;
    constexpr bool NEED_Y2 = !ISO || !KEEP_R;                                    // sweep 2 needs y_j again (EXPD: -c2 y_j)
    using Chunk = GradChunk<T, DC, NR>;

    const int tid = threadIdx.x;
    int64_t row = (int64_t)blockIdx.x * blockDim.x + tid;
    const bool live = row < n;
    if (!live) row = n - 1;
    const int64_t j0 = (int64_t)blockIdx.y * jchunk;
    const int64_t j1 = (j0 + jchunk < m) ? (j0 + jchunk) : m;

    T x[D], b[NR][D];
    {
        const T* xr = X + row * (int64_t)d;
        if (d == D) {   // common case: no padding, straight vector loads
#pragma unroll
            for (int l = 0; l < D; ++l) x[l] = (ISO ? xr[l] - Cn[l] : xr[l]) * kp.gamma;      // common centre: dense_mvm.hpp
        } else {
#pragma unroll
            for (int l = 0; l < D; ++l) x[l] = (l < d) ? (ISO ? xr[l] - Cn[l] : xr[l]) * kp.gamma : (T)0;
        }
#pragma unroll
        for (int rr = 0; rr < NR; ++rr)
#pragma unroll
            for (int l = 0; l < D; ++l) b[rr][l] = (T)0;
    }
    [[maybe_unused]] T nx = (T)0, csum[NR] = {};                    // EXPD: |x'_i|^2 and sum_j c2_j (b_i += x'_i csum at the end)
    if constexpr (EXPD) {
#pragma unroll
        for (int l = 0; l < D; ++l) nx = cg_fma(x[l], x[l], nx);
    }

    // chunk loaders: uniform addresses -> s_load_dwordx16.  The tail chunk of a D that is not a multiple of DC is
    // clamped into the record (it overlaps the previous chunk; the overlapped lanes are skipped by the consumer).
    auto load_ya = [&](const T* __restrict__ rec, int c) {
        Chunk ch;
        const int base = (c * DC + DC <= D) ? c * DC : D - DC;
#pragma unroll
        for (int e = 0; e < DC; ++e) {
            ch.y[e] = rec[base + e];
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) ch.a[rr][e] = rec[(1 + rr) * D + base + e];
        }
        return ch;
    };
    auto load_a = [&](const T* __restrict__ rec, int c) {
        Chunk ch;
        const int base = (c * DC + DC <= D) ? c * DC : D - DC;
#pragma unroll
        for (int e = 0; e < DC; ++e) {
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) ch.a[rr][e] = rec[(1 + rr) * D + base + e];
            if constexpr (NEED_Y2) ch.y[e] = rec[base + e];
            else ch.y[e] = (T)0;
        }
        return ch;
    };

    // Sweep 2 reads the stream through P2 — the SAME address passed as a second kernel argument — so the compiler cannot
    // merge its loads with sweep 1's (it would re-materialise them right before use instead of prefetching them).
    const int cnt = (int)(j1 - j0);
    const T* __restrict__ p = P + j0 * RS;
    const T* __restrict__ q = P2 + j0 * RS;
    Chunk cur = load_ya(p, 0);
    const T* __restrict__ a0p = VG ? A0 + NR * j0 : nullptr;       // NR value weights per column
    T a0[NR] = {}, a0n[NR] = {}, b0[NR] = {};
    if constexpr (VG) {
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) a0[rr] = a0p[rr];
    }
    // EXPD: the column scalars (|y'_j|^2, y'_j . a_j) stream beside the records, prefetched one column ahead like a0
    const T* __restrict__ exq = EXPD ? Ex + (1 + NR) * j0 : nullptr;   // (|y'_j|^2, y'_j . a_j^(0), ...)
    [[maybe_unused]] T eny = (T)0, enyn = (T)0, eya[NR] = {}, eyan[NR] = {};
    if constexpr (EXPD) {
        eny = exq[0];
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) eya[rr] = exq[1 + rr];
    }
    // the column loop as a generic lambda: MaternP kernels run it with the order fixed at compile time when it is 2 (nu = 5/2)
    auto columns = [&](auto pfix_) {
    constexpr int PFIX = decltype(pfix_)::value;
    for (int jj = 0; jj < cnt; ++jj, p += RS, q += RS) {
        T s = (T)0, t[NR] = {};
        T r[(ISO && KEEP_R) ? D : 1];
        // ---- sweep 1: s = |r|^2 (or x.y), t = r.a (or x.a) ------------------------------------------------------
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int base = (c * DC + DC <= D) ? c * DC : D - DC;
            const int skip = c * DC - base;     // lanes of a clamped tail chunk that chunk c-1 already consumed
            Chunk nxt;
#pragma unroll
            for (int e = 0; e < DC; ++e) {
                if (e < skip) continue;
                const int l = base + e;
                if constexpr (EXPD) {
                    s = cg_fma(x[l], cur.y[e], s);
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) t[rr] = cg_fma(x[l], cur.a[rr][e], t[rr]);
                } else if constexpr (ISO) {
                    const T rl = x[l] - cur.y[e];
                    if constexpr (KEEP_R) r[l] = rl;
                    s = cg_fma(rl, rl, s);
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) t[rr] = cg_fma(rl, cur.a[rr][e], t[rr]);
                } else {
                    s = cg_fma(x[l], cur.y[e], s);
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) t[rr] = cg_fma(x[l], cur.a[rr][e], t[rr]);
                }
                if (e == skip) {   // the first use above carried the wait; now put the next chunk in flight
                    __builtin_amdgcn_sched_barrier(0);
                    nxt = (c + 1 < NC) ? load_ya(p, c + 1) : load_a(q, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            asm("" : "+v"(s));            // pin the reductions at the chunk boundary (bounds VGPR live ranges)
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) asm("" : "+v"(t[rr]));
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
        if constexpr (EXPD) {                                     // x'.y' -> |x' - y'|^2 (never negative), x'.a -> r'.a
            s = cg_fma((T)-2, s, nx + eny);
            s = (s < (T)0) ? (T)0 : s;      // rounding may take s a few ulp below zero; a NaN coordinate stays a NaN (as on the direct-difference path)
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) t[rr] -= eya[rr];
        }
        T k1, k2, c2[NR];
        if constexpr (VG) {
            T v;
            grad_jet<FAM, T, POW, PFIX>(s, kp, v, k1, k2);
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                c2[rr] = cg_fma(vg_c * k1, a0[rr], ISO ? (T)2 * k2 * t[rr] : k2 * t[rr]);
                b0[rr] = cg_fma(v, a0[rr], cg_fma(vg_b * k1, t[rr], b0[rr]));
            }
        } else {
            { T v_; grad_jet<FAM, T, POW, PFIX>(s, kp, v_, k1, k2); }
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                if constexpr (FAM == COVGRAM_EQ && !POW) c2[rr] = -k1 * t[rr];       // EQ: 2 k2 = -k1 (both exact scalings of the same exponential)
                else c2[rr] = ISO ? (T)2 * k2 * t[rr] : k2 * t[rr];
            }
        }
        if constexpr (EXPD) {                                     // b += k1 a - c2 y' here, + c2 x' through csum
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) { csum[rr] += c2[rr]; c2[rr] = -c2[rr]; }
        }
        // ---- sweep 2: b += k1 a + c2 r   (or k1 a + c2 y) --------------------------------------------------------
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int base = (c * DC + DC <= D) ? c * DC : D - DC;
            const int skip = c * DC - base;
            Chunk nxt;
#pragma unroll
            for (int e = 0; e < DC; ++e) {
                if (e < skip) continue;
                const int l = base + e;
                T v;
                if constexpr (EXPD) {
                    v = cur.y[e];
                } else if constexpr (ISO) {
                    if constexpr (KEEP_R) v = r[l];
                    else v = x[l] - cur.y[e];
                } else {
                    v = cur.y[e];
                }
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) b[rr][l] = cg_fma(c2[rr], v, cg_fma(k1, cur.a[rr][e], b[rr][l]));
                if (e == skip) {
                    __builtin_amdgcn_sched_barrier(0);
                    nxt = (c + 1 < NC) ? load_a(q, c + 1) : load_ya(p + RS, 0);   // last chunk: next column (stream is padded)
                    if constexpr (VG) {
                        if (c + 1 == NC) {
#pragma unroll
                            for (int rr = 0; rr < NR; ++rr) a0n[rr] = a0p[NR * (jj + 1) + rr];
                        }
                    }
                    if constexpr (EXPD) {
                        if (c + 1 == NC) {
                            enyn = exq[(1 + NR) * (jj + 1)];
#pragma unroll
                            for (int rr = 0; rr < NR; ++rr) eyan[rr] = exq[(1 + NR) * (jj + 1) + 1 + rr];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
        if constexpr (VG) {
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) a0[rr] = a0n[rr];
        }
        if constexpr (EXPD) {
            eny = enyn;
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) eya[rr] = eyan[rr];
        }
    }
    };
    if constexpr (FAM == COVGRAM_MATERNP) {
        if (kp.p == 2) columns(std::integral_constant<int, 2>());
        else columns(std::integral_constant<int, -1>());
    } else {
        columns(std::integral_constant<int, -1>());
    }
    if constexpr (EXPD) {
#pragma unroll
        for (int rr = 0; rr < NR; ++rr)
#pragma unroll
            for (int l = 0; l < D; ++l) b[rr][l] = cg_fma(x[l], csum[rr], b[rr][l]);
    }

    if (!live) return;
    constexpr int VGI = VG ? 1 : 0;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        if (final_store) {
            T* yp = out + rr * ldy + row * (int64_t)(d + VGI);
            if constexpr (VG) {
                T v = alpha0 * b0[rr];
                if (beta != (T)0) v = cg_fma(beta, yp[0], v);
                yp[0] = v;
            }
#pragma unroll
            for (int l = 0; l < D; ++l) {
                if (l < d) {
                    T v = alpha * b[rr][l];
                    if (beta != (T)0) v = cg_fma(beta, yp[VGI + l], v);
                    yp[VGI + l] = v;
                }
            }
        } else {
            // partial slab [jsplit][NR][D (+1: the value row)][npad]: lane-contiguous rows -> coalesced stores
            T* op = out + ((int64_t)blockIdx.y * NR + rr) * (D + VGI) * npad + row;
#pragma unroll
            for (int l = 0; l < D; ++l) op[(int64_t)l * npad] = b[rr][l];
            if constexpr (VG) op[(int64_t)D * npad] = b0[rr];
        }
    }
}

// y[i*d + l] = alpha * sum_s partial[s][l][i] + beta * y ; one thread per (i, l) with i fastest: coalesced slab reads.
// vg = 1: blocks of d+1 — output entry 0 is the value row (slab row D, scaled by alpha0), entry 1+l the gradient row l.
// blockIdx.z = right-hand side (slabs [split][nr][D (+1)][npad], outputs ldy apart).
template <typename T>
__global__ __launch_bounds__(256) void grad_reduce_kernel(const T* __restrict__ partial, int64_t npad, int32_t D, int32_t jsplit,
                                                          T* __restrict__ y, int64_t n, int32_t d, T alpha, T beta, int32_t vg,
                                                          T alpha0, int64_t ldy = 0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;                                  // output entry within the block
    const int rr = blockIdx.z, nr = gridDim.z;
    if (i >= n || e >= d + vg) return;
    const int l = vg ? (e == 0 ? D : e - 1) : e;               // slab row
    if (vg && e == 0) alpha = alpha0;
    T s = (T)0;
    for (int sp = 0; sp < jsplit; ++sp) s += partial[(((int64_t)sp * nr + rr) * (D + vg) + l) * npad + i];
    T* yp = y + rr * ldy + i * (int64_t)(d + vg) + e;
    T v = alpha * s;
    if (beta != (T)0) v = cg_fma(beta, *yp, v);
    *yp = v;
}

// P[j][0..D) = gamma * Y[j][0..d), P[j][D..2D) = a[j*d + 0..d)   (zero padded); record m (one past the end) is zeroed:
// the kernel's software pipeline prefetches it and never consumes it.
// vg = 1: A holds blocks of d+1 (value weight first); the value weights go to A0[0..m] (A0[m] = 0, prefetch only).
// nr right-hand sides lda apart: records of (1 + nr) D scalars, nr value weights per column.
template <typename T>
__global__ __launch_bounds__(256) void grad_pack_kernel(const T* __restrict__ Y, int64_t m, int32_t d, const T* __restrict__ A,
                                                        T* __restrict__ P, int32_t D, T gamma, int32_t vg, T* __restrict__ A0,
                                                        const T* __restrict__ Cn, int32_t nr = 1, int64_t lda = 0) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (m + 1) * (int64_t)D) return;
    const int64_t j = e / D;
    const int l = (int)(e - j * D);
    T* p = P + j * (int64_t)((1 + nr) * D);
    const bool real = (j < m) && (l < d);
    p[l] = real ? (Y[j * (int64_t)d + l] - (Cn ? Cn[l] : (T)0)) * gamma : (T)0;
    for (int rr = 0; rr < nr; ++rr) {
        p[(1 + rr) * D + l] = real ? A[rr * lda + j * (int64_t)(d + vg) + vg + l] : (T)0;
        if (vg && l == 0) A0[nr * j + rr] = (j < m) ? A[rr * lda + j * (int64_t)(d + 1)] : (T)0;
    }
}

// EXPD: Ex[2 j] = |gamma (y_j - c)|^2, Ex[2 j + 1] = gamma (y_j - c) . a_j for j < m; entry m (prefetch only) is zero
template <typename T>
__global__ __launch_bounds__(256) void grad_pack_extra_kernel(const T* __restrict__ Y, int64_t m, int32_t d, const T* __restrict__ A, T gamma, int32_t vg,
                                                              const T* __restrict__ Cn, T* __restrict__ Ex, int32_t nr = 1, int64_t lda = 0) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > m) return;
    T ny = (T)0, ya[2] = {(T)0, (T)0};
    if (j < m)
        for (int l = 0; l < d; ++l) {
            const T yl = (Y[j * (int64_t)d + l] - (Cn ? Cn[l] : (T)0)) * gamma;
            ny = cg_fma(yl, yl, ny);
            for (int rr = 0; rr < nr; ++rr) ya[rr] = cg_fma(yl, A[rr * lda + j * (int64_t)(d + vg) + vg + l], ya[rr]);
        }
    Ex[(1 + nr) * j] = ny;
    for (int rr = 0; rr < nr; ++rr) Ex[(1 + nr) * j + 1 + rr] = ya[rr];
}

// The same for d = DL a power of two (8 .. 64): DL lanes share a column, one coordinate each — a wave's load is consecutive memory (round 5).  With a
// thread per column every lane of a load sits on its own cache line (rows of 256 bytes at d = 32 fp64): 16.7 us for 8 MB at the C4 shape.
template <typename T, int DL>
__global__ __launch_bounds__(256) void grad_pack_extra_lanes_kernel(const T* __restrict__ Y, int64_t m, const T* __restrict__ A, T gamma, int32_t vg,
                                                                    const T* __restrict__ Cn, T* __restrict__ Ex, int32_t nr, int64_t lda) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j = g / DL;
    const int l = (int)(g % DL);
    if (j > m) return;                                           // (whole lane groups: DL divides the block)
    T ny = (T)0, ya[2] = {(T)0, (T)0};
    if (j < m) {
        const T yl = (Y[j * (int64_t)DL + l] - (Cn ? Cn[l] : (T)0)) * gamma;
        ny = yl * yl;
        for (int rr = 0; rr < nr; ++rr) ya[rr] = yl * A[rr * lda + j * (int64_t)(DL + vg) + vg + l];
    }
#pragma unroll
    for (int o = DL / 2; o > 0; o >>= 1) { ny += __shfl_xor(ny, o); ya[0] += __shfl_xor(ya[0], o); ya[1] += __shfl_xor(ya[1], o); }
    if (l == 0) {
        Ex[(1 + nr) * j] = ny;
        for (int rr = 0; rr < nr; ++rr) Ex[(1 + nr) * j + 1 + rr] = ya[rr];
    }
}

template <int FAM, int D> static bool launch_grad_bcast(const GradArgs& a);   // grad_bcast.hpp

template <typename T, int FAM, int D>
static int launch_grad_one(const GradArgs& a) {
    if constexpr (sizeof(T) == 8 && fam_is_iso<FAM> && !fam_is_expr<FAM>) {
        if (a.bcast && a.expd && a.nr == 1 && a.hk->k.power == 1) {
            if (launch_grad_bcast<FAM, D>(a)) {
                hipError_t e = hipGetLastError();
                if (e != hipSuccess) { set_error("grad_bcast launch failed: %s", hipGetErrorString(e)); return COVGRAM_EHIP; }
                return COVGRAM_OK;
            }
        }
    }
    const typename ParamsOf<FAM, T>::type kp = make_params<FAM, T>(*a.hk);
    const int threads = grad_block_threads((int)sizeof(T), D, FAM);
    dim3 grid((unsigned)((a.n + threads - 1) / threads), (unsigned)a.jsplit);
    const int final_store = (a.jsplit == 1) ? 1 : 0;
    // keep r = x - y in registers (4 flops per dim and block) while 3 d-vectors of state leave >= 2 waves per SIMD,
    // else recompute it in the second sweep (5 flops per dim, 2 d-vectors of state)
    constexpr int W3 = 3 * D * (int)(sizeof(T) / 4);
    constexpr bool CAN_KEEP = (W3 <= 200);
    // measured (profiles/r01_gradbench_sweep_v3.txt): keeping r wins only while it costs no occupancy that matters —
    // fp32 d <= 32 and fp64 d <= 16; beyond that the recomputing variant's extra wave per SIMD is worth more than a flop per dim
    bool keep = CAN_KEEP && (W3 <= 100);
    // composites (per-pair interpreter): keeping r pays only in fp32 at small d (tools/compgrad_ab.py: fp64 d = 8 7.35 ms with r kept,
    // 4.08 without; fp32 d = 32 2.06 / 1.69, d = 8 1.07 / 1.17)
    if constexpr (fam_is_expr<FAM>) keep = keep && sizeof(T) == 4 && D <= 16;
    if (a.keep_r == 0) keep = false;
    if (a.keep_r == 1) keep = CAN_KEEP;
    const bool pow = !fam_is_expr<FAM> && a.hk->k.power != 1;
    constexpr bool POWT = !fam_is_expr<FAM>;   // composites apply Power per factor: no POW = true instantiation
#define CG_GRAD_LAUNCH_N(KEEPV, POWV, VGV, EXPV, NRV)                                                                                   \
    hipLaunchKernelGGL((grad_mvm_kernel<T, FAM, D, KEEPV, POWV, VGV, EXPV, NRV>), grid, dim3(threads), 0, a.stream, (const T*)a.X, a.n, a.d, \
                       (const T*)a.P, (const T*)a.P, a.m, (T*)a.out, a.npad, a.jchunk, (T)a.alpha, (T)a.beta, final_store,             \
                       (const T*)a.A0, (T)a.alpha0, (T)a.vg_c, (T)a.vg_b, (const T*)a.C, kp, (const T*)((EXPV) ? a.Ex : nullptr), a.ldy)
#define CG_GRAD_LAUNCH(KEEPV, POWV, VGV) CG_GRAD_LAUNCH_N(KEEPV, POWV, VGV, false, 1)
    bool done = false;
    // two right-hand sides per pass: r, s, phi', phi'' once per pair, two accumulators (the caller checked grad_two_rhs_ok: no Power
    // wrapper, 3 d-vectors of state within the registers); r is recomputed in the second sweep
    if constexpr (grad_two_rhs_ok(sizeof(T), D, FAM)) {
        if (a.nr == 2) {
            bool x2 = false;
            if constexpr (fam_is_iso<FAM> && !fam_is_expr<FAM>) {
                if (a.expd) { if (a.vg) CG_GRAD_LAUNCH_N(false, false, true, true, 2); else CG_GRAD_LAUNCH_N(false, false, false, true, 2); x2 = true; }
            }
            if (!x2) { if (a.vg) CG_GRAD_LAUNCH_N(false, false, true, false, 2); else CG_GRAD_LAUNCH_N(false, false, false, false, 2); }
            done = true;
        }
    }
    if (!done && a.nr != 1) { set_error("grad_mvm: %d right-hand sides per launch are not compiled for this shape", a.nr); return COVGRAM_EUNSUPPORTED; }
    if constexpr (fam_is_iso<FAM> && !fam_is_expr<FAM>) {
        if (!done && a.expd) {   // expanded form: 4 fma per dimension and pair (header); fp32: without a Power wrapper (the host's rule)
            if constexpr (sizeof(T) == 8) {
                if (a.vg) { if (pow) CG_GRAD_LAUNCH_N(false, POWT, true, true, 1); else CG_GRAD_LAUNCH_N(false, false, true, true, 1); }
                else { if (pow) CG_GRAD_LAUNCH_N(false, POWT, false, true, 1); else CG_GRAD_LAUNCH_N(false, false, false, true, 1); }
            } else {
                if (a.vg) CG_GRAD_LAUNCH_N(false, false, true, true, 1); else CG_GRAD_LAUNCH_N(false, false, false, true, 1);
            }
            done = true;
        }
    }
    if (!done && a.vg) {   // the value-gradient variant always recomputes r (one instantiation per D)
        if (pow) CG_GRAD_LAUNCH(false, POWT, true); else CG_GRAD_LAUNCH(false, false, true);
        done = true;
    }
    if constexpr (CAN_KEEP) {
        if (!done && keep) {
            if (pow) CG_GRAD_LAUNCH(true, POWT, false); else CG_GRAD_LAUNCH(true, false, false);
            done = true;
        }
    }
    if (!done) { if (pow) CG_GRAD_LAUNCH(false, POWT, false); else CG_GRAD_LAUNCH(false, false, false); }
#undef CG_GRAD_LAUNCH
#undef CG_GRAD_LAUNCH_N
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("grad_mvm launch failed: %s", hipGetErrorString(e)); return COVGRAM_EHIP; }
    return COVGRAM_OK;
}

template <typename T, int FAM>
static int launch_grad_T(const GradArgs& a) {
    constexpr int MAXD = (sizeof(T) == 8) ? 48 : 64;   // 2 d-vectors of state must fit in VGPRs
    if (a.Dpad > MAXD) {
        set_error("grad_mvm: padded dimension %d exceeds the lane-per-row limit %d for this dtype", a.Dpad, MAXD);
        return COVGRAM_EUNSUPPORTED;
    }
    switch (a.Dpad) {
        case 1: return launch_grad_one<T, FAM, 1>(a);
        case 2: return launch_grad_one<T, FAM, 2>(a);
        case 3: return launch_grad_one<T, FAM, 3>(a);
        case 4: return launch_grad_one<T, FAM, 4>(a);
        case 6: return launch_grad_one<T, FAM, 6>(a);
        case 8: return launch_grad_one<T, FAM, 8>(a);
        case 12: return launch_grad_one<T, FAM, 12>(a);
        case 16: return launch_grad_one<T, FAM, 16>(a);
        case 24: return launch_grad_one<T, FAM, 24>(a);
        case 32: return launch_grad_one<T, FAM, 32>(a);
        case 48: return launch_grad_one<T, FAM, 48>(a);
        case 64:
            if constexpr (sizeof(T) == 4) return launch_grad_one<T, FAM, 64>(a);
        default: set_error("grad_mvm: padded dimension %d not compiled", a.Dpad); return COVGRAM_EUNSUPPORTED;
    }
}

template <int FAM>
int launch_grad_family(const GradArgs& a, int dtype) {
    if (dtype == COVGRAM_F32) return launch_grad_T<float, FAM>(a);
    return launch_grad_T<double, FAM>(a);
}

}  // namespace covgram
