// dense_mfma_sym2.hpp — the symmetric EQ kernel with TWO row tiles per wave (round 5).
//
// dense_mfma_sym_kernel<FAM_EQFAST*, K2> (dense_mfma.hpp) runs 8 waves x ONE row tile per 256-row panel at 4 waves per SIMD.  Its PMC pass on C3's
// symmetric partial (profiles/r05_c3_sym_pmc.txt) counts 3.0 VALU instructions per 64 evaluated entries where the arithmetic needs 2.0 (one v_exp_f32,
// two halves of a v_pk_fma_f32): the third is per-tile and per-stage bookkeeping — the column sums' 3 adds + mask + LDS store per (row tile, column
// tile), the flush's 16 LDS reads + 15 adds per column tile, weight reads, LDS addresses, masks — all of it per ROW tile.
//
// Here a panel is 4 waves x 2 row tiles: a wave keeps two sets of row fragments, accumulators and row weights, reads every column fragment and weight
// from LDS ONCE for both, adds both row tiles' column sums in registers before they leave the wave (one reduction, one mask, one LDS store per column
// tile and wave; the flush adds 8 partials instead of 16), and three 4-wave workgroups per CU give 3 waves per SIMD — where the tile body itself runs
// fastest (profiles/r05_tile_body_probe.txt: 16.5 against 17.6 cycles per 64 entries at 4).  Same panels, chunks, slabs, masks, workgroup list and
// reduce kernel as dense_mfma_sym_kernel: the two are interchangeable per launch (option "mfma_sym_rt"), and every sum is formed in the same order
// within a row tile; across the two row tiles of a wave the column sums are added in registers (row tile 2 w, then 2 w + 1) where the 8-wave kernel
// adds them in LDS in wave order — the same order.
//
// Generic form (round 5, second half): the same kernel for the isotropic single profiles (FAM = the launcher family, ORD / GFMT
// as in dense_mfma_sym_kernel) — row fragments by gen_row_fragments, the profile by mfma_profile_block on both row tiles' 16 entries, W = the per-MVM
// packed weights, no fraction factors.  These kernels are bound by their transcendentals; what two row tiles per wave halve is the per-(row tile,
// column tile) bookkeeping, which the 4-wave one-row-tile panels of MaternP / RQ paid in full.
// (included from the middle of dense_mfma.hpp, after the helpers it uses and before the launchers that name it)
#pragma once

namespace covgram {

// ST_: column tiles per stage (one barrier, one round of LDS-DMA issues, weight reads and column-sum flushes per stage): 8 where the LDS allows
// three workgroups per CU (K2 <= 2: 49 KB each), else 4
template <int FAM, int K2, int ST_ = (K2 <= 2 ? 8 : 4), int ORD = 0, int GFMT = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void dense_mfma_sym2_kernel(
    const float* __restrict__ X, int64_t n, int32_t d, const uint4* __restrict__ PB, const float* __restrict__ W, int64_t ntile,
    float* __restrict__ R, float* __restrict__ S, int64_t npad, int32_t tchunk, float g, const float* __restrict__ Cn,
    int32_t pfirst, int32_t pstride, const int32_t* __restrict__ wgmap, const typename SymParamsOf<FAM>::type kp, const float* __restrict__ EF) {
    constexpr bool FAST = (FAM == FAM_EQFAST || FAM == FAM_EQFAST_H);
    static_assert(FAST || (fam_is_iso<FAM> && !fam_is_expr<FAM>), "the two-row-tile form: the EQ kernels and the isotropic single profiles");
    constexpr int FMT = (FAM == FAM_EQFAST_H || (!FAST && GFMT == 1)) ? 1 : 0;
    constexpr int NW = 4, RT = 2, TPP = NW * RT, ST = ST_, SPW = ST / NW;   // 8 row tiles per panel; stages of ST column tiles, SPW fetched (and later flushed) by each wave
    auto wt = [&](int64_t j) { if constexpr (FAST) return j < n ? W[j] * EF[j] : 0.0f; else return W[j]; };
    const int32_t wm = wgmap[blockIdx.x];
    const int64_t lp = wm >> 12;
    const int64_t cabs = wm & 4095;
    const int64_t p = pfirst + (int64_t)pstride * lp;
    const int64_t T1a = (cabs + 1) * tchunk;
    const int64_t T0 = (cabs * tchunk > TPP * p) ? cabs * tchunk : TPP * p;
    if (T1a <= TPP * p || T0 >= ntile) return;
    const int64_t T1 = T1a < ntile ? T1a : ntile;
    const int nt = (int)(T1 - T0);
    const int l = threadIdx.x & 63, t = l & 31, h = l >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t I0 = TPP * p + RT * wv;                           // this wave's first row tile (the second is I0 + 1)
    Frag a[RT][K2];
    float er[RT];
    f32x2 u2[RT][8], acc2[RT][8];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int64_t i0 = (I0 + r) * 32;
        int64_t row = i0 + t;
        if (row >= n) row = n - 1;
        if constexpr (FAST) er[r] = eq_row_fragments_fmt<K2, FMT>(X + row * (int64_t)d, Cn, d, g, h, a[r]);
        else { gen_row_fragments<K2, GFMT, true>(X + row * (int64_t)d, Cn, d, kp.gamma, h, a[r]); er[r] = 1.0f; }
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            float uv[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int vv = 2 * v + e;
                int64_t ri = i0 + 8 * (vv >> 2) + 4 * h + (vv & 3);
                const float keep = ri < n ? 1.0f : 0.0f;
                if (ri >= n) ri = n - 1;
                uv[e] = wt(ri) * keep;
            }
            u2[r][v] = (f32x2){uv[0], uv[1]};
            acc2[r][v] = (f32x2){0.0f, 0.0f};
        }
    }

    const uint4* __restrict__ pbase = PB + (T0 * K2) * 64;
    __shared__ uint4 sfA[ST][K2][64], sfB[ST][K2][64];
    __shared__ float swA[ST][32], swB[ST][32];
    __shared__ float csA[NW][ST][64], csB[NW][ST][64];
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int nstage = (nt + ST - 1) / ST;
    float gw[SPW];
    // one column tile against both row tiles: E once per row tile, row sums with the column weight, column sums of both row tiles in one pair of
    // register accumulators.  MASKED (stages that touch the panel's diagonal block): row tile I takes row sums from tiles J >= I, column sums from J > I.
    auto process = [&](auto masked, const Frag (&f)[K2], float w, int64_t J, float& cpart) {
        constexpr bool MASKED = decltype(masked)::value;
        f32x16 D[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            D[r] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int mm = 0; mm < K2; ++mm) D[r] = eq_mma<FMT>(a[r][mm], f[mm], D[r]);
        }
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            if constexpr (FAST) {
#pragma unroll
                for (int v = 0; v < 16; ++v) D[r][v] = __builtin_amdgcn_exp2f(D[r][v]);
            } else mfma_profile_block<FAM, ORD>(D[r], kp);
        }
        f32x2 c01 = {0.0f, 0.0f}, c23 = {0.0f, 0.0f};
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const float wr = (!MASKED || J >= I0 + r) ? w : 0.0f;
            f32x2 d01 = {0.0f, 0.0f}, d23 = {0.0f, 0.0f};
#pragma unroll
            for (int v = 0; v < 8; v += 2) {
                const f32x2 e01 = {D[r][2 * v], D[r][2 * v + 1]}, e23 = {D[r][2 * v + 2], D[r][2 * v + 3]};
                acc2[r][v] = pk_fma((f32x2){wr, wr}, e01, acc2[r][v]);
                acc2[r][v + 1] = pk_fma((f32x2){wr, wr}, e23, acc2[r][v + 1]);
                if constexpr (MASKED) { d01 = pk_fma(u2[r][v], e01, d01); d23 = pk_fma(u2[r][v + 1], e23, d23); }
                else { c01 = pk_fma(u2[r][v], e01, c01); c23 = pk_fma(u2[r][v + 1], e23, c23); }
            }
            if constexpr (MASKED) { if (J > I0 + r) { c01 = c01 + d01; c23 = c23 + d23; } }
        }
        cpart = (c01[0] + c01[1]) + (c23[0] + c23[1]);
    };
#define CG2_DMA(stage, SF)                                                                      \
        _Pragma("unroll") for (int q_ = 0; q_ < SPW; ++q_) {                                    \
            const int ti_ = (stage) * ST + wv + NW * q_;                                        \
            const int tc_ = ti_ < nt ? ti_ : nt - 1;                                            \
            _Pragma("unroll") for (int mm = 0; mm < K2; ++mm)                                   \
                __builtin_amdgcn_global_load_lds((gptr_t)(pbase + (tc_ * K2 + mm) * 64 + l), (lptr_t)&SF[wv + NW * q_][mm][0], 16, 0, 0); \
            gw[q_] = wt((T0 + tc_) * 32 + t) * (ti_ < nt ? 1.0f : 0.0f);                        \
        }
#define CG2_PUTW(SW) if (h == 0) { _Pragma("unroll") for (int q_ = 0; q_ < SPW; ++q_) SW[wv + NW * q_][t] = gw[q_]; }
    /* one column tile per iteration: the two row tiles already give the wave two independent MFMA chains and 32 exponentials to overlap, and a second \
       column tile's fragments and results in flight spilled (48-100 B at K2 <= 2, 500 B at K2 = 4) */ \
#define CG2_STAGE_M(M_, st_, SF, SW, CS)                                                        \
        _Pragma("unroll 1") for (int k = 0; k < ST; ++k) {                                      \
            Frag f0[K2];                                                                        \
            _Pragma("unroll") for (int mm = 0; mm < K2; ++mm) f0[mm].u = SF[k][mm][l];          \
            const float w0 = SW[k][t];                                                          \
            float cp0;                                                                          \
            process(std::integral_constant<bool, M_>(), f0, w0, T0 + (int64_t)(st_) * ST + k, cp0);     \
            CS[wv][k][l] = cp0;                                                                 \
        }
#define CG2_STAGE(st_, SF, SW, CS)                                                              \
        if (T0 + (int64_t)(st_) * ST >= TPP * p + TPP) { CG2_STAGE_M(false, st_, SF, SW, CS) }  \
        else { CG2_STAGE_M(true, st_, SF, SW, CS) }
    // after the stage's barrier: wave w adds the 4 waves x 2 half-waves' column sums of tile w of that stage (fixed order)
#define CG2_FLUSH(st_, CS)                                                                      \
        _Pragma("unroll") for (int q_ = 0; q_ < SPW; ++q_) {                                    \
            const int64_t J_ = T0 + (int64_t)(st_) * ST + wv + NW * q_;                         \
            if (h == 0 && J_ < T1) {                                                            \
                float s_ = 0.0f;                                                                \
                _Pragma("unroll") for (int w_ = 0; w_ < NW; ++w_) s_ += CS[w_][wv + NW * q_][t] + CS[w_][wv + NW * q_][32 + t]; \
                S[lp * npad + 32 * J_ + t] = s_;                                                \
            }                                                                                   \
        }
    CG2_DMA(0, sfA)
    CG2_PUTW(swA)
    __syncthreads();
    for (int st = 0; st < nstage; st += 2) {
        CG2_DMA(st + 1 < nstage ? st + 1 : st, sfB)
        if (st > 0) CG2_FLUSH(st - 1, csB)
        CG2_STAGE(st, sfA, swA, csA)
        CG2_PUTW(swB)
        __syncthreads();
        if (st + 1 >= nstage) { CG2_FLUSH(st, csA) break; }
        CG2_DMA(st + 2 < nstage ? st + 2 : st + 1, sfA)
        CG2_FLUSH(st, csA)
        CG2_STAGE(st + 1, sfB, swB, csB)
        CG2_PUTW(swA)
        __syncthreads();
        if (st + 2 >= nstage) { CG2_FLUSH(st + 1, csB) }
    }
#undef CG2_DMA
#undef CG2_PUTW
#undef CG2_STAGE
#undef CG2_STAGE_M
#undef CG2_FLUSH

    const int vsel = (t & 3) + 4 * (t >> 3);
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        float tot = 0.0f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            float s = acc2[r][v >> 1][v & 1];
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8); s += __shfl_xor(s, 16);
            tot = (vsel == v) ? s : tot;
        }
        const int64_t i = (I0 + r) * 32 + t;
        if (((t >> 2) & 1) != h || i >= n) continue;
        R[cabs * npad + i] = FAST ? er[r] * tot : tot;
    }
}

}  // namespace covgram
