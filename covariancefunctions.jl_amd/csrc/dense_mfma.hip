// dense_mfma.hip — the fp32 ExponentiatedQuadratic Gramian MVM on the matrix cores.
//
// The lane-per-row kernel of dense_mvm.hpp spends 23 issue cycles per 64 pairs, two thirds of them on the distance
// (3 d packed VALU ops per column pair): it sits on the VALU issue ceiling.  For the EQ profile the
// distance can leave the VALU altogether:
//     exp(-|x-y|^2 / (2 l^2)) = exp2(-|x~|^2/2) * exp2(x~ . y~) * exp2(-|y~|^2/2),      x~ = sqrt(log2 e) x / l
// so that   b_i = alpha * e_i * sum_j w_j exp2(x~_i . y~_j),   e_i = exp2(-|x~_i|^2/2),  w_j = a_j exp2(-|y~_j|^2/2):
// a 32 x 32 GEMM tile per wave and 32 rows, then ONE v_exp_f32 and ONE v_fma_f32 per pair.
//
// Which MFMA.  v_mfma_f32_32x32x2_f32 is exact but shares the fp32 FMA datapath with the VALU: measured on C2 it added
// its full 64 cycles per instruction to the VALU time (2.66 ms, no gain; profiles/r01_mfma_eq_variants.txt).  The bf16
// matrix pipe does run beside the VALU (it holds vector issue for 8 of its 32 cycles), so the fp32 dot product is
// rebuilt on it from a three-way bf16 split of every coordinate, x~ = x1 + x2 + x3 (8 + 8 + 8 mantissa bits, bf16 keeps
// the fp32 exponent range): the eight products x1y1, x1y2, x2y1, x1y3, x2y2, x3y1, x2y3, x3y2 (everything down to
// 2^-24 |x~||y~|; only x3y3 ~ 2^-32 is dropped) are exact in the fp32 accumulator, and they fill exactly the 8 K-slots a
// lane owns in v_mfma_f32_32x32x16_bf16: one MFMA covers two coordinates (lane half h takes coordinate 2 mm + h).
//
// Which split (round 4).  Inside g^2 R^2 <= MFMA_F16_GATE (72) the coordinates are split into TWO fp16 pieces instead, x~ = h1 + h2, and only the
// three products h1 h1, h1 h2, h2 h1 are formed (v_mfma_f32_32x32x16_f16; a lane's 8 K-slots hold two coordinates + the norm slots, so one MFMA
// covers FOUR coordinates): what is dropped is ~2^-22 |x~_c y~_c| per coordinate — one fp32 rounding of the dot product, which the accumulator
// commits anyway — and the matrix-core work per pair halves.  The d = 5 ... 8 kernels were power-bound (2.0 GHz under them), so that is fewer cycles
// AND a higher clock: mvm_eq_mfma below ("Which split") has the measurements and the gate; everything after the MFMA is the same for both splits.
//
// Tile orientation: D = X~tile (32 rows i) * Y~tile^T (32 columns j).  A lane holds column j = lane & 31 and the 16 rows
// i = (v & 3) + 8 (v >> 2) + 4 (lane >> 5) of the result, so the column weight w_j is ONE per-lane register per tile and
// the 16 accumulators of a lane collect "row i, columns == lane (mod 32)" over all column tiles; the sum over the 32
// lanes happens once per wave, at the end (5 butterfly steps).  RT row tiles per wave share every B fragment.
//
// Numerics.  This is NOT the reference's direct-difference r^2 (src/util.jl:40-47): the exponent
//     x~ . y~ - |x~|^2/2 - |y~|^2/2  =  -|x~ - y~|^2 / 2
// is formed in the MFMA's fp32 accumulator from the split products plus the INTEGER parts of the two half-norms (two extra
// K-slots); their fractions are fp32 factors in (1/2, 1] (dense_mfma.hpp: norm_split) — so every exponential is <= 4: it can
// neither overflow nor lose a row or a column to an underflowing norm factor (round 1 kept e_i = exp2(-|x~_i|^2/2) and
// a_j e_j whole, which flushed every row of an X cluster > ~16 scaled units from the column centre to 0), and the weights
// stay within a factor 2 of the caller's a_j.  What remains is the cancellation: the partial sums reach max(|x~|^2, |y~|^2),
// so the exponent carries an absolute error of a few fp32 roundings of that size.  The path is therefore taken only when BOTH
// point sets lie within
//     g^2 max_i |x_i - c|^2 <= MFMA_GATE  and  g^2 max_j |y_j - c|^2 <= MFMA_GATE      (c = the column side's centre)
// (= 126: measured contribution to the MVM's relative error ~4e-9 per unit of the bound, i.e. <= 5e-7 against the 1e-5
// fp32 tolerance of BASELINE.json; common.hpp) — the radii are computed once when the covgram_points handle is created —
// and otherwise the exact direct-difference kernel runs.
// Option "dense_variant": 0 = this rule, 1 = always direct differences, 2 = MFMA whenever the shape allows (tests).
//
// Kernels in this file and in dense_mfma.hpp (one path each, chosen on the host in covgram_mvm):
//   dense_mfma_eq_kernel<K2, RT, WPB, LDS>   EQ, any X / Y.  LDS = 0: one wave per workgroup with its own fragment loads;
//                                            LDS = 1 (d <= 8) / 2 (d > 8): four waves share the column tiles through LDS
//                                            (global_load_lds DMA, double buffered) when the column chunks are long.
//   dense_mfma_gen_kernel<FAM, ...>          RQ, Cauchy, IMQ, MaternP, Dot^p, ExponentialDot, EQ^p, several right-hand sides:
//                                            the MFMA yields the profile argument s (dense_mfma.hpp).
//   dense_mfma_sym_kernel / _sym_wide_kernel gramian(k, x), one point set: the upper triangle once, row AND column sums
//                                            ("Symmetric Gramian" below); also rank r of P's share for the multi-GPU form.
#include "dense_mfma.hpp"
#include "dense_mfma_sym2.hpp"

namespace covgram {

// B fragments of every column tile in MFMA lane order + the fraction factors of the column norms (both depend on the points
// and g only: cached in the points handle).  Lane (r, h) of MFMA mm of tile T holds, for coordinate c = 2 mm + h of column
// j = 32 T + r, the eight K-slots  [y1, y2, y1, y3, y2, y1, y3, y2]  (16 bytes) of y~_c, which meet the row side's
// [x1, x1, x2, x1, x2, x3, x2, x3]; the integer part k_j of -|y~_j|^2/2 sits in the idle coordinate c = d as [1, k_j, 0, ...]
// (d odd) or replaces the last two slots of coordinate 0, [.., 1, k_j] (d even) — dense_mfma.hpp: norm_split.
//   PB[(T * K2 + mm) * 64 + l] = that fragment (zero outside the point set / dimension)
//   EF[32 T + r]               = exp2(f_j) in (1/2, 1]                          (0 for padding columns)
__global__ __launch_bounds__(256) void mfma_pack_kernel(const float* __restrict__ Y, int64_t m, int32_t d, uint4* __restrict__ PB,
                                                        float* __restrict__ EF, int32_t K2, float g, const float* __restrict__ Cn, int32_t fmt) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (tile, mm, lane)
    const int64_t ntile = (m + 31) / 32;
    if (e >= ntile * K2 * 64) return;
    const int l = (int)(e & 63);
    const int64_t q = e >> 6;
    const int mm = (int)(q % K2);
    const int64_t T = q / K2;
    const int64_t j = 32 * T + (l & 31);
    if (fmt == 1) {
        // fp16 two-way split: lane (r, h) of MFMA mm holds coordinates c0 = 4 mm + 2 h and c0 + 1 of column j as
        // [y1, y2, y1 | y1, y2, y1 | 1, k_j] — the last two slots in lane half 0 of MFMA 0 only (dense_mfma.hpp: eq_row_fragments_h)
        const int c0 = 4 * mm + 2 * (l >> 5);
        const bool nl = mm == 0 && (l >> 5) == 0;
        uint4 frag = make_uint4(0, 0, 0, 0);
        if (j < m) {
            const float ya = c0 < d ? g * (Y[j * (int64_t)d + c0] - Cn[c0]) : 0.0f;
            const float yb = c0 + 1 < d ? g * (Y[j * (int64_t)d + c0 + 1] - Cn[c0 + 1]) : 0.0f;
            unsigned a1, a2, b1, b2;
            split2h(ya, a1, a2);
            split2h(yb, b1, b2);
            frag = make_uint4(a1 | (a2 << 16), a1 | (b1 << 16), b2 | (b1 << 16), 0u);
            if (nl) {
                double ny = 0.0;
                for (int cc = 0; cc < d; ++cc) { const double yc = (double)(g * (Y[j * (int64_t)d + cc] - Cn[cc])); ny = __builtin_fma(yc, yc, ny); }
                const double hn = -0.5 * ny, k = __builtin_ceil(hn);
                frag.w = F16_ONE | (f16_bits((float)k) << 16);
                EF[j] = __builtin_amdgcn_exp2f((float)(hn - k));
            }
        } else if (nl) {
            EF[j] = 0.0f;
        }
        PB[e] = frag;
        return;
    }
    const int c = 2 * mm + (l >> 5);
    uint4 frag = make_uint4(0, 0, 0, 0);
    const bool norm_lane = (d & 1) ? (c == d) : (c == 0);          // the lane that carries k_j (one per column)
    if (j < m) {
        if (c < d) {
            const float yt = g * (Y[j * (int64_t)d + c] - Cn[c]);
            unsigned y1, y2, y3;
            split3(yt, y1, y2, y3);
            frag = make_uint4(y1 | (y2 << 16), y1 | (y3 << 16), y2 | (y1 << 16), y3 | (y2 << 16));
        }
        if (norm_lane) {
            double ny = 0.0;
            for (int cc = 0; cc < d; ++cc) { const double yc = (double)(g * (Y[j * (int64_t)d + cc] - Cn[cc])); ny = __builtin_fma(yc, yc, ny); }
            const NormSplit sp = norm_split(ny);
            if (d & 1) frag = make_uint4(BF16_ONE | (sp.kbits << 16), 0, 0, 0);
            else frag.w = BF16_ONE | (sp.kbits << 16);
            EF[j] = sp.ef;
        }
    } else if (norm_lane) {
        EF[j] = 0.0f;
    }
    PB[e] = frag;
}

// the per-MVM part of the pack: W[j] = a_j * exp2(f_j) (the cached fraction factor, in (1/2, 1]), 0 for padding columns
__global__ __launch_bounds__(256) void mfma_pack_w_kernel(const float* __restrict__ A, const float* __restrict__ EF, float* __restrict__ W,
                                                          int64_t m, int64_t mpad) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < mpad) W[j] = j < m ? A[j] * EF[j] : 0.0f;
}

// LDS: 0 = every wave loads its own column tiles; 1 = stages of WPB tiles shared through LDS (K2 <= 4, RT = 2);
//      2 = ONE tile per stage, its K2 fragment slices fetched by the WPB waves in turn (K2 > 4, RT = 1)
// STAMP = 1: the diagnostic build of the SAME loop that reads the shader clock (s_memtime) and the constant 100 MHz counter
// (s_memrealtime) around the column loop of every workgroup — clock = d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md,
// DVFS give-back item 6).  The stamps go to a buffer of their own; no product launch ever runs this instantiation.
template <int K2, int RT, int WPB = 1, int LDS = 0, int STAMP = 0, int FMT = 0>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(LDS == 1 ? (RT == 4 ? 2 : K2 <= 2 ? 4 : 3) : 1, LDS == 1 ? (RT == 4 ? 2 : K2 <= 2 ? 4 : 3) : 8))) void dense_mfma_eq_kernel(const float* __restrict__ X, int64_t n, int32_t d,
                                                           const uint4* __restrict__ PB, const float* __restrict__ W, int64_t ntile,
                                                           float* __restrict__ out, int64_t npad, int64_t tchunk, float g,
                                                           float alpha, float beta, int32_t final_store, const float* __restrict__ Cn,
                                                           unsigned long long* __restrict__ stamps, unsigned* __restrict__ tickets,
                                                           float* __restrict__ yfinal, const float* __restrict__ EF, int64_t mcols) {
    // WPB waves per workgroup take consecutive row tiles and walk the SAME column tiles at the same pace (no barrier,
    // nothing shared explicitly): their fragment loads coalesce in the CU's vector L1 instead of each going to L2
    const int l = threadIdx.x & 63, t = l & 31, h = l >> 5;
    const int64_t i0 = ((int64_t)blockIdx.x * WPB + (threadIdx.x >> 6)) * (32 * RT);
    // A fragments: lane (t, h) holds the split of x~[row][c = 2 mm + h] (+ the integer part of the row's half-norm)
    Frag a[RT][K2];
    float er[RT];                                                // exp2 of the fraction of the row's half-norm, in (1/2, 1]
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        int64_t row = i0 + 32 * r + t;
        if (row >= n) row = n - 1;                               // clamp: computed, never stored
        er[r] = eq_row_fragments_fmt<K2, FMT>(X + row * (int64_t)d, Cn, d, g, h, a[r]);
    }

    unsigned long long st_c0 = 0, st_r0 = 0;
    if constexpr (STAMP) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    const int64_t T0 = (int64_t)blockIdx.y * tchunk;
    const int64_t T1 = (T0 + tchunk < ntile) ? (T0 + tchunk) : ntile;
    // the accumulators live as register PAIRS: with one or two MFMAs per tile the weighted sums are v_pk_fma_f32 (two fmas per instruction — nearly twice
    // the rate while no MFMA executes, slower than two v_fma_f32 beside one: tools/pkfma_probe.hip; C2-shaped 1.44 -> 1.33 ms, four MFMAs per tile +2-3 %)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc2[RT][8];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int v = 0; v < 8; ++v) acc2[r][v] = (f32x2){0.0f, 0.0f};

    // Column tiles in pairs with two operand buffers (no register copies); the prefetch of a tile past the chunk is clamped
    // to the last tile (a harmless re-read) instead of branching.  Uniform base + 32-bit lane offset -> saddr loads.
    const uint4* __restrict__ pbase = PB + (T0 * K2) * 64;
    // column weights w_j = a_j exp2(f_j) (0 for padding columns).  EF != nullptr (round 4): W is the caller's a itself and the product with the cached
    // fraction factors is formed here, where a tile's weights are fetched: no weight-pack launch in front of the MVM (mvm_eq_mfma)
    const float* __restrict__ wbase = W + T0 * 32;
    const int nt = (int)(T1 - T0);
    auto weight = [&](int tc) {
        if (EF == nullptr) return wbase[tc * 32 + t];
        const int64_t j = (T0 + tc) * 32 + t;
        return j < mcols ? W[j] * EF[j] : 0.0f;
    };
    auto load_tile = [&](int ti, Frag (&f)[K2], float& w) {
        const int tc = ti < nt ? ti : nt - 1;
#pragma unroll
        for (int mm = 0; mm < K2; ++mm) f[mm].u = pbase[(tc * K2 + mm) * 64 + l];
        w = weight(tc);
    };
    auto process = [&](const Frag (&f)[K2], float w) {
        f32x16 D[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            D[r] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int mm = 0; mm < K2; ++mm) D[r] = eq_mma<FMT>(a[r][mm], f[mm], D[r]);
        }
        // all exponentials of the tile first, then the weighted accumulation: no exp -> fma wait states in between
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int v = 0; v < 16; ++v) D[r][v] = __builtin_amdgcn_exp2f(D[r][v]);
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                if constexpr (K2 <= 2) acc2[r][v] = __builtin_elementwise_fma((f32x2){w, w}, (f32x2){D[r][2 * v], D[r][2 * v + 1]}, acc2[r][v]);
                else { acc2[r][v][0] = __builtin_fmaf(w, D[r][2 * v], acc2[r][v][0]); acc2[r][v][1] = __builtin_fmaf(w, D[r][2 * v + 1], acc2[r][v][1]); }
            }
    };
    if constexpr (LDS == 2) {
        // Long fragments (K2 KB per tile): the workgroup stages ONE tile at a time, wave w fetching the fragment slices
        // mm = w, w + WPB, ... of the NEXT tile (global_load_lds_dwordx4) while all waves stream the CURRENT tile's slices
        // from the other buffer into their MFMA chain; one barrier per tile (a tile is >= 32 K2 matrix-pipe cycles).
        static_assert(RT == 1, "split-tile staging is built for one row tile per wave");
        __shared__ uint4 tfA[K2][64], tfB[K2][64];
        __shared__ float twA[32], twB[32];
        typedef __attribute__((address_space(1))) const void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        // (gw, ge): loaded with the tile's LDS-DMA, multiplied only where the weight is stored after the tile's arithmetic (as in the LDS == 1 form
        // below; round 5: here the product sat right behind the loads — one exposed memory round trip per TILE for wave 0, and the barrier passes it on)
        float gw = 0.0f, ge = 1.0f;
#define CG_DMA2(tile, TF)                                                                       \
        {                                                                                       \
            const int ti_ = (tile);                                                             \
            const int tc_ = ti_ < nt ? ti_ : nt - 1;                                            \
            _Pragma("unroll") for (int q = 0; q < (K2 + WPB - 1) / WPB; ++q) {                  \
                const int mm = wv + q * WPB;                                                    \
                if (mm < K2) __builtin_amdgcn_global_load_lds((gptr_t)(pbase + (tc_ * K2 + mm) * 64 + l), (lptr_t)&TF[mm][0], 16, 0, 0); \
            }                                                                                   \
            if (wv == 0) {                                                                      \
                if (EF == nullptr) gw = ti_ < nt ? wbase[tc_ * 32 + t] : 0.0f;                  \
                else {                                                                          \
                    const int64_t j_ = (T0 + tc_) * 32 + t;                                     \
                    const bool in_ = ti_ < nt && j_ < mcols;                                    \
                    gw = in_ ? W[j_] : 0.0f; ge = in_ ? EF[j_] : 0.0f;                          \
                }                                                                               \
            }                                                                                   \
        }
#define CG_TILE2(TF, TW)                                                                        \
        {                                                                                       \
            f32x16 D = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                        \
            Frag f_[K2];                                                                        \
            _Pragma("unroll") for (int mm = 0; mm < K2; ++mm) f_[mm].u = TF[mm][l];             \
            const float w = TW[t];                                                              \
            _Pragma("unroll") for (int mm = 0; mm < K2; ++mm) D = eq_mma<FMT>(a[0][mm], f_[mm], D); \
            _Pragma("unroll") for (int v = 0; v < 16; ++v) D[v] = __builtin_amdgcn_exp2f(D[v]); \
            _Pragma("unroll") for (int v = 0; v < 8; ++v) { acc2[0][v][0] = __builtin_fmaf(w, D[2 * v], acc2[0][v][0]); acc2[0][v][1] = __builtin_fmaf(w, D[2 * v + 1], acc2[0][v][1]); } \
        }
        CG_DMA2(0, tfA)
        if (wv == 0 && h == 0) twA[t] = gw * ge;
        __syncthreads();
        for (int ti = 0; ti < nt; ti += 2) {
            CG_DMA2(ti + 1, tfB)                                    // past the chunk: a re-fetch nobody reads
            CG_TILE2(tfA, twA)
            if (wv == 0 && h == 0) twB[t] = gw * ge;
            __syncthreads();
            if (ti + 1 >= nt) break;
            CG_DMA2(ti + 2, tfA)
            CG_TILE2(tfB, twB)
            if (wv == 0 && h == 0) twA[t] = gw * ge;
            __syncthreads();
        }
#undef CG_DMA2
#undef CG_TILE2
    } else if constexpr (LDS == 1) {
        // The WPB waves of the workgroup walk the same column tiles: each stage of WPB tiles is fetched ONCE — wave w moves
        // tile w of the NEXT stage straight from global memory into LDS (global_load_lds_dwordx4: no staging registers; lane
        // order in LDS = lane order of the packed fragments) while the current stage is read by all waves from the other
        // buffer — which cuts the L2 -> L1 fragment traffic by WPB.  Two buffers as two distinct arrays (the compiler's
        // LDS-DMA wait tracking tells them apart), the stage loop unrolled by two, one barrier per stage.
        __shared__ uint4 sfA[WPB][K2][64], sfB[WPB][K2][64];
        __shared__ float swA[WPB][32], swB[WPB][32];
        typedef __attribute__((address_space(1))) const void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int nstage = (nt + WPB - 1) / WPB;
        // (gw, ge): the weight's two factors a_j and exp2(f_j) when the kernel forms the product itself (EF != nullptr) — loaded here, multiplied
        // only where the weight goes to LDS at the end of the stage: a product right behind the loads would hold this wave until the LDS-DMA
        // it has just issued lands (the same counter); ge = 1 when W holds packed weights
        float gw, ge = 1.0f;
#define CG_DMA(stage, SF)                                                                       \
        {                                                                                       \
            const int ti_ = (stage) * WPB + wv;                                                 \
            const int tc_ = ti_ < nt ? ti_ : nt - 1;                                            \
            _Pragma("unroll") for (int mm = 0; mm < K2; ++mm)                                   \
                __builtin_amdgcn_global_load_lds((gptr_t)(pbase + (tc_ * K2 + mm) * 64 + l), (lptr_t)&SF[wv][mm][0], 16, 0, 0); \
            if (EF == nullptr) gw = ti_ < nt ? wbase[tc_ * 32 + t] : 0.0f;   /* tiles past the chunk: weight 0 */ \
            else {                                                                              \
                const int64_t j_ = (T0 + tc_) * 32 + t;                                         \
                const bool in_ = ti_ < nt && j_ < mcols;                                        \
                gw = in_ ? W[j_] : 0.0f; ge = in_ ? EF[j_] : 0.0f;                              \
            }                                                                                   \
        }
#define CG_STAGE(SF, SW)                                                                        \
        _Pragma("unroll 1") for (int k = 0; k < WPB; k += 2) {                                  \
            Frag f0[K2], f1[K2];                                                                \
            _Pragma("unroll") for (int mm = 0; mm < K2; ++mm) f0[mm].u = SF[k][mm][l];          \
            const float w0 = SW[k][t];                                                          \
            _Pragma("unroll") for (int mm = 0; mm < K2; ++mm) f1[mm].u = SF[k + 1][mm][l];      \
            const float w1 = SW[k + 1][t];                                                      \
            process(f0, w0);                                                                    \
            process(f1, w1);                                                                    \
        }
        CG_DMA(0, sfA)
        if (h == 0) swA[wv][t] = gw * ge;
        __syncthreads();
        for (int st = 0; st < nstage; st += 2) {
            CG_DMA(st + 1 < nstage ? st + 1 : st, sfB)              // past the last stage: a re-fetch nobody reads
            CG_STAGE(sfA, swA)
            if (h == 0) swB[wv][t] = gw * ge;
            __syncthreads();
            if (st + 1 >= nstage) break;
            CG_DMA(st + 2 < nstage ? st + 2 : st + 1, sfA)
            CG_STAGE(sfB, swB)
            if (h == 0) swA[wv][t] = gw * ge;
            __syncthreads();
        }
#undef CG_DMA
#undef CG_STAGE
    } else {
    Frag f0[K2], f1[K2];
    float w0, w1;
    load_tile(0, f0, w0);
    for (int ti = 0; ti < nt; ti += 2) {
        load_tile(ti + 1, f1, w1);
        process(f0, w0);
        load_tile(ti + 2, f0, w0);
        if (ti + 1 < nt) process(f1, w1);
    }
    }

    if constexpr (STAMP) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) {
            unsigned long long* o = stamps + 4 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
            o[0] = st_c0; o[1] = st_r0; o[2] = c1; o[3] = r1;
        }
    }
    // lane (t, h) owns output row t of each row tile iff bit 2 of t equals h; its register is v = (t & 3) + 4 (t >> 3)
    const int vsel = (t & 3) + 4 * (t >> 3);
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        // sum over the 32 column lanes of each half: afterwards every lane of half h holds the totals of rows i(v, h)
        float tot = 0.0f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            float s = acc2[r][v >> 1][v & 1];
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8); s += __shfl_xor(s, 16);
            tot = (vsel == v) ? s : tot;
        }
        const int64_t i = i0 + 32 * r + t;
        if (((t >> 2) & 1) != h || i >= n) continue;
        const float res = er[r] * tot;
        if (final_store) {
            float v = alpha * res;
            if (beta != 0.0f) v = __builtin_fmaf(beta, out[i], v);
            out[i] = v;
        } else {
            slab_store(out + (int64_t)blockIdx.y * npad + i, res, tickets != nullptr);
        }
    }
    // column split with tickets: the LAST workgroup of this row block to arrive adds the block's partials in dense_reduce_kernel's
    // order and applies alpha / beta (pack.hpp: last_arrival) — no reduce launch, no dependent-launch gap behind the kernel
    if (!final_store && tickets != nullptr) {
        if (!last_arrival(tickets + blockIdx.x, gridDim.y)) return;
        const int64_t r0 = (int64_t)blockIdx.x * (WPB * 32 * RT);
        for (int idx = threadIdx.x; idx < WPB * 32 * RT; idx += 64 * WPB) {
            const int64_t i = r0 + idx;
            if (i >= n) continue;
            float v = alpha * ordered_split_sum<float>(out + i, npad, (int)gridDim.y);
            if (beta != 0.0f) v = __builtin_fmaf(beta, yfinal[i], v);
            yfinal[i] = v;
        }
    }
}

// Column tiles for the generic kernel.  Lane (r, h) of MFMA mm of tile T holds coordinate c = 2 mm + h of column
// j = 32 T + r:  c < d: the split of (iso ? -2 g y_c : g y_c) as [y1, y2, y1, y3, y2, y1, y3, y2];  c == d (iso): the norm
// slots [1, 1, 1, n1, n2, n3, 0, 0] with n = |g y|^2;  beyond: zeros.   W[(T * NR + c) * 32 + r] = A[j + c lda].
__global__ __launch_bounds__(256) void mfma_pack_gen_kernel(const float* __restrict__ Y, int64_t m, int32_t d, const float* __restrict__ A,
                                                            int64_t lda, int32_t nrhs, int32_t c0, uint4* __restrict__ PB,
                                                            float* __restrict__ W, int32_t K2, int32_t NR, float g, int32_t iso,
                                                            const float* __restrict__ Cn, int32_t fmt) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (tile, mm, lane)
    const int64_t ntile = (m + 31) / 32;
    if (e >= ntile * K2 * 64) return;
    const int l = (int)(e & 63);
    const int64_t q = e >> 6;
    const int mm = (int)(q % K2);
    const int64_t T = q / K2;
    const int64_t j = 32 * T + (l & 31);
    const int c = 2 * mm + (l >> 5);
    uint4 frag = make_uint4(0, 0, 0, 0);
    if (fmt == 1) {
        // fp16 two-way split (dense_mfma.hpp: gen_row_fragments, GFMT = 1): positions q0 = 4 mm + 2 h and q0 + 1, three slots each — a coordinate
        // [y1, y2, y1] of -2 g (y - c), the row norm's position d as [1, 1, 1], the column norm at d + 1 as [k, f1, f2]
        if (j < m) {
            float ny = 0.0f;
            for (int cc = 0; cc < d; ++cc) { const float yc = g * (Y[j * (int64_t)d + cc] - Cn[cc]); ny = __builtin_fmaf(yc, yc, ny); }
            unsigned sl[6] = {0, 0, 0, 0, 0, 0};
            for (int w = 0; w < 2; ++w) {
                const int qq = 4 * mm + 2 * (l >> 5) + w;
                if (qq < d) { unsigned y1, y2; split2h(-2.0f * (g * (Y[j * (int64_t)d + qq] - Cn[qq])), y1, y2); sl[3 * w] = y1; sl[3 * w + 1] = y2; sl[3 * w + 2] = y1; }
                else if (qq == d) { sl[3 * w] = sl[3 * w + 1] = sl[3 * w + 2] = F16_ONE; }
                else if (qq == d + 1) { const Norm16 nn = norm16(ny); sl[3 * w] = nn.k; sl[3 * w + 1] = nn.f1; sl[3 * w + 2] = nn.f2; }
            }
            frag = make_uint4(sl[0] | (sl[1] << 16), sl[2] | (sl[3] << 16), sl[4] | (sl[5] << 16), 0u);
        }
    } else
    if (j < m) {
        if (c < d) {
            const float yt = iso ? -2.0f * (g * (Y[j * (int64_t)d + c] - Cn[c])) : g * Y[j * (int64_t)d + c];
            unsigned y1, y2, y3;
            split3(yt, y1, y2, y3);
            frag = make_uint4(y1 | (y2 << 16), y1 | (y3 << 16), y2 | (y1 << 16), y3 | (y2 << 16));
        } else if (iso && c == d) {
            float ny = 0.0f;
            for (int cc = 0; cc < d; ++cc) { const float yc = g * (Y[j * (int64_t)d + cc] - Cn[cc]); ny = __builtin_fmaf(yc, yc, ny); }
            unsigned n1, n2, n3;
            split3(ny, n1, n2, n3);
            frag = make_uint4(BF16_ONE | (BF16_ONE << 16), BF16_ONE | (n1 << 16), n2 | (n3 << 16), 0);
        }
    }
    PB[e] = frag;
    if (mm == 0 && l < 32)
        for (int cr = 0; cr < NR; ++cr) W[(T * NR + cr) * 32 + l] = (j < m && c0 + cr < nrhs) ? A[j + (int64_t)(c0 + cr) * lda] : 0.0f;
}

// A operands of dense_mfma_mrhs_kernel: AP[((T * NB + b) * 64 + l) * 16 + v] = a[j][c0 + 32 b + (l & 31)] with
// j = 32 T + 8 (v / 4) + 4 (l >> 5) + v % 4 (zero outside the matrix): lane l's sixteen values of one tile are contiguous
__global__ __launch_bounds__(256) void mfma_pack_rhs_kernel(const float* __restrict__ A, int64_t lda, int32_t nrhs, int32_t c0, int64_t m,
                                                            float* __restrict__ AP, int32_t NB) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (tile, block, lane, v)
    const int64_t ntile = (m + 31) / 32;
    if (e >= ntile * NB * 64 * 16) return;
    const int v = (int)(e & 15), l = (int)((e >> 4) & 63);
    const int64_t q = e >> 10;
    const int b = (int)(q % NB);
    const int64_t T = q / NB;
    const int64_t j = 32 * T + 8 * (v >> 2) + 4 * (l >> 5) + (v & 3);
    const int c = c0 + 32 * b + (l & 31);
    AP[e] = (j < m && c < nrhs) ? A[j + (int64_t)c * lda] : 0.0f;
}

// max_i |x_i|^2 of a point set (fp32 or fp64 points), via atomicMax on the bit pattern of a non-negative float
// DL > 1 (d == DL, a power of two 8 .. 64): DL lanes share a point, one coordinate each — consecutive lanes read consecutive memory (round 5: a thread per
// point put every lane of a load on its own cache line once rows were long: 373 us for n = 2^20, d = 32 fp64 at handle creation)
template <typename T, int DL = 1>
__global__ __launch_bounds__(256) void max_norm2_kernel(const T* __restrict__ X, int64_t n, int32_t d, unsigned* __restrict__ outbits,
                                                        const T* __restrict__ Cn) {
    // outbits[0]: max |x_i|^2 (dot-product kernels), outbits[1]: max |x_i - c|^2 (isotropic kernels, c = the handle's centre)
    float v = 0.0f, vc = 0.0f;
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gs = (int64_t)gridDim.x * blockDim.x;
    const int cl = (int)(g0 % DL);                               // (DL divides the block: a lane keeps its coordinate across the stride)
    for (int64_t i = g0 / DL; i < (n + 63) / 64 * 64; i += gs / DL) {   // grid-stride; whole waves stay in the loop for the shuffles below
        double s = 0, sc = 0;
        if constexpr (DL == 1) {
            if (i >= n) continue;
            for (int c = 0; c < d; ++c) {
                const double xc = (double)X[i * (int64_t)d + c], xd = xc - (double)Cn[c];
                s += xc * xc; sc += xd * xd;
            }
        } else {
            if (i < n) { const double xc = (double)X[i * (int64_t)DL + cl], xd = xc - (double)Cn[cl]; s = xc * xc; sc = xd * xd; }
#pragma unroll
            for (int o = DL / 2; o > 0; o >>= 1) { s += __shfl_xor(s, o); sc += __shfl_xor(sc, o); }
        }
        float f = (float)s, fc = (float)sc;
        if (!(f >= 0.0f)) f = __builtin_inff();                  // NaN / overflow: never eligible
        if (!(fc >= 0.0f)) fc = __builtin_inff();
        v = fmaxf(v, f * 1.000001f);                             // round up: the bound must not under-estimate
        vc = fmaxf(vc, fc * 1.000001f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { v = fmaxf(v, __shfl_xor(v, o)); vc = fmaxf(vc, __shfl_xor(vc, o)); }
    __shared__ float wmax[4], wcmax[4];
    if ((threadIdx.x & 63) == 0) { wmax[threadIdx.x >> 6] = v; wcmax[threadIdx.x >> 6] = vc; }
    __syncthreads();
    if (threadIdx.x == 0) {                                      // one pair of atomics per workgroup
        atomicMax(outbits, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
        atomicMax(outbits + 1, __float_as_uint(fmaxf(fmaxf(wcmax[0], wcmax[1]), fmaxf(wcmax[2], wcmax[3]))));
    }
}

int points_max_norm2(covgram_points* p) {
    p->max_norm2 = 0.0; p->max_cnorm2 = 0.0;
    p->center = nullptr; p->center_host.assign((size_t)p->d, 0.0);
    if (p->n == 0) return COVGRAM_OK;
    // The centre: the mean of up to 1024 evenly spaced points (one strided 2-D copy, averaged on the host, written to a small
    // device buffer the handle owns).  ANY common point keeps the isotropic kernels exact; one near the middle of the cloud also
    // keeps max |x - c| — and with it the matrix-core gate and the rounding of the pre-scaled coordinates — as small as the
    // cloud allows (C3: the first point sat 2.8 from the middle of a cloud of radius 6.5 and pushed the gate value to 127).
    const size_t ts = dtype_size(p->dtype);
    const int64_t ns = std::min<int64_t>(p->n, 1024), stride = p->n / ns;
    std::vector<char> samp((size_t)ns * p->d * ts);
    hipStream_t st = p->ctx->stream;
    hipError_t e = hipMemcpy2DAsync(samp.data(), (size_t)p->d * ts, p->dptr, (size_t)stride * p->d * ts, (size_t)p->d * ts, (size_t)ns,
                                    hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { set_error("centre sample copy failed: %s", hipGetErrorString(e)); return COVGRAM_EHIP; }
    std::vector<char> cdev((size_t)p->d * ts);
    for (int c = 0; c < p->d; ++c) {
        double s = 0;
        for (int64_t i = 0; i < ns; ++i)
            s += p->dtype == COVGRAM_F32 ? (double)((const float*)samp.data())[i * p->d + c] : ((const double*)samp.data())[i * p->d + c];
        s /= (double)ns;
        if (!(s - s == 0.0)) s = 0.0;                            // NaN / inf in the sample: that coordinate is not centred
        if (p->dtype == COVGRAM_F32) { ((float*)cdev.data())[c] = (float)s; p->center_host[c] = (double)(float)s; }
        else { ((double*)cdev.data())[c] = s; p->center_host[c] = s; }
    }
    e = hipMalloc(&p->center_buf, cdev.size());
    if (e != hipSuccess) { p->center_buf = nullptr; set_error("hipMalloc(%zu) failed: %s", cdev.size(), hipGetErrorString(e)); return COVGRAM_ENOMEM; }
    p->center = p->center_buf;
    p->owns_center = true;
    unsigned* dbits = nullptr;
    e = hipMemcpyAsync(p->center_buf, cdev.data(), cdev.size(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);           // cdev is a local
    if (e == hipSuccess) e = hipMalloc(&dbits, 2 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemsetAsync(dbits, 0, 2 * sizeof(unsigned), st);
    if (e == hipSuccess) {
        const unsigned grid = (unsigned)std::min<int64_t>((p->n + 255) / 256, 2048);
#define CG_MN(TT, DLV) hipLaunchKernelGGL((max_norm2_kernel<TT, DLV>), dim3((unsigned)std::min<int64_t>((p->n * DLV + 255) / 256, 512)), dim3(256), 0, st, (const TT*)p->dptr, p->n, p->d, dbits, (const TT*)p->center)
#define CG_MND(TT) do { const bool big = p->n * (int64_t)p->d >= ((int64_t)1 << 22);   /* (small sets: the thread-per-point form's 64 atomics beat 512) */ \
                        if (big && p->d == 8) CG_MN(TT, 8); else if (big && p->d == 16) CG_MN(TT, 16); else if (big && p->d == 32) CG_MN(TT, 32); else if (big && p->d == 64) CG_MN(TT, 64); \
                        else hipLaunchKernelGGL((max_norm2_kernel<TT, 1>), dim3(grid), dim3(256), 0, st, (const TT*)p->dptr, p->n, p->d, dbits, (const TT*)p->center); } while (0)
        if (p->dtype == COVGRAM_F32) CG_MND(float); else CG_MND(double);
#undef CG_MND
#undef CG_MN
        e = hipGetLastError();
    }
    unsigned bits[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(bits, dbits, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (dbits) (void)hipFree(dbits);
    if (e != hipSuccess) {
        (void)hipFree(p->center_buf); p->center_buf = nullptr; p->center = nullptr; p->owns_center = false;
        set_error("max-norm reduction failed: %s", hipGetErrorString(e)); return COVGRAM_EHIP;
    }
    float f[2];
    memcpy(f, bits, sizeof(f));
    p->max_norm2 = (double)f[0];
    p->max_cnorm2 = (double)f[1];
    return COVGRAM_OK;
}

// bound on max_i |x_i - c_Y| from the per-handle quantities: |x - c_Y| <= |x - c_X| + |c_X - c_Y|
static double centred_radius(const covgram_points* X, const covgram_points* Y) {
    double s = 0;
    for (int c = 0; c < X->d; ++c) { const double t = X->center_host[c] - Y->center_host[c]; s += t * t; }
    if (!(s >= 0)) return INFINITY;
    return sqrt(X->max_cnorm2) + sqrt(s) * 1.000001;
}

// the larger of the two squared radii about the column side's centre: every partial sum of the cancellation
// |x~|^2 + |y~|^2 - 2 x~.y~ (and of the EQ exponent) is bounded by it, so the absolute rounding error of s scales with it
double gate_radius2(const covgram_points* X, const covgram_points* Y) {
    const double rx = centred_radius(X, Y), ry = sqrt(Y->max_cnorm2);
    const double r = rx > ry ? rx : ry;
    return r * r;
}

static int eq_k2_for(int d) {   // MFMAs per tile of the EQ kernels (two coordinates each), from the compiled set
    static const int ks[] = {1, 2, 3, 4, 6, 8, 12, 16};
    const int need = (d + 1) / 2;
    for (int k : ks) if (k >= need) return k;
    return -1;
}

bool mfma_eq_eligible(const covgram_ctx* ctx, const HostKernel& hk, const covgram_points* X, const covgram_points* Y, int nrhs) {
    if (ctx->dense_variant == 1) return false;
    if (hk.tu_family != COVGRAM_EQ || hk.k.power != 1 || X->dtype != COVGRAM_F32 || nrhs != 1) return false;
    if (X->d > 32 || Y->n == 0) return false;                   // beyond d = 32 the fragments leave one wave per SIMD
    if (ctx->dense_variant == 2) return true;
    const double g2 = 1.4426950408889634074 / (hk.k.lengthscale * hk.k.lengthscale);   // |x~|^2 = g2 |x|^2
    return g2 * gate_radius2(X, Y) <= mfma_gate_of(ctx);                 // BOTH sides: a far X cluster must not ride on a compact Y
}

// resident single-wave workgroups per CU of the kernel instance (register-limited), for the grid sizing below
template <int K2, int RT>
static int mfma_blocks_per_cu() {
    static int cached = 0;
    if (!cached) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, dense_mfma_eq_kernel<K2, RT>, 64, 0) != hipSuccess || nb <= 0) nb = 16;
        cached = nb;
    }
    return cached;
}
template <int K2>
static int mfma_blocks(int rt) {
    if constexpr (K2 <= 8) { if (rt == 2) return mfma_blocks_per_cu<K2, 2>(); }
    return mfma_blocks_per_cu<K2, 1>();
}

template <int K2, int FMT = 0>
static void launch_mfma(int rt, bool lds4, dim3 grid, hipStream_t st, const float* X, int64_t n, int32_t d, const uint4* PB, const float* W,
                        int64_t ntile, float* out, int64_t npad, int64_t tchunk, float g, float alpha, float beta, int final_store, const float* Cn,
                        unsigned long long* stamps, dim3* launched, unsigned* tickets, float* yfinal, const float* EF, int64_t mcols, int64_t* inst) {
    constexpr bool NARROW = K2 <= MFMA_NARROW_MAXK2;
    unsigned long long* const ns = nullptr;
    auto enc = [](int k2, int rtt, int wpb, int lds, int stamp) { return (int64_t)k2 * 100000 + rtt * 10000 + wpb * 1000 + lds * 100 + stamp * 10 + FMT; };
    if (lds4 && K2 <= 2 && grid.x >= 1024) {   // d <= 4 and many row tiles: eight waves share each column tile (C2: 1.569 -> 1.550 ms; not for a 16384-row shard)
        *launched = dim3((grid.x + 7) / 8, grid.y);
        if constexpr (K2 <= 2) { if (stamps) { *inst = enc(K2, 2, 8, 1, 1); hipLaunchKernelGGL((dense_mfma_eq_kernel<K2, 2, 8, 1, 1, FMT>), *launched, dim3(512), 0, st, X, n, d, PB, W, ntile, out, npad, tchunk, g, alpha, beta, final_store, Cn, stamps, tickets, yfinal, EF, mcols); return; } }
        *inst = enc((K2 <= 2 ? K2 : 1), 2, 8, 1, 0); hipLaunchKernelGGL((dense_mfma_eq_kernel<(K2 <= 2 ? K2 : 1), 2, 8, 1, 0, FMT>), *launched, dim3(512), 0, st, X, n, d, PB, W, ntile, out, npad, tchunk, g, alpha, beta, final_store, Cn, ns, tickets, yfinal, EF, mcols);
    } else if (lds4) {   // 256-thread workgroups: four waves on consecutive row tiles share the column tiles through LDS
        *launched = dim3((grid.x + 3) / 4, grid.y);
        if constexpr (K2 == 3 || K2 == 4) { if (rt == 4) { *inst = enc(K2, 4, 4, 1, 0); hipLaunchKernelGGL((dense_mfma_eq_kernel<K2, 4, 4, 1, 0, FMT>), *launched, dim3(256), 0, st, X, n, d, PB, W, ntile, out, npad, tchunk, g, alpha, beta, final_store, Cn, ns, tickets, yfinal, EF, mcols); return; } }
        if constexpr (K2 == 1 || K2 == 2 || K2 == 4) { if (stamps) { *inst = enc(K2, 2, 4, 1, 1); hipLaunchKernelGGL((dense_mfma_eq_kernel<K2, 2, 4, 1, 1, FMT>), *launched, dim3(256), 0, st, X, n, d, PB, W, ntile, out, npad, tchunk, g, alpha, beta, final_store, Cn, stamps, tickets, yfinal, EF, mcols); return; } }
        constexpr int LDSF = (NARROW || K2 == 6) ? 1 : 2;   // six MFMAs per tile: stages of four tiles too (round 5; one tile per stage before, as the longer fragments still are)
        *inst = enc(K2, (NARROW ? 2 : 1), 4, LDSF, 0); hipLaunchKernelGGL((dense_mfma_eq_kernel<K2, (NARROW ? 2 : 1), 4, LDSF, 0, FMT>), *launched, dim3(256), 0, st, X, n, d, PB, W, ntile, out, npad, tchunk, g, alpha, beta, final_store, Cn, ns, tickets, yfinal, EF, mcols);
    } else if (rt == 2 && K2 <= 8) {
        *launched = grid;
        *inst = enc((K2 <= 8 ? K2 : 1), 2, 1, 0, 0); hipLaunchKernelGGL((dense_mfma_eq_kernel<(K2 <= 8 ? K2 : 1), 2, 1, 0, 0, FMT>), grid, dim3(64), 0, st, X, n, d, PB, W, ntile, out, npad, tchunk, g, alpha, beta, final_store, Cn, ns, tickets, yfinal, EF, mcols);
    } else {
        *launched = grid;
        *inst = enc(K2, 1, 1, 0, 0); hipLaunchKernelGGL((dense_mfma_eq_kernel<K2, 1, 1, 0, 0, FMT>), grid, dim3(64), 0, st, X, n, d, PB, W, ntile, out, npad, tchunk, g, alpha, beta, final_store, Cn, ns, tickets, yfinal, EF, mcols);
    }
}

// the column fragments and norm fraction factors of point set Y for (g, K2): packed once into the handle and reused by every
// later MVM (they depend on the points and the lengthscale only; the kernels form the weights a_j EF[j] themselves)
static int eq_fragments(covgram_ctx* ctx, const covgram_points* Y, int K2, float g, const float* Cn, const uint4** PB, const float** EF, int fmt = 0) {
    const int64_t m = Y->n, ntile = (m + 31) / 32;
    const size_t fbytes = (size_t)ntile * K2 * 64 * sizeof(uint4);
    const size_t total = fbytes + (size_t)ntile * 32 * sizeof(float);   // fragments + EF
    covgram_points::FragSlot* hit = nullptr;
    covgram_points::FragSlot* victim = nullptr;
    // a captured graph replays the kernels with the slot's ADDRESS baked in: a slot handed out during a capture is pinned — later eager
    // MVMs with other lengthscales evict the other slots only (the LRU re-pack silently changed what a replay reads)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = ctx->stream && hipStreamIsCapturing(ctx->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    for (auto& f : Y->frag) {
        if (f.ptr && f.bytes == total && f.g == g && f.k2 == K2 && f.fmt == fmt) { hit = &f; break; }
        if (f.pinned) continue;
        // reuse order: an empty slot first, then a slot of another size (its buffer is replaced), then the least recently used
        const auto rank = [&](const covgram_points::FragSlot& s) { return !s.ptr ? 0 : (s.bytes != total ? 1 : 2); };
        if (!victim || rank(f) < rank(*victim) || (rank(f) == rank(*victim) && f.used < victim->used)) victim = &f;
    }
    if (!hit) {
        CG_REQUIRE(victim != nullptr, COVGRAM_EUNSUPPORTED, "all %d fragment slots of this point set are pinned by captured graphs (one per (lengthscale, d) used "
                   "inside a capture): destroy the handle, or keep to %d lengthscales per point set in graphs", covgram_points::FRAG_SLOTS, covgram_points::FRAG_SLOTS);
        // ANY miss during capture is refused: a re-pack captured into the graph would leave the slot's key naming fragments that do not exist
        // until the first replay (an eager MVM in between would read the previous lengthscale's), and it would pin a slot per captured lengthscale
        CG_REQUIRE(!capturing, COVGRAM_EUNSUPPORTED,
                   "the fragments of this (point set, lengthscale) must be packed once OUTSIDE stream capture: run one eager MVM with this lengthscale first");
        if (victim->ptr && victim->bytes != total) {   // only when the handle is re-used at another K2: off the steady-state path
            CG_CHECK_HIP(hipStreamSynchronize(ctx->stream)); (void)hipFree(victim->ptr); victim->ptr = nullptr;
        }
        if (!victim->ptr) {
            hipError_t me = hipMalloc(&victim->ptr, total);
            if (me != hipSuccess) { victim->ptr = nullptr; set_error("hipMalloc(%zu) failed: %s", total, hipGetErrorString(me)); return COVGRAM_ENOMEM; }
            victim->bytes = total;
        }
        // in place, stream-ordered behind every MVM that still reads this slot
        victim->g = g; victim->k2 = K2; victim->fmt = fmt;
        const int64_t pe = ntile * K2 * 64;
        hipLaunchKernelGGL(mfma_pack_kernel, dim3((unsigned)((pe + 255) / 256)), dim3(256), 0, ctx->stream, (const float*)Y->dptr, m, Y->d,
                           (uint4*)victim->ptr, (float*)((char*)victim->ptr + fbytes), K2, g, Cn, fmt);
        hit = victim;
    }
    if (capturing) hit->pinned = true;
    hit->used = ++Y->frag_clock;
    *PB = (const uint4*)hit->ptr;
    *EF = (const float*)((const char*)hit->ptr + fbytes);
    return COVGRAM_OK;
}

// y <- alpha * scale * G a + beta * y for ONE right-hand side (device pointers)
int mvm_eq_mfma(covgram_ctx* ctx, const HostKernel& hk, const covgram_points* X, const covgram_points* Y, const float* a, float* y,
                double alpha, double beta) {
    const int64_t n = X->n, m = Y->n;
    const int d = X->d;
    // Which split (round 4).  The bf16 three-way split is exact to 2^-24 |x~_c y~_c| per coordinate and costs one MFMA per TWO coordinates.  The
    // fp16 two-way split (x~ = h1 + h2, 11 + 11 bits; products h1 h1, h1 h2, h2 h1) drops terms of ~2^-22 |x~_c y~_c| — the size of ONE fp32 rounding
    // of the dot product itself, which the accumulator commits anyway — and covers FOUR coordinates per MFMA: half the matrix-core work per pair.
    // The d = 5 ... 8 kernels are power-bound (2.0 GHz under C3's shard kernel, profiles/r04_c3_power_bound.txt), so that is both fewer cycles and a
    // higher clock.  Taken while BOTH clouds lie within g^2 R^2 <= MFMA_F16_GATE (the added exponent error grows with the radius like the
    // accumulator's own: common.hpp); beyond it, up to MFMA_GATE, the bf16 split serves as before.  Option "mfma_f16": -1 / 1 = this rule,
    // 0 = never, 2 = wherever the matrix-core gate admits the cloud (measurements: tools/f16_split_ab.py).
    const double g2r2 = 1.4426950408889634074 / (hk.k.lengthscale * hk.k.lengthscale) * gate_radius2(X, Y);
    const int fmt = (ctx->mfma_f16 != 0 && g2r2 <= (ctx->mfma_f16 == 2 ? mfma_gate_of(ctx) : mfma_f16_gate_of(ctx))) ? 1 : 0;
    ctx->last_mfma_f16 = fmt;
    const int K2 = fmt ? eq_k2_for(2 * ((d + 3) / 4)) : eq_k2_for(d);      // fp16: one MFMA per four coordinates
    CG_REQUIRE(K2 > 0, COVGRAM_EUNSUPPORTED, "dense_mfma: d = %d has no matrix-core instance", d);
    const int64_t ntile = (m + 31) / 32;
    const float g = (float)(sqrt(1.4426950408889634074) / hk.k.lengthscale);
    const float* Cn = (const float*)Y->center;                    // both sides are taken relative to the column side's centre
    const uint4* PB;
    const float* EF;
    int rc = eq_fragments(ctx, Y, K2, g, Cn, &PB, &EF, fmt);
    if (rc) return rc;
    // the weights a_j exp2(f_j) of THIS right-hand side: formed inside the kernel (round 4) — the wave that fetches a tile loads a_j and the cached
    // exp2(f_j) with it and multiplies where the weight goes to LDS at the end of the stage.  (Rounds 1-2 had measured the in-kernel product 3 %
    // SLOWER than the pack launch: the product stood right behind its two loads and so behind the LDS-DMA the wave had just issued — one counter.
    // Deferred: C2-shaped 1359 -> 1340 us, a 16384-row shard 190.1 -> 186.5, n = 16384 35.1 -> 33.2, n = 2048 12.1 -> 10.8;
    // profiles/r04_fuse_w_ab.txt.)  Option "mfma_fuse_w" = 0 restores the pack launch; results are bit-identical either way.
    // ... unless a and y overlap (an in-place MVM, y == a): the workgroups of a final-store or ticketed launch write y while others still read
    // a — the pack launch copies the weights first, as every dense path did before round 4 (include/covgram.h: aliasing)
    const bool overlap = (const char*)a < (const char*)(y + n) && (const char*)y < (const char*)(a + m);
    const bool fuse_w = ctx->mfma_fuse_w != 0 && !overlap;
    const float* W = a;
    if (!fuse_w) {
        void* Wp;
        rc = ws_reserve(ctx, 0, (size_t)ntile * 32 * sizeof(float), &Wp);
        if (rc) return rc;
        hipLaunchKernelGGL(mfma_pack_w_kernel, dim3((unsigned)((ntile * 32 + 255) / 256)), dim3(256), 0, ctx->stream, a, EF, (float*)Wp, m, ntile * 32);
        W = (const float*)Wp;
    }
    const float* EFk = fuse_w ? EF : nullptr;
    // split the column tiles so that the grid holds ~CUs * 128 waves (as the lane-per-row kernel, profiles/r01_quickbench_wg64.txt)
    // row tiles per wave: two share every B fragment while the state fits (d <= 8); option "rows_per_lane" = 1 / 2 forces it
    int rt = ctx->rows_per_lane == 1 ? 1 : (ctx->rows_per_lane == 2 ? 2 : (K2 <= MFMA_NARROW_MAXK2 ? 2 : 1));
    if (K2 > 8) rt = 1;
    // grid = a whole number (4) of rounds of resident waves: a fractional last round costs up to one round of idle SIMDs
    int nb = 16;
#define CG_NB_CASE(K) case K: nb = mfma_blocks<K>(rt); break;
    switch (K2) {
        CG_NB_CASE(1) CG_NB_CASE(2) CG_NB_CASE(3) CG_NB_CASE(4) CG_NB_CASE(6) CG_NB_CASE(8) CG_NB_CASE(12) CG_NB_CASE(16)
        default: break;
    }
#undef CG_NB_CASE
    struct Plan { int rt; int64_t rowtiles, npad, js, tchunk; bool lds4; };
    auto plan_for = [&](int r) {
        Plan pl; pl.rt = r;
        pl.rowtiles = (n + 32 * r - 1) / (32 * r);
        pl.npad = pl.rowtiles * 32 * r;
        const int64_t target = ctx->target_wgs > 0 ? ctx->target_wgs : (int64_t)ctx->num_cus * nb * 4;
        int64_t js = ctx->jsplit > 0 ? ctx->jsplit : std::max<int64_t>(1, (target + pl.rowtiles / 2) / pl.rowtiles);
        // GP-sized problems (n = 8k ... 32k): a wave's prologue (row fragments), pipeline fill and epilogue (shuffles, slab row) are
        // worth ~40 column tiles of work, so the split stops at 64 tiles per wave — as long as the grid still holds >= 2048 waves,
        // half of the chip's wave slots (tools/eq_small_sweep.py: d = 3, n = 16384 45.9 -> 37.9 us, n = 32768 118.8 -> 113.1;
        // d = 8, n = 16384 60.9 -> 47.4, n = 32768 175 -> 152; n = 8192 17.7 -> 16.9 / 25.3 -> 21.8 us)
        // (few rows — prediction shapes —: the 2048-wave floor itself stops at 16 tiles per wave: 256 x 65536, d = 3: 26.8 -> 22.2 us)
        if (ctx->jsplit <= 0) js = std::max<int64_t>(std::min<int64_t>((2048 + pl.rowtiles - 1) / pl.rowtiles, std::max<int64_t>(1, ntile / 16)),
                                                     std::min<int64_t>(js, std::max<int64_t>(1, ntile / 64)));
        js = std::max<int64_t>(1, std::min<int64_t>(js, std::max<int64_t>(1, ntile / 8)));     // >= 8 tiles (256 columns) per wave
        pl.tchunk = (ntile + js - 1) / js;
        pl.js = (ntile + pl.tchunk - 1) / pl.tchunk;
        // long column chunks: four waves of a workgroup share every column tile through LDS; short chunks would only pay its
        // prologue and barriers (tools/mfma_lds_ab.py)
        pl.lds4 = ((r >= 2 && K2 <= MFMA_NARROW_MAXK2) || (r == 1 && K2 > MFMA_NARROW_MAXK2)) &&
                  (ctx->mfma_lds == 1 || (ctx->mfma_lds < 0 && pl.tchunk >= MFMA_LDS_MIN_TILES && pl.rowtiles >= 64));
        return pl;
    };
    Plan pl = plan_for(rt);
    // d = 7, 8 (four MFMAs per tile) on the LDS-shared form: FOUR row tiles per wave, two waves per SIMD (round 4) — every column fragment
    // read from LDS, every LDS-DMA issue and every stage barrier serves twice the pairs: C3's row shard 4.31 -> 4.13 ms, a 16384-row shard
    // of 131072 columns 313 -> 305 us, 131072^2 at d = 7 1.565 -> 1.516 ms (tools/c3_rt_ab.py; d = 5, 6: 3.67 -> 3.66, left alone).
    // Option "rows_per_lane" = 4 asks for it wherever the instance exists (K2 = 3, 4), 2 keeps two row tiles.
    if ((K2 == 3 || K2 == 4) && pl.lds4 && pl.rt == 2 && (ctx->rows_per_lane == 4 || (ctx->rows_per_lane == 0 && K2 == 4 && !ctx->mfma_stamp))) {
        const Plan p4 = plan_for(4);
        if (p4.lds4) pl = p4;
    }
    rt = pl.rt;
    const int64_t rowtiles = pl.rowtiles, npad = pl.npad, tchunk = pl.tchunk;
    const int64_t js = pl.js;
    const bool lds4 = pl.lds4;
    const double alpha_eff = alpha * hk.kp.scale;
    float* out = y;
    if (js > 1) { void* slab; rc = ws_reserve(ctx, 1, (size_t)js * npad * sizeof(float), &slab); if (rc) return rc; out = (float*)slab; }
    const dim3 grid((unsigned)rowtiles, (unsigned)js);
    const int fs = js == 1 ? 1 : 0;
    // option "mfma_stamp": the clock-stamping diagnostic build of the two LDS-shared instances (C2's and C3's kernels)
    unsigned long long* stamps = nullptr;
    ctx->stamp_count = 0;
    if (ctx->mfma_stamp && lds4 && (K2 == 1 || K2 == 2 || K2 == 4)) {
        const size_t need = (size_t)rowtiles * js * 4 * sizeof(unsigned long long);
        if (need > ctx->stamp_cap) {
            if (ctx->stamp_buf) { CG_CHECK_HIP(hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->stamp_buf); ctx->stamp_buf = nullptr; ctx->stamp_cap = 0; }
            CG_CHECK_HIP(hipMalloc(&ctx->stamp_buf, need));
            ctx->stamp_cap = need;
        }
        CG_CHECK_HIP(hipMemsetAsync(ctx->stamp_buf, 0, need, ctx->stream));
        stamps = (unsigned long long*)ctx->stamp_buf;
    }
    // js > 1: the kernel can sum its own split-J slab (the last workgroup of each row block to arrive, fixed order; pack.hpp).  Measured
    // (profiles/r04_inkernel_reduce_ab.txt, separate launch -> in-kernel): n = 2048 16.2 -> 13.2 us, 4096 16.3 -> 14.9, 8192 17.2 -> 18.8,
    // 16384 37.9 -> 37.2, rank 0 of 8's shard of C2 214.4 -> 213.5, of 16 116.3 -> 115.0: it pays where the whole MVM is launch latency
    // and is noise elsewhere, so the automatic rule takes it up to n = 4096 (option "inkernel_reduce" = 1 / 0 forces it on / off)
    unsigned* tickets = nullptr;
    const bool ikr = inkernel_reduce_on(ctx, true, n, js);
    ctx->last_inkernel_reduce = ikr ? 1 : 0;
    if (ikr) { rc = tickets_reserve(ctx, (size_t)rowtiles, &tickets); if (rc) return rc; }
    auto* tm = timer_next(ctx);
    if (tm) (void)hipEventRecord(tm->first, ctx->stream);
    ctx->last_mfma_lds = lds4 ? 1 : 0;
    dim3 launched;
#define CG_MFMA_CASE(K) case K: if (fmt) { if constexpr (K <= 8) launch_mfma<K, 1>(rt, lds4, grid, ctx->stream, (const float*)X->dptr, n, d, PB, W, ntile, out, npad, tchunk, g, (float)alpha_eff, (float)beta, fs, Cn, stamps, &launched, tickets, y, EFk, m, &ctx->last_mfma_instance); } else launch_mfma<K>(rt, lds4, grid, ctx->stream, (const float*)X->dptr, n, d, PB, W, ntile, out, npad, tchunk, g, (float)alpha_eff, (float)beta, fs, Cn, stamps, &launched, tickets, y, EFk, m, &ctx->last_mfma_instance); break;
    switch (K2) {
        CG_MFMA_CASE(1) CG_MFMA_CASE(2) CG_MFMA_CASE(3) CG_MFMA_CASE(4) CG_MFMA_CASE(6) CG_MFMA_CASE(8) CG_MFMA_CASE(12) CG_MFMA_CASE(16)
        default: set_error("dense_mfma: K2 = %d not compiled", K2); return COVGRAM_EUNSUPPORTED;
    }
#undef CG_MFMA_CASE
    if (tm) (void)hipEventRecord(tm->second, ctx->stream);
    if (stamps) ctx->stamp_count = (size_t)launched.x * launched.y;
    if (js > 1 && !ikr)
        hipLaunchKernelGGL(dense_reduce_kernel<float>, dim3((unsigned)((n + 63) / 64), 1), dim3(256), 0, ctx->stream, (const float*)out, npad, 1,
                           (int)js, y, n, n, 1, (float)alpha_eff, (float)beta);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dense_mfma launch failed: %s", hipGetErrorString(e)); return COVGRAM_EHIP; }
    return COVGRAM_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// Symmetric Gramian (gramian(k, x): y IS x): every tile on or above the diagonal is evaluated ONCE and used twice,
//     b_i += e_i sum_j (a_j e_j) E_ij          (row sums, as dense_mfma_eq_kernel)
//     b_j += e_j sum_i (a_i e_i) E_ij          (column sums of the same exponentials, tiles strictly above the diagonal)
// with E_ij = exp2(x~_i . x~_j + k_i + k_j) from the MFMA and e = exp2(f) in (1/2, 1] the fraction factors of the half-norms
// (dense_mfma.hpp: norm_split): one v_exp_f32 and TWO v_fma_f32 per evaluated pair, i.e. 6 VALU
// issue cycles per Gramian entry instead of 10, and half the MFMAs.  The reference evaluates all n^2 entries
// (src/gramian.jl:78-87 does not look at issymmetric); the sums are the same numbers in a different order.
//
// Work decomposition: a workgroup = 8 waves x 1 row tile = a PANEL of 256 rows; workgroup (p, c) walks the column tiles
// [max(8 p, c tchunk), (c + 1) tchunk) — everything from the panel's own first tile to the right — with the LDS staging
// of the non-symmetric kernel.  Inside the panel's diagonal block the tiles below the diagonal are masked (weight 0 for
// the row sums of tiles J < I, for the column sums of tiles J <= I): < 0.5 % wasted work.  Row sums leave through the
// per-chunk slab R[c][i]; the column sums of a tile are added over the 8 waves in LDS in fixed order and stored to
// S[p][j] (each (p, tile) exactly once); dense_mfma_sym_reduce_kernel adds R over the panel's chunks and S over the panels
// p <= panel(j) in fixed order: deterministic, no float atomics.
// ------------------------------------------------------------------------------------------------------------------------
// b_i = alpha (sum_{chunks c the panel of i visited} R[c][i] + e_i sum_{p <= panel(i)} S[p][i]) + beta b_i, fixed order
// (EF = the fraction factors e_i the EQ form's column sums still lack; nullptr: generic form).
// 64 rows per workgroup, the panel index strided over the 4 waves (as dense_reduce_kernel).
__global__ __launch_bounds__(1024) void dense_mfma_sym_reduce_kernel(int64_t n, const float* __restrict__ R, const float* __restrict__ S, int64_t npad,
                                                                    int64_t ntile, int32_t tchunk, const float* __restrict__ EF, float* __restrict__ y,
                                                                    float alpha, float beta, int32_t pfirst, int32_t pstride, int32_t tpp) {
    // 256 rows per workgroup — FOUR consecutive rows per lane (16-byte loads: the slab is hundreds of MB read once; round 5, 4-byte loads before:
    // 3.1 TB/s) —, the panel index strided over 16 waves (hundreds of panels per row: many independent loads in flight).  Per row the sums are formed
    // in the same order as before (panels part, part + 16, ...; then the 16 parts; then the chunks): bit-identical results.
    // (npad is a multiple of 4 and the slabs are 256-byte aligned; rows >= n of a lane's four are read — the slabs are padded — and not stored)
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int64_t i = ((int64_t)blockIdx.x * 64 + lane) * 4;
    __shared__ f4 red[16][64];
    f4 s = {0.0f, 0.0f, 0.0f, 0.0f};
    if (i < n) {
        const int64_t pi = i / (32 * tpp);                                           // the rows' panel (tpp row tiles: 256 or 128 rows — four rows share it)
        // local panels lp (global pfirst + pstride lp) up to the row's own panel
        const int64_t nlp = pi >= pfirst ? (pi - pfirst) / pstride + 1 : 0;
#pragma unroll 4
        for (int64_t lp = part; lp < nlp; lp += 16) s += *reinterpret_cast<const f4*>(S + lp * npad + i);
    }
    red[part][lane] = s;
    __syncthreads();
    if (part != 0 || i >= n) return;
    f4 cs = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < 16; ++q) cs += red[q][lane];                                  // fixed order
    const int64_t pi = i / (32 * tpp);
    const int64_t cfirst = (tpp * pi) / tchunk, cend = (ntile + tchunk - 1) / tchunk;   // the absolute chunks the row's panel visited
    f4 rs = {0.0f, 0.0f, 0.0f, 0.0f};
    if (pi >= pfirst && (pi - pfirst) % pstride == 0)                                 // the row sums exist only where this rank owns the panel
        for (int64_t c = cfirst; c < cend; ++c) rs += *reinterpret_cast<const f4*>(R + c * npad + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (i + e >= n) break;
        float v = alpha * (EF ? __builtin_fmaf(EF[i + e], cs[e], rs[e]) : cs[e] + rs[e]);
        if (beta != 0.0f) v = __builtin_fmaf(beta, y[i + e], v);
        y[i + e] = v;
    }
}

bool mfma_eq_sym_eligible(const covgram_ctx* ctx, const HostKernel& hk, const covgram_points* X, const covgram_points* Y, int nrhs) {
    if (ctx->mfma_sym == 0 || X->dptr != Y->dptr || X->n != Y->n) return false;       // gramian(k, x): the same point set on both sides
    if (!mfma_eq_eligible(ctx, hk, X, Y, nrhs)) return false;
    const int tpp = eq_k2_for(X->d) > MFMA_NARROW_MAXK2 ? 4 : 8;                    // row tiles per panel (dense_mfma_sym_wide_kernel: 4)
    const int64_t ntile = (X->n + 31) / 32, panels = (ntile + tpp - 1) / tpp;
    if ((size_t)panels * (size_t)(panels * 32 * tpp) * sizeof(float) > ((size_t)16 << 30)) return false;   // column-sum slab <= 16 GiB of the 288
    return ctx->mfma_sym == 1 || X->n >= (X->d <= 4 ? MFMA_SYM_MIN_N_EQ : MFMA_SYM_MIN_N_EQ_WIDE);
}

static int mfma_k2_for(int dims);
static int mfma_gen_fmt(const covgram_ctx* ctx, const HostKernel& hk, int lfam, const covgram_points* X, const covgram_points* Y, int maxk2, int* K2);
static int mfma_family_of(const covgram_ctx* ctx, const HostKernel& hk, float* gamma);
bool mfma_gen_eligible(const covgram_ctx* ctx, const HostKernel& hk, const covgram_points* X, const covgram_points* Y, int nrhs);
// the generic profiles' symmetric form: same conditions on top of the generic matrix-core gate (hk: gamma = 1/l parameter block)
bool mfma_gen_sym_eligible(const covgram_ctx* ctx, const HostKernel& hk, const covgram_points* X, const covgram_points* Y, int nrhs) {
    if (ctx->mfma_sym == 0 || nrhs != 1 || X->dptr != Y->dptr || X->n != Y->n) return false;
    if (!mfma_gen_eligible(ctx, hk, X, Y, nrhs)) return false;
    const int k2 = mfma_k2_for(X->d + (hk.k.trait == COVGRAM_ISOTROPIC ? 1 : 0));
    if (k2 < 0) return false;
    float gam_unused;
    const int lfam = mfma_family_of(ctx, hk, &gam_unused);
    if (lfam == FAM_SUM_ISO && k2 > 8) return false;                              // no one-pass symmetric instance: one (symmetric) MVM per term instead
    const int tpp = mfma_sym_tiles_per_panel(lfam, k2);                           // (an upper bound of the slab whichever form runs: fewer tiles per panel = more panels)
    const int64_t ntile = (X->n + 31) / 32, panels = (ntile + tpp - 1) / tpp;
    if ((size_t)panels * (size_t)(panels * 32 * tpp) * sizeof(float) > ((size_t)16 << 30)) return false;
    const bool heavy = hk.tu_family == COVGRAM_MATERNP || hk.tu_family == COVGRAM_RQ || hk.tu_family >= COVGRAM_NFAMILY;   // profile costs several exponentials
    // (heavy profiles at d > 4: the all-entries kernel's cost grows with the fragment length, the symmetric kernel's floor does not — tools/sym_threshold_sweep.py,
    //  MaternP(2) d = 8: n = 10000 60.0 / 59.8 us, 12000 77.4 / 66.7)
    return ctx->mfma_sym == 1 || X->n >= (heavy ? (X->d > 4 ? 10000 : MFMA_SYM_MIN_N_HEAVY) : MFMA_SYM_MIN_N_EQ_WIDE);
}

// y <- alpha * scale * G_part a + beta * y for the symmetric Gramian of ONE point set and one right-hand side, where G_part
// holds the entries (i, j), (j, i) whose upper-triangle tile lies in the panels pfirst, pfirst + pstride, ... — all of G
// for (0, 1); the partial products of the ranks (g, P), g < P, add up to G a (covgram_mvm_sym_partial).
int mvm_eq_mfma_sym(covgram_ctx* ctx, const HostKernel& hk, const covgram_points* X, const float* a, float* y, double alpha, double beta,
                    int pfirst, int pstride, const covgram_kernel* kgen) {
    // kgen == nullptr: the EQ form (hk is the dense EQ parameter block); else the generic form for kernel *kgen
    const bool fast = kgen == nullptr;
    HostKernel hkg;
    if (!fast) {   // gamma = 1/l and the profile sees s / l^2 — EQ / MaternP: the folded block (dense_mfma.hpp: mfma_folded)
        int rcg = make_host_kernel(kgen, COVGRAM_F32, !(kgen->family == COVGRAM_EQ || kgen->family == COVGRAM_MATERNP), &hkg);
        if (rcg) return rcg;
    }
    const HostKernel& hku = fast ? hk : hkg;
    const bool iso = hku.k.trait == COVGRAM_ISOTROPIC;
    float gamg = 0.0f;
    const int lfam = fast ? COVGRAM_EQ : mfma_family_of(ctx, hku, &gamg);
    if (!fast) ctx->last_sum_fused = lfam == FAM_SUM_ISO ? 1 : 0;
    const int64_t n = X->n;
    const int d = X->d;
    // EQ form: the fp16 two-way split inside its gate, as the general kernel (mvm_eq_mfma, "Which split"); every rank of a multi-GPU call sees the same
    // cloud and lengthscale, hence the same split and the same panel size
    const double g2r2 = 1.4426950408889634074 / (hk.k.lengthscale * hk.k.lengthscale) * gate_radius2(X, X);
    const int fmt = (fast && ctx->mfma_f16 != 0 && g2r2 <= (ctx->mfma_f16 == 2 ? mfma_gate_of(ctx) : mfma_f16_gate_of(ctx))) ? 1 : 0;
    if (fast) ctx->last_mfma_f16 = fmt;
    int K2 = fast ? (fmt ? eq_k2_for(2 * ((d + 3) / 4)) : eq_k2_for(d)) : mfma_k2_for(d + (iso ? 1 : 0));
    // generic form: the fp16 split where the staged (narrow) symmetric kernel of the family has the instance (dense_mfma.hpp: mfma_sym_narrow_maxk2)
    int gfmt = 0;
    if (!fast) {
        int k2h = 0;
        gfmt = mfma_gen_fmt(ctx, hku, lfam, X, X, lfam == FAM_SUM_ISO ? 2 : 8, &k2h);   // (the Sum: its staged kernel only)
        if (gfmt) K2 = k2h;
        ctx->last_mfma_f16 = gfmt;
    }
    CG_REQUIRE(K2 > 0, COVGRAM_EUNSUPPORTED, "dense_mfma_sym: d = %d has no matrix-core instance", d);
    // slab row stride: rows of R / S one panel apart must not sit a power of two apart (n = 49152: 192 KB stride, every panel's
    // stores to the same columns hit the same memory channel: 1.46 ms instead of 0.25); + 4.25 KB staggers them
    // row tiles per panel: 8 waves x 1 tile, 6 for the heavy profiles' 3-waves-per-SIMD form, or 4 x 1 for long fragments (dense_mfma.hpp)
    // generic form at one or two MFMAs per tile: two row tiles per wave as the EQ form (dense_mfma_sym2.hpp; option "mfma_sym_rt" = 1: the one-row-tile panels)
    const int gen_rt = (!fast && ctx->mfma_sym_rt != 1 && K2 <= 2 && mfma_sym2_family(lfam)) ? 2 : 1;
    // Six MFMAs per tile (d = 17 .. 24 with the fp16 split, 11 .. 12 with bf16): the STAGED 8-wave kernel (four column tiles per stage and barrier, one
    // tile at a time in registers) instead of the one-tile-per-stage 4-wave kernel — round 5, tools/wide_staged_ab.py: n = 131072, d = 24 2.19 -> 1.89 ms,
    // d = 20 2.16 -> 1.82.  Eight MFMAs per tile need 169 registers there (two waves per SIMD): 2.55 -> 2.60 ms, so they keep the 4-wave kernel
    // (option "mfma_sym_st" = 16 forces the staged form for measurements).
    const bool staged_wide = fast && (K2 == 6 || (K2 == 8 && fmt == 1 && ctx->mfma_sym_st == 16));
    const int tpp = fast ? ((K2 > MFMA_NARROW_MAXK2 && !staged_wide) ? 4 : 8) : mfma_sym_tiles_per_panel(lfam, K2, gen_rt);
    const int64_t ntile = (n + 31) / 32, panels = (ntile + tpp - 1) / tpp, npad = panels * 32 * tpp + 1088;
    const float g = (float)(sqrt(1.4426950408889634074) / hk.k.lengthscale);
    const float* Cn = (const float*)X->center;
    int rc;
    const float* W;
    const uint4* PBu;
    const float* EF = nullptr;
    if (fast) {
        // the fragments AND the fraction factors e_j are cached in the points handle; the kernel forms a_j e_j itself, so a
        // steady-state MVM launches no pack kernel
        rc = eq_fragments(ctx, X, K2, g, Cn, &PBu, &EF, fmt);
        if (rc) return rc;
        W = a;
    } else {                                                       // generic fragments (norm pseudo-coordinate) + W = a, packed per MVM
        void* P;
        rc = ws_reserve(ctx, 0, (size_t)ntile * ((size_t)K2 * 64 * sizeof(uint4) + 32 * sizeof(float)), &P);
        if (rc) return rc;
        uint4* PB = (uint4*)P;
        float* Wg = (float*)(PB + ntile * K2 * 64);
        const int64_t pe = ntile * K2 * 64;
        hipLaunchKernelGGL(mfma_pack_gen_kernel, dim3((unsigned)((pe + 255) / 256)), dim3(256), 0, ctx->stream, (const float*)X->dptr, n, d, a,
                           n, 1, 0, PB, Wg, K2, 1, gamg, iso ? 1 : 0, Cn, gfmt);
        PBu = PB; W = Wg;
    }
    // column chunk: a multiple of the 4-tile stage; ~8 rounds of the resident workgroups (2 per CU) over the triangle
    const int64_t lpanels = panels > pfirst ? (panels - pfirst + pstride - 1) / pstride : 0;   // this call's panels
    const int64_t tileops = (panels * ntile / 2 + panels * 4) / pstride;         // (panel, tile) visits
    // ~8 rounds of the resident workgroups, ~4 for one rank's share of a multi-GPU MVM (the list below puts the cut, shorter
    // chunks last, so the last round balances); >= 64 tiles — shorter chunks do not amortise a workgroup's prologue (rows, row
    // weights, first stage) —, >= 128 for a rank's share (tools/sym_tchunk_sweep.py, tools/sym_shard_probe.py: rank r of 8 at
    // C2 size 170-185 us with 128-tile chunks, 183-201 with 64)
    const bool rt2_plan = fast && K2 <= (ctx->mfma_sym_rt == 2 ? MFMA_NARROW_MAXK2 : 2) && ctx->mfma_sym_rt != 1;     // 4-wave workgroups: three per CU
    // (4-wave workgroups: three per CU.  The two-row-tile kernels — heavier prologue: two sets of row fragments and weights per wave — take ~5 rounds:
    //  tools/sym_rounds_ab.py, interleaved, n = 131072: EQ d = 3 980 / 962 / 958 us at 8 / 5 / 4 rounds, d = 8 1154 / 1131 / 1117; n = 262144 level)
    const bool two_row = rt2_plan || gen_rt == 2;
    int64_t target = ctx->target_wgs > 0 ? ctx->target_wgs : (int64_t)ctx->num_cus * ((tpp == 4 || two_row) ? 3 : 2) * (pstride > 1 ? 4 : (two_row ? 5 : 8));
    int64_t tchunk = ctx->jsplit > 0 ? (ntile + ctx->jsplit - 1) / ctx->jsplit : (tileops + target - 1) / target;
    tchunk = std::max<int64_t>(pstride > 1 ? 128 : 64, std::min<int64_t>(((tchunk + 3) / 4) * 4, 1024));
    const int64_t maxc = (ntile + tchunk - 1) / tchunk;
    void *Rp, *Sp;
    rc = ws_reserve(ctx, 1, (size_t)maxc * npad * sizeof(float), &Rp); if (rc) return rc;
    rc = ws_reserve(ctx, 4, (size_t)std::max<int64_t>(lpanels, 1) * npad * sizeof(float), &Sp); if (rc) return rc;
    const double alpha_eff = alpha * hku.kp.scale;
    // the (local panel, absolute chunk) pairs that exist, chunk-major (the workgroups in flight share a chunk's fragments in
    // L2); cached in the context by its key
    CG_REQUIRE(maxc <= 4096 && lpanels < ((int64_t)1 << 19), COVGRAM_EUNSUPPORTED, "dense_mfma_sym: work list out of range");
    const int64_t key[4] = {ntile, tchunk * 16 + tpp, pfirst, pstride};
    if (ctx->sym_map == nullptr || memcmp(ctx->sym_key, key, sizeof(key)) != 0) {
        // whole chunks first (chunk-major), then the cut ones (a panel's first chunk, the last chunk of the row): the short
        // workgroups fill the last round instead of leaving most of the chip idle behind a few long ones
        std::vector<int32_t> list, tail;
        for (int64_t c = 0; c < maxc; ++c)
            for (int64_t lp = 0; lp < lpanels; ++lp) {
                const int64_t p8 = tpp * ((int64_t)pfirst + (int64_t)pstride * lp);   // the panel's first tile
                if (p8 >= (c + 1) * tchunk) break;                 // panels are ascending: the rest start right of this chunk
                const int64_t t0 = std::max(c * tchunk, p8), t1 = std::min((c + 1) * tchunk, ntile);
                if (t0 >= ntile) continue;
                ((t1 - t0 == tchunk) ? list : tail).push_back((int32_t)((lp << 12) | c));
            }
        std::sort(tail.begin(), tail.end(), [&](int32_t u, int32_t v) {   // longer cut chunks first
            auto len = [&](int32_t w) {
                const int64_t lp = w >> 12, c = w & 4095, p8 = tpp * ((int64_t)pfirst + (int64_t)pstride * lp);
                return std::min((c + 1) * tchunk, ntile) - std::max(c * tchunk, p8);
            };
            const int64_t lu = len(u), lv = len(v);
            return lu != lv ? lu > lv : u < v;
        });
        list.insert(list.end(), tail.begin(), tail.end());
        if (list.size() > ctx->sym_map_cap) {
            if (ctx->sym_map) { CG_CHECK_HIP(hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->sym_map); ctx->sym_map = nullptr; }
            ctx->sym_map_cap = std::max<size_t>(list.size(), 4096);
            CG_CHECK_HIP(hipMalloc((void**)&ctx->sym_map, ctx->sym_map_cap * sizeof(int32_t)));
        } else if (ctx->sym_map) {
            CG_CHECK_HIP(hipStreamSynchronize(ctx->stream));       // a launch in flight may still read the old list
        }
        if (!list.empty()) CG_CHECK_HIP(hipMemcpy(ctx->sym_map, list.data(), list.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        memcpy(ctx->sym_key, key, sizeof(key));
        ctx->sym_map_len = list.size();
    }
    const dim3 grid((unsigned)std::max<size_t>(ctx->sym_map_len, 1));
    auto* tm = timer_next(ctx);
    if (tm) (void)hipEventRecord(tm->first, ctx->stream);
#define CG_SYM_CASE(K) case K: hipLaunchKernelGGL((dense_mfma_sym_kernel<FAM_EQFAST, K>), grid, dim3(512), 0, ctx->stream, (const float*)X->dptr, n, d, \
                                                  PBu, W, ntile, (float*)Rp, (float*)Sp, npad, (int)tchunk, g, Cn, pfirst, pstride, ctx->sym_map, \
                                                  KParams<float>{}, EF); break;
#define CG_SYMW_CASE(K) case K: hipLaunchKernelGGL((dense_mfma_sym_wide_kernel<FAM_EQFAST, K>), grid, dim3(256), 0, ctx->stream, (const float*)X->dptr, n, d, \
                                                   PBu, W, ntile, (float*)Rp, (float*)Sp, npad, (int)tchunk, g, Cn, pfirst, pstride, ctx->sym_map, \
                                                   KParams<float>{}, EF); break;
#define CG_SYMH_CASE(K) case K: hipLaunchKernelGGL((dense_mfma_sym_kernel<FAM_EQFAST_H, K>), grid, dim3(512), 0, ctx->stream, (const float*)X->dptr, n, d, \
                                                   PBu, W, ntile, (float*)Rp, (float*)Sp, npad, (int)tchunk, g, Cn, pfirst, pstride, ctx->sym_map, \
                                                   KParams<float>{}, EF); break;
#define CG_SYMWH_CASE(K) case K: hipLaunchKernelGGL((dense_mfma_sym_wide_kernel<FAM_EQFAST_H, K>), grid, dim3(256), 0, ctx->stream, (const float*)X->dptr, n, d, \
                                                    PBu, W, ntile, (float*)Rp, (float*)Sp, npad, (int)tchunk, g, Cn, pfirst, pstride, ctx->sym_map, \
                                                    KParams<float>{}, EF); break;
    ctx->last_mfma_instance = fast ? -(int64_t)(K2 * 10 + fmt) : 0;
    // the EQ forms up to four MFMAs per tile: two row tiles per wave, 4-wave workgroups (dense_mfma_sym2.hpp) — option "mfma_sym_rt": -1 / 2 = that, 1 = the
    // 8-wave one-row-tile kernel
    const bool rt2 = fast && K2 <= (ctx->mfma_sym_rt == 2 ? MFMA_NARROW_MAXK2 : 2) && ctx->mfma_sym_rt != 1;
    ctx->last_mfma_sym_rt = rt2 ? 2 : 1;
#define CG_SYM2_CASE(F, K) case K: if (K <= 2 && ctx->mfma_sym_st != 4) hipLaunchKernelGGL((dense_mfma_sym2_kernel<F, K, (K <= 2 ? 8 : 4)>), grid, dim3(256), 0, ctx->stream, (const float*)X->dptr, n, d, \
                                                      PBu, W, ntile, (float*)Rp, (float*)Sp, npad, (int)tchunk, g, Cn, pfirst, pstride, ctx->sym_map, KParams<float>{}, EF); \
                                   else hipLaunchKernelGGL((dense_mfma_sym2_kernel<F, K, 4>), grid, dim3(256), 0, ctx->stream, (const float*)X->dptr, n, d, \
                                                      PBu, W, ntile, (float*)Rp, (float*)Sp, npad, (int)tchunk, g, Cn, pfirst, pstride, ctx->sym_map, KParams<float>{}, EF); break;
    if (rt2 && fmt) {
        switch (K2) { CG_SYM2_CASE(FAM_EQFAST_H, 1) CG_SYM2_CASE(FAM_EQFAST_H, 2) CG_SYM2_CASE(FAM_EQFAST_H, 3) CG_SYM2_CASE(FAM_EQFAST_H, 4) default: break; }
    } else if (rt2) {
        switch (K2) { CG_SYM2_CASE(FAM_EQFAST, 1) CG_SYM2_CASE(FAM_EQFAST, 2) CG_SYM2_CASE(FAM_EQFAST, 3) CG_SYM2_CASE(FAM_EQFAST, 4) default: break; }
    } else
#undef CG_SYM2_CASE
    if (fast && fmt) {
        switch (K2) {
            CG_SYMH_CASE(1) CG_SYMH_CASE(2) CG_SYMH_CASE(3) CG_SYMH_CASE(4)
#define CG_SYMSW_CASE(K) case K: if (staged_wide) hipLaunchKernelGGL((dense_mfma_sym_kernel<FAM_EQFAST_H, K>), grid, dim3(512), 0, ctx->stream, (const float*)X->dptr, n, d, \
                                                   PBu, W, ntile, (float*)Rp, (float*)Sp, npad, (int)tchunk, g, Cn, pfirst, pstride, ctx->sym_map, KParams<float>{}, EF); \
                               else hipLaunchKernelGGL((dense_mfma_sym_wide_kernel<FAM_EQFAST_H, K>), grid, dim3(256), 0, ctx->stream, (const float*)X->dptr, n, d, \
                                                   PBu, W, ntile, (float*)Rp, (float*)Sp, npad, (int)tchunk, g, Cn, pfirst, pstride, ctx->sym_map, KParams<float>{}, EF); break;
            CG_SYMSW_CASE(6) CG_SYMSW_CASE(8)
#undef CG_SYMSW_CASE
            default: set_error("dense_mfma_sym: K2 = %d not compiled for the fp16 split", K2); return COVGRAM_EUNSUPPORTED;
        }
    } else if (fast) {
        switch (K2) {
            CG_SYM_CASE(1) CG_SYM_CASE(2) CG_SYM_CASE(3) CG_SYM_CASE(4)
            case 6: hipLaunchKernelGGL((dense_mfma_sym_kernel<FAM_EQFAST, 6>), grid, dim3(512), 0, ctx->stream, (const float*)X->dptr, n, d, PBu, W, ntile, (float*)Rp,
                                       (float*)Sp, npad, (int)tchunk, g, Cn, pfirst, pstride, ctx->sym_map, KParams<float>{}, EF); break;
            CG_SYMW_CASE(8) CG_SYMW_CASE(12) CG_SYMW_CASE(16)
            default: set_error("dense_mfma_sym: K2 = %d not compiled", K2); return COVGRAM_EUNSUPPORTED;
        }
    } else {
        MfmaArgs ma;
        ma.X = (const float*)X->dptr; ma.n = n; ma.d = d; ma.PB = PBu; ma.W = W; ma.ntile = ntile; ma.out = nullptr; ma.npad = npad; ma.ldy = n;
        ma.nrhs = 1; ma.tchunk = tchunk; ma.alpha = 1.0f; ma.beta = 0.0f; ma.final_store = 0; ma.K2 = K2; ma.RT = 1; ma.NR = 1;
        ma.hk = &hku; ma.stream = ctx->stream; ma.grid = grid; ma.Cn = Cn;
        ma.sym = 1; ma.R = (float*)Rp; ma.S = (float*)Sp; ma.wgmap = ctx->sym_map; ma.pfirst = pfirst; ma.pstride = pstride; ma.fmt = gfmt; ma.sym_rt = gen_rt;
        ctx->last_mfma_sym_rt = gen_rt;
        mfma_launch_fn launch = mfma_launcher(lfam);
        CG_REQUIRE(launch != nullptr, COVGRAM_EUNSUPPORTED, "dense_mfma_sym: family %d has no matrix-core path", lfam);
        rc = launch(ma, false);
        if (rc) return rc;
    }
#undef CG_SYM_CASE
#undef CG_SYMW_CASE
#undef CG_SYMH_CASE
#undef CG_SYMWH_CASE
    if (tm) (void)hipEventRecord(tm->second, ctx->stream);
    hipLaunchKernelGGL(dense_mfma_sym_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(1024), 0, ctx->stream, n, (const float*)Rp,
                       (const float*)Sp, npad, ntile, (int)tchunk, EF, y, (float)alpha_eff, (float)beta, pfirst, pstride, tpp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dense_mfma_sym launch failed: %s", hipGetErrorString(e)); return COVGRAM_EHIP; }
    return COVGRAM_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// generic profiles / several right-hand sides (dense_mfma.hpp)
// ------------------------------------------------------------------------------------------------------------------------
#define CG_DECL(n) int launch_mfma_family_##n(const MfmaArgs&, bool query);
CG_DECL(0) CG_DECL(2) CG_DECL(4) CG_DECL(5) CG_DECL(6) CG_DECL(7) CG_DECL(8) CG_DECL(11) CG_DECL(12) CG_DECL(13)
#undef CG_DECL

mfma_launch_fn mfma_launcher(int family) {
    switch (family) {
        case COVGRAM_EQ: return launch_mfma_family_0;
        case COVGRAM_RQ: return launch_mfma_family_2;
        case COVGRAM_CAUCHY: return launch_mfma_family_4;
        case COVGRAM_IMQ: return launch_mfma_family_5;
        case COVGRAM_MATERNP: return launch_mfma_family_6;
        case COVGRAM_DOT: return launch_mfma_family_7;
        case COVGRAM_EXPDOT: return launch_mfma_family_8;
        case FAM_EXPR_ISO: return launch_mfma_family_11;          // Sum / Product / Power composites of the profiles above
        case FAM_EXPR_DOT: return launch_mfma_family_12;
        case FAM_SUM_ISO: return launch_mfma_family_13;           // a Sum of 2-3 single-profile isotropic terms in one pass (common.hpp: SumParams)
        default: return nullptr;
    }
}

// The generic kernels' split of the coordinates (round 5): the fp16 two-way split (dense_mfma.hpp: gen_row_fragments) for the isotropic single
// profiles and one-pass Sums while (a) its instance exists — d + 2 positions in at most `maxk2` MFMAs of four —, and (b) the cloud lies inside
// MFMA_F16_GATE / MFMA_GATE of the generic radius gate (gate_frac: the quantity mfma_gen_eligible admits up to 1), the EQ kernel's rule
// (option "mfma_f16": 0 never, 2 wherever the matrix-core gate admits the cloud).  Returns the format and sets the MFMAs per tile.
static double mfma_gen_gate_frac(const HostKernel& hk, const covgram_points* X, const covgram_points* Y);
static int mfma_gen_fmt(const covgram_ctx* ctx, const HostKernel& hk, int lfam, const covgram_points* X, const covgram_points* Y, int maxk2, int* K2) {
    const bool iso = hk.k.trait == COVGRAM_ISOTROPIC;
    const int d = X->d;
    *K2 = mfma_k2_for(d + (iso ? 1 : 0));
    if (ctx->mfma_f16 == 0 || !iso || lfam == FAM_EXPR_ISO || lfam == FAM_EXPR_DOT) return 0;
    int k2h = (d + 2 + 3) / 4;                                   // d + 2 positions, four per MFMA; compiled: 1, 2, 3, 4, 6, 8 (d <= 30)
    if (k2h == 5) k2h = 6; else if (k2h == 7) k2h = 8;
    if (k2h > maxk2 || k2h > 8) return 0;
    const double frac = mfma_gen_gate_frac(hk, X, Y);
    if (!(frac <= 0.01 * (double)ctx->mfma_gate_pct * (ctx->mfma_f16 == 2 ? 1.0 : MFMA_F16_GATE / MFMA_GATE))) return 0;
    *K2 = k2h;
    return 1;
}

// The launcher family and coordinate pre-scale of a generic matrix-core MVM: a Sum of single-profile terms runs its one-pass form
// (FAM_SUM_ISO: the pre-scale carries the first term's argument scale) unless option "sum_fused" = 0 keeps the composite interpreter.
static int mfma_family_of(const covgram_ctx* ctx, const HostKernel& hk, float* gamma) {
    if (ctx->sum_fused != 0 && sum_fusable(hk)) { *gamma = (float)make_sum_params(hk).gamma; return FAM_SUM_ISO; }
    *gamma = (float)hk.kp.gamma;
    return hk.tu_family;
}
// covgram_mvm: does this composite take the one-pass Sum kernels (instead of one MVM per term)?
bool sum_fused_applies(const covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, const covgram_points* Y, int nrhs) {
    if (ctx->sum_fused == 0 || k == nullptr || k->family != COVGRAM_COMPOSITE || X->dtype != COVGRAM_F32 || Y->n == 0) return false;
    HostKernel hk;
    if (make_host_kernel(k, COVGRAM_F32, false, &hk) != COVGRAM_OK) return false;
    if (!sum_fusable(hk) || !mfma_gen_eligible(ctx, hk, X, Y, nrhs)) return false;
    // gramian(k, x) at d >= 16: the one-pass symmetric kernel has no instance (K2 > 8), and one symmetric MVM per term evaluates half the pairs
    // of the one-pass general kernel
    const bool same = nrhs == 1 && ctx->mfma_sym != 0 && X->dptr == Y->dptr && X->n == Y->n && (ctx->mfma_sym == 1 || X->n >= (X->d > 4 ? 10000 : MFMA_SYM_MIN_N_HEAVY));
    if (same && mfma_k2_for(X->d + 1) > 8) return false;
    // Measured (d = 3, n = 131072; profiles/r05_sum_fused_ab.txt, r05_sum_general_ab.txt): what one pass shares is the matrix-core work, the weighted
    // sums and the slabs — every term's transcendentals remain, and they are what the kernels are bound by.  Three terms: 4.07 against 4.83 ms for one
    // symmetric MVM per term, 0.95 / 1.01 against 0.98 / 1.09 on a 16384-row shard.  Two terms: one MVM per term wins or ties on both forms (symmetric
    // 3.05 against 3.32; shard, MaternP(2) + EQ: 0.62 against 0.71 — the EQ term costs 0.15 on its own kernel —, MaternP + MaternP 0.83 against 0.87,
    // RQ + MaternP 0.79 against 0.78): the one-pass kernel's wave-uniform family switches and its one-tile loop cost more than the second pass saves.
    // Option "sum_fused" = 1 takes the one-pass kernels wherever they exist (tests).
    if (ctx->sum_fused == 1) return true;
    return hk.nterms >= 3;
}

static int mfma_k2_for(int dims) {   // MFMAs per tile for `dims` (pseudo-)coordinates, from the compiled set
    static const int ks[] = {1, 2, 3, 4, 6, 8, 12, 16};
    const int need = (dims + 1) / 2;
    for (int k : ks) if (k >= need) return k;
    return -1;
}

// (radius-gate quantity of an ISOTROPIC generic matrix-core MVM) / (what mfma_gen_eligible admits): sensitivity x Power x g^2 R^2 over
// 0.5 MFMA_GATE / log2(e), single profiles and composites alike; infinity where a profile has no matrix-core form
static double mfma_gen_gate_frac(const HostKernel& hk, const covgram_points* X, const covgram_points* Y) {
    const double limit = 0.5 * MFMA_GATE / 1.4426950408889634074;
    auto sens_of = [](int fam, const KParams<double>& q) -> double {
        switch (fam) {
            case COVGRAM_EQ: case COVGRAM_RQ: return 0.5;
            case COVGRAM_CAUCHY: return 1.0;
            case COVGRAM_IMQ: return 0.5 / q.param;                      // param holds c^2
            case COVGRAM_MATERNP: return q.p < 1 ? INFINITY : (fabs(q.mp_d1) > 0.5 ? fabs(q.mp_d1) : 0.5);
            default: return INFINITY;
        }
    };
    const double Pn = gate_radius2(X, Y);
    if (!(Pn < 1e30)) return INFINITY;
    if (hk.tu_family >= COVGRAM_NFAMILY) {
        double worst = 0.0;
        int fi = 0;
        for (int t = 0; t < hk.nterms; ++t) {
            double sum = 0.0;
            for (int f = 0; f < hk.nfac[t]; ++f, ++fi) sum += sens_of(hk.ffam[fi], hk.fkp[fi]) * hk.fkp[fi].power * hk.fkp[fi].gamma2;
            worst = std::max(worst, sum);
        }
        return worst * Pn / limit;
    }
    return sens_of(hk.tu_family, hk.kp) * hk.k.power * Pn / (hk.k.lengthscale * hk.k.lengthscale) / limit;
}

bool mfma_gen_eligible(const covgram_ctx* ctx, const HostKernel& hk, const covgram_points* X, const covgram_points* Y, int nrhs) {
    if (ctx->dense_variant == 1 || X->dtype != COVGRAM_F32 || Y->n == 0) return false;
    if (mfma_launcher(hk.tu_family) == nullptr) return false;
    if (hk.tu_family == COVGRAM_MATERNP && hk.k.p < 1) return false;          // MaternP(0) = Exp: not differentiable in s at 0
    // MaternP(p >= 1): up to round 4 the lane-per-row kernels (packed fp32 profile) beat the matrix-core ones (profile entry by entry) at d <= 4.
    // Round 5: the profile arithmetic runs on register pairs (dense_mfma.hpp: mfma_profile_pairs) and the order is a compile-time constant of the
    // instance: the general kernel is ahead at every d (16384 x 131072, d = 3: 428 against 543 us lane-per-row; d = 8: 631 against 862;
    // profiles/r05_sum_fused_ab.txt), the symmetric one too (n = 131072, d = 3: 2.0 against 2.6 ms) — MaternP(p >= 1) no longer has an exception here.
    const bool iso = hk.k.trait == COVGRAM_ISOTROPIC;
    if (hk.tu_family >= COVGRAM_NFAMILY) {
        // composite: every factor must be one of the smooth matrix-core profiles, and the relative errors of a product add up:
        // the largest sum over a term's factors of (sensitivity x power / l^2) takes the place of the single profile's
        if (mfma_k2_for(X->d + (iso ? 1 : 0)) < 0) return false;
        double worst = 0.0;
        int fi = 0;
        for (int t = 0; t < hk.nterms; ++t) {
            double sum = 0.0;
            for (int f = 0; f < hk.nfac[t]; ++f, ++fi) {
                const KParams<double>& q = hk.fkp[fi];
                double sens;
                switch (hk.ffam[fi]) {
                    case COVGRAM_EQ: case COVGRAM_RQ: sens = 0.5; break;
                    case COVGRAM_CAUCHY: sens = 1.0; break;
                    case COVGRAM_IMQ: sens = 0.5 / q.param; break;                 // param holds c^2
                    case COVGRAM_MATERNP: if (q.p < 1) return false; sens = fabs(q.mp_d1) > 0.5 ? fabs(q.mp_d1) : 0.5; break;
                    case COVGRAM_DOT: case COVGRAM_EXPDOT: sens = 0.0; break;
                    default: return false;                                     // Exp, gammaExp, Matern(nu), asin: direct differences only
                }
                sum += sens * q.power * q.gamma2;
            }
            worst = std::max(worst, sum);
        }
        if (ctx->dense_variant == 2) return true;
        if (!iso) return sqrt(X->max_norm2) * sqrt(Y->max_norm2) < 1e30;
        const double Pn = gate_radius2(X, Y);                                  // natural units: every factor has its own 1 / l^2
        return worst * Pn <= 0.5 * mfma_gate_of(ctx) / 1.4426950408889634074;
    }
    if (mfma_k2_for(X->d + (iso ? 1 : 0)) < 0) return false;
    if (ctx->dense_variant == 2) return true;
    const double P = (iso ? gate_radius2(X, Y) : sqrt(X->max_norm2) * sqrt(Y->max_norm2)) /
                     (hk.k.lengthscale * hk.k.lengthscale);
    if (!(P < 1e30)) return false;
    if (!iso) return true;                                                     // x.y itself: no cancellation to gate
    // relative sensitivity of the profile to an absolute error in s (natural units): |phi'/phi| at its maximum
    double sens = 1.0;
    switch (hk.tu_family) {
        case COVGRAM_EQ: case COVGRAM_RQ: sens = 0.5; break;
        case COVGRAM_CAUCHY: sens = 1.0; break;
        case COVGRAM_IMQ: sens = 0.5 / (hk.k.param * hk.k.param); break;   // 1/sqrt(s + c^2) at s = 0 (s already / l^2)
        case COVGRAM_MATERNP: sens = fabs(hk.kp.mp_d1) > 0.5 ? fabs(hk.kp.mp_d1) : 0.5; break;
        default: break;
    }
    return sens * hk.k.power * P <= 0.5 * mfma_gate_of(ctx) / 1.4426950408889634074;   // the EQ gate in natural units
}

// y[:, 0..nrhs) <- alpha * scale * G a + beta * y (device pointers, column-major with lda / ldy)
int mvm_mfma_gen(covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, const covgram_points* Y, const float* a, int64_t lda,
                 float* y, int64_t ldy, int32_t nrhs, double alpha, double beta) {
    HostKernel hk;
    // gamma = 1/l and the profile sees s / l^2 — EQ / MaternP: the folded block (dense_mfma.hpp: mfma_folded)
    int rc = make_host_kernel(k, COVGRAM_F32, !(k->family == COVGRAM_EQ || k->family == COVGRAM_MATERNP), &hk);
    if (rc) return rc;
    const int64_t n = X->n, m = Y->n;
    const int d = X->d;
    const bool iso = hk.k.trait == COVGRAM_ISOTROPIC;
    const int64_t ntile = (m + 31) / 32;
    float gam;
    const int lfam = mfma_family_of(ctx, hk, &gam);
    int K2bf = 0;
    const int gfmt_all = mfma_gen_fmt(ctx, hk, lfam, X, Y, 8, &K2bf);   // the VALU forms (up to four right-hand sides); the many-column GEMM form keeps bf16
    const int K2m = mfma_k2_for(d + (iso ? 1 : 0));                      // ... with this many MFMAs per tile
    int K2 = K2m;
    ctx->last_sum_fused = lfam == FAM_SUM_ISO ? 1 : 0;
    mfma_launch_fn launch = mfma_launcher(lfam);
    const double alpha_eff = alpha * hk.kp.scale;
    int cdone = 0;
    // many right-hand sides: blocks of up to 64 on the fp32 matrix cores (dense_mfma_mrhs_kernel).  Its time is flat up to 32 columns
    // (n = 32768: 0.76 ms EQ, 0.86 ms MaternP(2)); the VALU form steps up every four columns (EQ d = 3: 0.29 / 0.56 / 0.81 ms at 4 / 8 /
    // 12; MaternP(2): 0.48 / 0.94 / 1.40): the crossover is 9 columns for the cheap profiles at d <= 3, 5 otherwise —
    // option "mfma_mrhs": -1 this rule, 0 never, 1 from 2 columns (tests)
    const bool cheap = K2 <= 2 && (hk.tu_family == COVGRAM_EQ || hk.tu_family == COVGRAM_CAUCHY || hk.tu_family == COVGRAM_IMQ ||
                                   hk.tu_family == COVGRAM_DOT || hk.tu_family == COVGRAM_EXPDOT);
    const int mrhs_min = ctx->mfma_mrhs == 1 ? 2 : (cheap ? 9 : 5);
    while (nrhs - cdone >= mrhs_min && ctx->mfma_mrhs != 0) {
        const int c0 = cdone;
        const int nr = std::min(64, nrhs - c0);
        const int NB = nr > 32 ? 2 : 1;
        float* y_c = y + (size_t)c0 * ldy;
        void* P;
        const size_t fb = (size_t)ntile * K2 * 64 * sizeof(uint4), wb = (size_t)ntile * 32 * sizeof(float);
        rc = ws_reserve(ctx, 0, fb + wb + (size_t)ntile * NB * 64 * 16 * sizeof(float), &P);
        if (rc) return rc;
        uint4* PB = (uint4*)P;
        float* Wd = (float*)((char*)P + fb);                       // the pack kernel's weight output: unused here
        float* AP = (float*)((char*)P + fb + wb);
        const int64_t pe = ntile * K2 * 64;
        hipLaunchKernelGGL(mfma_pack_gen_kernel, dim3((unsigned)((pe + 255) / 256)), dim3(256), 0, ctx->stream, (const float*)Y->dptr, m, d, a,
                           lda, 0, 0, PB, Wd, K2, 1, gam, iso ? 1 : 0, (const float*)Y->center, 0);
        const int64_t ae = ntile * NB * 64 * 16;
        hipLaunchKernelGGL(mfma_pack_rhs_kernel, dim3((unsigned)((ae + 255) / 256)), dim3(256), 0, ctx->stream, a, lda, c0 + nr, c0, m, AP, NB);
        const int64_t rowtiles = (n + 31) / 32, npad = rowtiles * 32;
        int64_t target = ctx->target_wgs > 0 ? ctx->target_wgs : (int64_t)ctx->num_cus * 32;        // the fp32 MFMAs bound this kernel: 2 rounds of 4 waves per SIMD
        int64_t js = ctx->jsplit > 0 ? ctx->jsplit : std::max<int64_t>(1, (target + rowtiles / 2) / rowtiles);
        js = std::max<int64_t>(1, std::min<int64_t>(js, std::max<int64_t>(1, ntile / 8)));
        const int64_t tchunk = (ntile + js - 1) / js;
        js = (ntile + tchunk - 1) / tchunk;
        float* out = y_c;
        if (js > 1) { void* slab; rc = ws_reserve(ctx, 1, (size_t)js * 32 * NB * npad * sizeof(float), &slab); if (rc) return rc; out = (float*)slab; }
        MfmaArgs ma;
        ma.K2 = K2; ma.NR = 32 * NB; ma.RT = 1; ma.mrhs = NB;
        ma.hk = &hk; ma.stream = ctx->stream; ma.Cn = (const float*)Y->center;
        ma.X = (const float*)X->dptr; ma.n = n; ma.d = d; ma.PB = PB; ma.W = AP; ma.ntile = ntile; ma.out = out; ma.npad = npad; ma.ldy = ldy;
        ma.nrhs = nr; ma.tchunk = tchunk; ma.alpha = (float)alpha_eff; ma.beta = (float)beta; ma.final_store = js == 1 ? 1 : 0;
        ma.grid = dim3((unsigned)rowtiles, (unsigned)js);
        auto* tm = timer_next(ctx);
        if (tm) (void)hipEventRecord(tm->first, ctx->stream);
        rc = launch(ma, false);
        if (rc) return rc;
        if (tm) (void)hipEventRecord(tm->second, ctx->stream);
        if (js > 1)
            hipLaunchKernelGGL(dense_reduce_kernel<float>, dim3((unsigned)((n + 63) / 64), nr), dim3(256), 0, ctx->stream, (const float*)out, npad,
                               32 * NB, (int)js, y_c, n, ldy, nr, (float)alpha_eff, (float)beta);
        cdone += nr;
    }
    const int gfmt = gfmt_all;
    if (gfmt) K2 = K2bf;                                                  // (mfma_gen_fmt returned the fp16 split's MFMAs per tile)
    ctx->last_mfma_f16 = gfmt;
    for (int c0 = cdone; c0 < nrhs; c0 += 4) {
        const int nr = std::min(4, nrhs - c0);
        const int NR = nr == 1 ? 1 : 4;
        const float* a_c = a + (size_t)c0 * lda;
        float* y_c = y + (size_t)c0 * ldy;
        void* P;
        rc = ws_reserve(ctx, 0, (size_t)ntile * ((size_t)K2 * 64 * sizeof(uint4) + (size_t)NR * 32 * sizeof(float)), &P);
        if (rc) return rc;
        uint4* PB = (uint4*)P;
        float* W = (float*)(PB + ntile * K2 * 64);
        const int64_t pe = ntile * K2 * 64;
        hipLaunchKernelGGL(mfma_pack_gen_kernel, dim3((unsigned)((pe + 255) / 256)), dim3(256), 0, ctx->stream, (const float*)Y->dptr, m, d, a_c,
                           lda, nr, 0, PB, W, K2, NR, gam, iso ? 1 : 0, (const float*)Y->center, gfmt);
        MfmaArgs ma;
        ma.K2 = K2; ma.NR = NR; ma.fmt = gfmt;
        ma.RT = (NR == 1 && K2 <= 4 && ctx->rows_per_lane != 1) ? 2 : 1;
        ma.hk = &hk; ma.stream = ctx->stream; ma.Cn = (const float*)Y->center;
        const int nb = launch(ma, true);
        const int64_t rowtiles = (n + 32 * ma.RT - 1) / (32 * ma.RT);
        const int64_t npad = rowtiles * 32 * ma.RT;
        int64_t target = ctx->target_wgs > 0 ? ctx->target_wgs : (int64_t)ctx->num_cus * nb * 4;
        int64_t js = ctx->jsplit > 0 ? ctx->jsplit : std::max<int64_t>(1, (target + rowtiles / 2) / rowtiles);
        // (the 64-tiles-per-wave stop of mvm_eq_mfma does not carry over: these profiles cost several times EQ's per tile, so a wave's fixed
        //  costs amortise over far fewer tiles — tools/rect_sweep.py, MaternP(2) 4096 x 65536: 32 splits 156 us, 128-256 splits 120-127 us)
        js = std::max<int64_t>(1, std::min<int64_t>(js, std::max<int64_t>(1, ntile / 8)));
        const int64_t tchunk = (ntile + js - 1) / js;
        js = (ntile + tchunk - 1) / tchunk;
        float* out = y_c;
        if (js > 1) { void* slab; rc = ws_reserve(ctx, 1, (size_t)js * NR * npad * sizeof(float), &slab); if (rc) return rc; out = (float*)slab; }
        ma.X = (const float*)X->dptr; ma.n = n; ma.d = d; ma.PB = PB; ma.W = W; ma.ntile = ntile; ma.out = out; ma.npad = npad; ma.ldy = ldy;
        ma.nrhs = nr; ma.tchunk = tchunk; ma.alpha = (float)alpha_eff; ma.beta = (float)beta; ma.final_store = js == 1 ? 1 : 0;
        ma.grid = dim3((unsigned)rowtiles, (unsigned)js);
        // long fragments (d > 8) and long column chunks: four waves share every column tile through LDS, one tile per stage
        // (tools/mfma_lds_ab.py <kernel>: 11-28 % faster at d = 12 .. 24; at d <= 8 the barrier per tile costs 5-18 %, so not there)
        ma.lds = (ctx->mfma_lds == 1 || (ctx->mfma_lds < 0 && K2 > 4 && tchunk >= MFMA_LDS_MIN_TILES && rowtiles >= 64)) ? 1 : 0;
        ctx->last_mfma_lds = ma.lds;
        auto* tm = timer_next(ctx);
        if (tm) (void)hipEventRecord(tm->first, ctx->stream);
        rc = launch(ma, false);
        if (rc) return rc;
        if (tm) (void)hipEventRecord(tm->second, ctx->stream);
        if (js > 1)
            hipLaunchKernelGGL(dense_reduce_kernel<float>, dim3((unsigned)((n + 63) / 64), nr), dim3(256), 0, ctx->stream, (const float*)out, npad,
                               NR, (int)js, y_c, n, ldy, nr, (float)alpha_eff, (float)beta);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dense_mfma launch failed: %s", hipGetErrorString(e)); return COVGRAM_EHIP; }
    return COVGRAM_OK;
}

}  // namespace covgram
