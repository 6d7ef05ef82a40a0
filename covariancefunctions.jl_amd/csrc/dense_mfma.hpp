// dense_mfma.hpp — the matrix-core dense Gramian MVM for fp32 profiles other than the specialised EQ kernel of
// dense_mfma.hip (whose header explains the bf16 three-way split and the tile orientation), and for several right-hand
// sides.  Generic form: the MFMA produces the profile argument itself,
//     isotropic:    s_ij = |x~_i|^2 + |y~_j|^2 - 2 x~_i . y~_j      (x~ = x / l; the two norms ride in one extra pseudo-
//                   coordinate: A slots [nx1, nx2, nx3, 1, 1, 1, 0, 0] against B slots [1, 1, 1, ny1, ny2, ny3, 0, 0])
//     dot product:  s_ij = x_i . y_j                                 (no cancellation: as accurate as the fmaf chain)
// and the VALU evaluates phi(s) and NR weighted accumulations per pair.  Profiles that are not differentiable in s at 0
// (Exp = MaternP(0), gammaExp) never take this path: an absolute error of 1e-7 P in s would become 3e-4 sqrt(P) in r.
#pragma once
#include "dense_mvm.hpp"

namespace covgram {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// round-to-nearest-even bf16 of a finite float, as its 16-bit pattern
__device__ __forceinline__ unsigned bf16_bits(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
// f = p1 + p2 + p3 (+ O(2^-27 f)) with bf16 pieces
__device__ __forceinline__ void split3(float f, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = bf16_bits(f);
    const float r1 = f - __uint_as_float(p1 << 16);
    p2 = bf16_bits(r1);
    const float r2 = r1 - __uint_as_float(p2 << 16);
    p3 = bf16_bits(r2);
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
union Frag {
    uint4 u;
    bf16x8 v;
    f16x8 hv;
};

// ---- the fp16 two-way split of the general EQ kernel (round 4; dense_mfma.hip, "Which split") ----------------------------------
// f = h1 + h2 (+ O(2^-23 f)) with fp16 pieces (11 significant bits each; |f| <= sqrt(126): no overflow, and pieces below 6e-5 are fp16
// subnormals with an ABSOLUTE resolution of 6e-8 — the exponent's error budget is absolute)
__device__ __forceinline__ unsigned f16_bits(float f) { return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)f); }
__device__ __forceinline__ void split2h(float f, unsigned& h1, unsigned& h2) {
    const _Float16 a = (_Float16)f;
    h1 = (unsigned)__builtin_bit_cast(unsigned short, a);
    h2 = f16_bits(f - (float)a);
}
constexpr unsigned F16_ONE = 0x3C00u;
// MFMAs per tile: a lane's 8 K-slots hold TWO coordinates x 3 products (x1 y1, x1 y2, x2 y1) + two slots for the norms' integer parts
// (used by lane half 0 of MFMA 0 only), so one v_mfma_f32_32x32x16_f16 covers four coordinates
template <int FMT> constexpr int eq_coords_per_mfma = FMT == 1 ? 4 : 2;
template <int FMT>
__device__ __forceinline__ f32x16 eq_mma(const Frag& a, const Frag& b, f32x16 c) {
    if constexpr (FMT == 1) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hv, b.hv, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
}


constexpr unsigned BF16_ONE = 0x3F80u;
// EQ kernels up to this many MFMAs per tile (d <= 8) keep two row tiles per wave, share staged column tiles through LDS and
// use the 8-wave symmetric panels; longer fragments take the one-tile-per-stage forms
constexpr int MFMA_NARROW_MAXK2 = 4;

// ---- the norms of the fp32 EQ kernels (dense_mfma.hip) ------------------------------------------------------------------
// The exponent of an entry is  x~.y~ - |x~|^2/2 - |y~|^2/2 = -|x~ - y~|^2/2.  Each half-norm is split into an INTEGER and
// a fraction,  -|x~|^2/2 = k + f  with k = ceil(.) <= 0 and f in (-1, 0]:  the integers (exact in bf16 up to 256; the gate
// keeps them <= 63) ride through the MFMA in two K-slots, [k_i] x [1] and [1] x [k_j], so the matrix cores deliver
//     x~.y~ + k_i + k_j  =  -|x~ - y~|^2/2 - f_i - f_j   (<= 2),
// and the fractions are fp32 factors exp2(f) in (1/2, 1] — on the row side applied once per row, on the column side folded
// into the weight a_j exp2(f_j).  Nothing can overflow (every exponential is <= 4), no row or column is lost to an
// underflowing norm factor (round 1 kept exp2(-|x~|^2/2) whole: 2^-160 for a point 18 scaled units out), and the weights
// stay within a factor 2 of the caller's a_j over the whole fp32 range.
// Where the two slots live: d odd — in the idle coordinate c = d of the last MFMA (nothing is displaced); d even — in place
// of the two smallest split products (x2 y3, x3 y2 ~ 2^-24 |x_0 y_0|) of coordinate 0, so K2 stays d / 2.
struct NormSplit { unsigned kbits; float ef; };
__device__ __forceinline__ NormSplit norm_split(double n2) {       // n2 = |x~|^2 (fp64 sum of squares of the fp32 coordinates)
    const double hn = -0.5 * n2;
    const double k = __builtin_ceil(hn);
    NormSplit s;
    s.kbits = bf16_bits((float)k);
    s.ef = __builtin_amdgcn_exp2f((float)(hn - k));
    return s;
}

// A fragments of one row for the EQ kernels: lane (t, h) holds the split of x~[row][c = 2 mm + h]; returns exp2(f_i)
template <int K2>
__device__ __forceinline__ float eq_row_fragments(const float* __restrict__ xr, const float* __restrict__ Cn, int d, float g, int h,
                                                  Frag (&a)[K2]) {
    double part = 0.0;
#pragma unroll
    for (int mm = 0; mm < K2; ++mm) {
        const int c = 2 * mm + h;
        const float xt = (c < d) ? g * (xr[c] - Cn[c]) : 0.0f;
        part = __builtin_fma((double)xt, (double)xt, part);
        unsigned x1, x2, x3;
        split3(xt, x1, x2, x3);
        a[mm].u = make_uint4(x1 | (x1 << 16), x2 | (x1 << 16), x2 | (x3 << 16), x2 | (x3 << 16));
    }
    const NormSplit s = norm_split(part + __shfl_xor(part, 32));
    if (d & 1) {                                                   // idle coordinate c = d: slots [k_i, 1, 0, ...]
#pragma unroll
        for (int mm = 0; mm < K2; ++mm)
            if (2 * mm + h == d) a[mm].u = make_uint4(s.kbits | (BF16_ONE << 16), 0, 0, 0);
    } else if (h == 0) {                                           // coordinate 0: [x1, x1, x2, x1, x2, x3, k_i, 1]
        a[0].u.w = s.kbits | (BF16_ONE << 16);
    }
    return s.ef;
}

// the same for the fp16 two-way split: lane (t, h) of MFMA mm holds coordinates 4 mm + 2 h and 4 mm + 2 h + 1 as [x1, x1, x2 | x1, x1, x2 | k_i, 1]
// (the last two slots in lane half 0 of MFMA 0 only, zero elsewhere), against the column side's [y1, y2, y1 | y1, y2, y1 | 1, k_j]
template <int K2>
__device__ __forceinline__ float eq_row_fragments_h(const float* __restrict__ xr, const float* __restrict__ Cn, int d, float g, int h,
                                                    Frag (&a)[K2]) {
    double part = 0.0;
#pragma unroll
    for (int mm = 0; mm < K2; ++mm) {
        const int c0 = 4 * mm + 2 * h;
        const float xa = (c0 < d) ? g * (xr[c0] - Cn[c0]) : 0.0f;
        const float xb = (c0 + 1 < d) ? g * (xr[c0 + 1] - Cn[c0 + 1]) : 0.0f;
        part = __builtin_fma((double)xa, (double)xa, part);
        part = __builtin_fma((double)xb, (double)xb, part);
        unsigned a1, a2, b1, b2;
        split2h(xa, a1, a2);
        split2h(xb, b1, b2);
        a[mm].u = make_uint4(a1 | (a1 << 16), a2 | (b1 << 16), b1 | (b2 << 16), 0u);
    }
    const double n2 = part + __shfl_xor(part, 32);
    const double hn = -0.5 * n2;
    const double k = __builtin_ceil(hn);
    if (h == 0) a[0].u.w = f16_bits((float)k) | (F16_ONE << 16);
    return __builtin_amdgcn_exp2f((float)(hn - k));
}
template <int K2, int FMT>
__device__ __forceinline__ float eq_row_fragments_fmt(const float* __restrict__ xr, const float* __restrict__ Cn, int d, float g, int h, Frag (&a)[K2]) {
    if constexpr (FMT == 1) return eq_row_fragments_h<K2>(xr, Cn, d, g, h, a);
    else return eq_row_fragments<K2>(xr, Cn, d, g, h, a);
}

// ---- fragments of the GENERIC kernels (the MFMA yields the profile argument s = |x~|^2 + |y~|^2 - 2 x~.y~, or x.y) ------------------------
// GFMT 0: bf16 three-way split — a lane half of MFMA mm holds ONE coordinate c = 2 mm + h in its 8 K-slots, the two norms ride in the
//         pseudo-coordinate c = d (header of this file).
// GFMT 1 (round 5): fp16 two-way split, as the EQ kernel's (dense_mfma.hip, "Which split") — a lane half holds TWO positions q = 4 mm + 2 h + {0, 1}
//         of three slots each: a coordinate as [x1, x1, x2] against [y1, y2, y1] (y carries the -2), the ROW norm at position d as [k, f1, f2]
//         against [1, 1, 1], the COLUMN norm at position d + 1 as [1, 1, 1] against [k', f1', f2'] — a norm n = k + f with k = floor(n) exact in
//         fp16 (the gate keeps n < 2048) and f in [0, 1) as two fp16 pieces (2^-22 absolute).  Half the MFMAs per tile of the bf16 split from d = 5
//         (d = 8: 3 against 6, which also moves it from the one-tile-per-stage kernels to the staged ones); what it drops is ~2^-22 |x~_c y~_c| per
//         coordinate, so it is taken inside a tighter radius gate (dense_mfma.hip: mfma_gen_fmt).  Isotropic profiles only.
struct Norm16 { unsigned k, f1, f2; };
__device__ __forceinline__ Norm16 norm16(float n) {
    const float k = __builtin_floorf(n), f = n - k;
    const _Float16 a = (_Float16)f;
    Norm16 r;
    r.k = f16_bits(k); r.f1 = (unsigned)__builtin_bit_cast(unsigned short, a); r.f2 = f16_bits(f - (float)a);
    return r;
}
// the three slots of position q on the ROW side (lane of row `xr`)
__device__ __forceinline__ void gen16_row_triple(const float* __restrict__ xr, const float* __restrict__ Cn, int d, float g, float nx, int q,
                                                 unsigned& s0, unsigned& s1, unsigned& s2) {
    s0 = s1 = s2 = 0u;
    if (q < d) { unsigned x1, x2; split2h(g * (xr[q] - Cn[q]), x1, x2); s0 = x1; s1 = x1; s2 = x2; }
    else if (q == d) { const Norm16 nn = norm16(nx); s0 = nn.k; s1 = nn.f1; s2 = nn.f2; }
    else if (q == d + 1) { s0 = s1 = s2 = F16_ONE; }
}
template <int K2, int GFMT, bool ISO>
__device__ __forceinline__ void gen_row_fragments(const float* __restrict__ xr, const float* __restrict__ Cn, int d, float g, int h, Frag (&a)[K2]) {
    float nx = 0.0f;
    if constexpr (ISO)
        for (int cc = 0; cc < d; ++cc) { const float xc = g * (xr[cc] - Cn[cc]); nx = __builtin_fmaf(xc, xc, nx); }
    if constexpr (GFMT == 1) {
        static_assert(ISO || GFMT == 0, "the fp16 split serves the isotropic profiles");
#pragma unroll
        for (int mm = 0; mm < K2; ++mm) {
            const int q0 = 4 * mm + 2 * h;
            unsigned p0, p1, p2, r0, r1, r2;
            gen16_row_triple(xr, Cn, d, g, nx, q0, p0, p1, p2);
            gen16_row_triple(xr, Cn, d, g, nx, q0 + 1, r0, r1, r2);
            a[mm].u = make_uint4(p0 | (p1 << 16), p2 | (r0 << 16), r1 | (r2 << 16), 0u);
        }
    } else {
#pragma unroll
        for (int mm = 0; mm < K2; ++mm) {
            const int c = 2 * mm + h;
            uint4 f = make_uint4(0, 0, 0, 0);
            if (c < d) {
                unsigned x1, x2, x3;
                split3(ISO ? g * (xr[c] - Cn[c]) : g * xr[c], x1, x2, x3);
                f = make_uint4(x1 | (x1 << 16), x2 | (x1 << 16), x2 | (x3 << 16), x2 | (x3 << 16));
            } else if (ISO && c == d) {
                unsigned n1, n2, n3;
                split3(nx, n1, n2, n3);
                f = make_uint4(n1 | (n2 << 16), n3 | (BF16_ONE << 16), BF16_ONE | (BF16_ONE << 16), 0);
            }
            a[mm].u = f;
        }
    }
}

template <int FAM> constexpr bool mfma_folded = (FAM == COVGRAM_EQ || FAM == COVGRAM_MATERNP);

// ---- the profile on a tile, in register pairs (round 5) -------------------------------------------------------------------------
// A lane holds 16 entries of a 32 x 32 tile in 16 consecutive registers = 8 aligned pairs.  Only the transcendental instructions
// (v_sqrt / v_exp / v_log / v_rcp / v_rsq_f32: quarter rate, 8 issue cycles per wave) work entry by entry; everything around them — the
// polynomial of MaternP, the argument scaling of RQ / Cauchy / IMQ, the Power wrapper, a Sum's per-term scaling and accumulation — is
// v_pk_fma / v_pk_mul / v_pk_add_f32 on the pairs (two entries per instruction at ~2.2 cycles against ~2.7 each; profiles/
// r01_microbench_valu_rates.txt).  Round 4 had done this for the EQ kernel's weighted sums only; MaternP(2) on the symmetric kernel at C2
// size: 33 -> 22 issue cycles per 64 evaluated entries.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_splat(float x) { return (f32x2){x, x}; }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// sqrt of a profile argument that rounding may have left at -1e-7 P: |s| through the instruction's free source modifier instead of a
// separate v_max_f32 (either way the result is a rounding-sized r where the exact one is 0)
__device__ __forceinline__ f32x2 pk_sqrt_abs(f32x2 s) { return (f32x2){__builtin_amdgcn_sqrtf(__builtin_fabsf(s[0])), __builtin_amdgcn_sqrtf(__builtin_fabsf(s[1]))}; }
__device__ __forceinline__ f32x2 pk_exp2_neg(f32x2 s) { return (f32x2){__builtin_amdgcn_exp2f(-s[0]), __builtin_amdgcn_exp2f(-s[1])}; }
__device__ __forceinline__ f32x2 pk_exp2(f32x2 s) { return (f32x2){__builtin_amdgcn_exp2f(s[0]), __builtin_amdgcn_exp2f(s[1])}; }
__device__ __forceinline__ f32x2 pk_log2(f32x2 s) { return (f32x2){__builtin_amdgcn_logf(s[0]), __builtin_amdgcn_logf(s[1])}; }
__device__ __forceinline__ f32x2 pk_rcp(f32x2 s) { return (f32x2){__builtin_amdgcn_rcpf(s[0]), __builtin_amdgcn_rcpf(s[1])}; }
__device__ __forceinline__ f32x2 pk_rsq(f32x2 s) { return (f32x2){__builtin_amdgcn_rsqf(s[0]), __builtin_amdgcn_rsqf(s[1])}; }

// Register pairs are worked on NP = 4 at a time (8 entries): the symmetric kernels run 4 waves per SIMD = 128 VGPRs, and a whole tile's
// worth of temporaries (16 roots + 16 exponentials beside the 16 arguments, the accumulators and the row weights) spilled to scratch.
constexpr int PNP = 4;

// MaternP of order P on the folded argument (sqrt(s) = r log2 e): q(sqrt s) exp2(-sqrt s), q's coefficients as splat pairs
template <int P>
__device__ __forceinline__ void maternp_pairs(f32x2 (&S)[PNP], f32x2 h0, f32x2 h1, f32x2 h2, f32x2 h3) {
    f32x2 r[PNP], e[PNP];
#pragma unroll
    for (int v = 0; v < PNP; ++v) r[v] = pk_sqrt_abs(S[v]);
#pragma unroll
    for (int v = 0; v < PNP; ++v) e[v] = pk_exp2_neg(r[v]);
#pragma unroll
    for (int v = 0; v < PNP; ++v) {
        f32x2 q;
        if constexpr (P == 1) q = pk_fma(h1, r[v], h0);
        else if constexpr (P == 2) q = pk_fma(pk_fma(h2, r[v], h1), r[v], h0);
        else q = pk_fma(pk_fma(pk_fma(h3, r[v], h2), r[v], h1), r[v], h0);
        S[v] = q * e[v];
    }
}

// One single-profile family on PNP pairs, in place: s -> k(s); ORDER = MaternP's order (decided once per tile by the caller), 0 otherwise
template <int FAM, int ORDER, typename KP>
__device__ __forceinline__ void mfma_profile_pairs(f32x2 (&S)[PNP], const KP& kp) {
    if constexpr (FAM == COVGRAM_MATERNP) {
        if constexpr (ORDER >= 1 && ORDER <= 3) {
            maternp_pairs<ORDER>(S, pk_splat(kp.h0[0]), pk_splat(kp.h0[1]), pk_splat(kp.h0[2]), pk_splat(kp.h0[3]));
        } else {
#pragma unroll
            for (int v = 0; v < PNP; ++v)
                S[v] = (f32x2){Phi<FAM, float, true>::eval(__builtin_fabsf(S[v][0]), kp), Phi<FAM, float, true>::eval(__builtin_fabsf(S[v][1]), kp)};
        }
    } else if constexpr (FAM == COVGRAM_EQ) {                       // folded: exp2(-s)
#pragma unroll
        for (int v = 0; v < PNP; ++v) S[v] = pk_exp2_neg(S[v]);
    } else if constexpr (FAM == COVGRAM_RQ) {                       // (1 + s / (2 alpha))^-alpha = exp2(-alpha log2 u)
        const f32x2 c0 = pk_splat(kp.c0), one = pk_splat(1.0f), na = pk_splat(-kp.param);
#pragma unroll
        for (int v = 0; v < PNP; ++v) S[v] = pk_log2(pk_fma(S[v], c0, one));
#pragma unroll
        for (int v = 0; v < PNP; ++v) S[v] = pk_exp2(S[v] * na);
    } else if constexpr (FAM == COVGRAM_CAUCHY) {
        const f32x2 one = pk_splat(1.0f);
#pragma unroll
        for (int v = 0; v < PNP; ++v) S[v] = pk_rcp(S[v] + one);
    } else if constexpr (FAM == COVGRAM_IMQ) {
        const f32x2 c2 = pk_splat(kp.param);
#pragma unroll
        for (int v = 0; v < PNP; ++v) S[v] = pk_rsq(S[v] + c2);
    } else if constexpr (FAM == COVGRAM_EXPDOT) {
        const f32x2 l2e = pk_splat(1.44269504088896340736f);
#pragma unroll
        for (int v = 0; v < PNP; ++v) S[v] = pk_exp2(S[v] * l2e);
    } else if constexpr (FAM == COVGRAM_DOT) {
    } else {
#pragma unroll
        for (int v = 0; v < PNP; ++v)
            S[v] = (f32x2){Phi<FAM, float, mfma_folded<FAM>>::eval(S[v][0], kp), Phi<FAM, float, mfma_folded<FAM>>::eval(S[v][1], kp)};
    }
}

// ---- a Sum of single-profile terms in one pass (common.hpp: SumParams) -----------------------------------------------------------
// Term T on PNP pairs: K (+)= coef_T phi_T(S ratio_T).  T is a compile-time term index (its constants are hoisted into SGPRs), FAMILY and
// ORDER are compile-time too: the caller takes the wave-uniform switches once per tile and term.  LAST: S may be overwritten.
template <int T, int FAMILY, int ORDER, typename KP>
__device__ __forceinline__ void mfma_sum_term_pairs(f32x2 (&S)[PNP], f32x2 (&K)[PNP], const KP& kp) {
    constexpr bool FIRST = T == 0;
    const f32x2 c0 = pk_splat(kp.c[T][0]), c1 = pk_splat(kp.c[T][1]);
    f32x2 A[PNP];
    if constexpr (FIRST) {
#pragma unroll
        for (int v = 0; v < PNP; ++v) A[v] = S[v];
    } else {
        const f32x2 ratio = pk_splat(kp.ratio[T]);
#pragma unroll
        for (int v = 0; v < PNP; ++v) A[v] = S[v] * ratio;
    }
    if constexpr (FAMILY == COVGRAM_MATERNP) {
        maternp_pairs<ORDER>(A, c0, c1, pk_splat(kp.c[T][2]), pk_splat(kp.c[T][3]));
#pragma unroll
        for (int v = 0; v < PNP; ++v) K[v] = FIRST ? A[v] : K[v] + A[v];
    } else {
        if constexpr (FAMILY == COVGRAM_EQ) {
#pragma unroll
            for (int v = 0; v < PNP; ++v) A[v] = pk_exp2_neg(A[v]);
        } else if constexpr (FAMILY == COVGRAM_RQ) {
            const f32x2 one = pk_splat(1.0f);
#pragma unroll
            for (int v = 0; v < PNP; ++v) A[v] = pk_log2(A[v] + one);
#pragma unroll
            for (int v = 0; v < PNP; ++v) A[v] = pk_exp2(A[v] * c1);
        } else if constexpr (FAMILY == COVGRAM_CAUCHY) {
            const f32x2 one = pk_splat(1.0f);
#pragma unroll
            for (int v = 0; v < PNP; ++v) A[v] = pk_rcp(A[v] + one);
        } else {                                                    // COVGRAM_IMQ
#pragma unroll
            for (int v = 0; v < PNP; ++v) A[v] = pk_rsq(A[v] + c1);
        }
#pragma unroll
        for (int v = 0; v < PNP; ++v) K[v] = FIRST ? c0 * A[v] : pk_fma(c0, A[v], K[v]);
    }
}
// term T on the whole tile (both halves behind ONE family / order switch)
template <int T, typename KP>
__device__ __forceinline__ void mfma_sum_term(f32x2 (&S)[2][PNP], f32x2 (&K)[2][PNP], const KP& kp) {
    const int fam = kp.fam[T];
#define CG_SUM_BOTH(F, O) { mfma_sum_term_pairs<T, F, O>(S[0], K[0], kp); mfma_sum_term_pairs<T, F, O>(S[1], K[1], kp); }
    if (fam == COVGRAM_MATERNP) {
        const int p = kp.p[T];
        if (p == 2) CG_SUM_BOTH(COVGRAM_MATERNP, 2)
        else if (p == 1) CG_SUM_BOTH(COVGRAM_MATERNP, 1)
        else CG_SUM_BOTH(COVGRAM_MATERNP, 3)
    } else if (fam == COVGRAM_EQ) CG_SUM_BOTH(COVGRAM_EQ, 0)
    else if (fam == COVGRAM_RQ) CG_SUM_BOTH(COVGRAM_RQ, 0)
    else if (fam == COVGRAM_CAUCHY) CG_SUM_BOTH(COVGRAM_CAUCHY, 0)
    else CG_SUM_BOTH(COVGRAM_IMQ, 0)
#undef CG_SUM_BOTH
}

// The profile on the 16 entries a lane holds of one tile: s -> k(s), in place.  Everything uniform over the launch (MaternP's order, the Power
// wrapper, a Sum's families) is decided ONCE per tile: inside a per-entry loop hipcc kept those tests as scalar branches around every entry.
// ORD: MaternP's order as a compile-time constant (1 .. 3; the host picks the instance), 0 = decided here by a wave-uniform switch
template <int FAM, int ORD = 0, typename KP>
__device__ __forceinline__ void mfma_profile_block(f32x16& D, const KP& kp) {
    f32x2 S[2][PNP];
#pragma unroll
    for (int v = 0; v < 8; ++v) S[v / PNP][v % PNP] = (f32x2){D[2 * v], D[2 * v + 1]};
    if constexpr (FAM == FAM_SUM_ISO) {
        f32x2 K[2][PNP];
        mfma_sum_term<0>(S, K, kp);
        mfma_sum_term<1>(S, K, kp);
        if constexpr (ORD == 3) mfma_sum_term<2>(S, K, kp);        // (a Sum's ORD is its number of terms: 2 or 3; 0 = decided here)
        else if constexpr (ORD == 0) { if (kp.nterms > 2) mfma_sum_term<2>(S, K, kp); }
#pragma unroll
        for (int v = 0; v < 8; ++v) { D[2 * v] = K[v / PNP][v % PNP][0]; D[2 * v + 1] = K[v / PNP][v % PNP][1]; }
        return;
    } else if constexpr (FAM == COVGRAM_MATERNP && ORD >= 1 && ORD <= 3) {
        mfma_profile_pairs<FAM, ORD>(S[0], kp); mfma_profile_pairs<FAM, ORD>(S[1], kp);
    } else if constexpr (FAM == COVGRAM_MATERNP) {
        const int p = kp.p;
        if (p == 2) { mfma_profile_pairs<FAM, 2>(S[0], kp); mfma_profile_pairs<FAM, 2>(S[1], kp); }
        else if (p == 1) { mfma_profile_pairs<FAM, 1>(S[0], kp); mfma_profile_pairs<FAM, 1>(S[1], kp); }
        else if (p == 3) { mfma_profile_pairs<FAM, 3>(S[0], kp); mfma_profile_pairs<FAM, 3>(S[1], kp); }
        else { mfma_profile_pairs<FAM, 0>(S[0], kp); mfma_profile_pairs<FAM, 0>(S[1], kp); }
    } else {
        mfma_profile_pairs<FAM, 0>(S[0], kp); mfma_profile_pairs<FAM, 0>(S[1], kp);
    }
    if constexpr (ORD == 0) if (kp.power != 1) {           // (instances with a compile-time order serve power == 1 only: the host's dispatch)
        const int pw = kp.power;
#pragma unroll
        for (int v = 0; v < 8; ++v) { const f32x2 b = S[v / PNP][v % PNP]; f32x2 r = b; for (int i = 1; i < pw; ++i) r = r * b; S[v / PNP][v % PNP] = r; }
    }
#pragma unroll
    for (int v = 0; v < 8; ++v) { D[2 * v] = S[v / PNP][v % PNP][0]; D[2 * v + 1] = S[v / PNP][v % PNP][1]; }
}
// the weighted sums behind it as v_pk_fma_f32 on register pairs: with one or two MFMAs per tile (beside a RUNNING MFMA a packed fma costs more than
// the two it replaces, tools/pkfma_probe.hip) and for the profiles whose VALU work dwarfs the matrix pipe's at any d
template <int FAM, int K2> constexpr bool mfma_pk_sums = K2 <= 2 || FAM == COVGRAM_MATERNP || FAM == COVGRAM_RQ || FAM == FAM_SUM_ISO;

// EQ and MaternP take the dense path's FOLDED parameter block here too (log2(e) and sqrt(2p+1) in the coordinate pre-scale,
// rescaled tables: exp2 of the MFMA result, no multiplications in front) — the host passes make_host_kernel(.., for_gradient = false)

// LDS = 0: one wave per workgroup, its own fragment loads; LDS = 2: four waves on consecutive row tiles share every column tile
// through LDS, one tile per stage, its K2 fragment slices fetched by the waves in turn (as dense_mfma_eq_kernel<.., 4, 2>)
template <int FAM, int K2, int RT, int NR, int LDS = 0, int ORD = 0, int GFMT = 0>
__global__ __launch_bounds__(LDS ? 256 : 64) void dense_mfma_gen_kernel(const float* __restrict__ X, int64_t n, int32_t d,
                                                            const uint4* __restrict__ PB, const float* __restrict__ W, int64_t ntile,
                                                            float* __restrict__ out, int64_t npad, int64_t ldy, int32_t nrhs,
                                                            int64_t tchunk, float alpha, float beta, int32_t final_store,
                                                            const float* __restrict__ Cn, const typename ParamsOf<FAM, float>::type kp) {
    constexpr bool ISO = fam_is_iso<FAM>;
    constexpr int WPB = LDS ? 4 : 1;
    const int l = threadIdx.x & 63, t = l & 31, h = l >> 5;
    const int64_t i0 = ((int64_t)blockIdx.x * WPB + (threadIdx.x >> 6)) * (32 * RT);
    const float g = kp.gamma;
    Frag a[RT][K2];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        int64_t row = i0 + 32 * r + t;
        if (row >= n) row = n - 1;                               // clamp: computed, never stored
        gen_row_fragments<K2, GFMT, ISO>(X + row * (int64_t)d, Cn, d, g, h, a[r]);
    }

    const int64_t T0 = (int64_t)blockIdx.y * tchunk;
    const int64_t T1 = (T0 + tchunk < ntile) ? (T0 + tchunk) : ntile;
    constexpr bool PKS = mfma_pk_sums<FAM, K2>;                  // the weighted sums as v_pk_fma_f32 on register pairs
    f32x2 acc[RT][NR][8];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < NR; ++c)
#pragma unroll
            for (int v = 0; v < 8; ++v) acc[r][c][v] = (f32x2){0.0f, 0.0f};

    const uint4* __restrict__ pbase = PB + (T0 * K2) * 64;
    const float* __restrict__ wbase = W + T0 * (NR * 32);
    const int nt = (int)(T1 - T0);
    auto load_tile = [&](int ti, Frag (&f)[K2], float (&w)[NR]) {
        const int tc = ti < nt ? ti : nt - 1;
#pragma unroll
        for (int mm = 0; mm < K2; ++mm) f[mm].u = pbase[(tc * K2 + mm) * 64 + l];
#pragma unroll
        for (int c = 0; c < NR; ++c) w[c] = wbase[(tc * NR + c) * 32 + t];
    };
    auto process = [&](const Frag (&f)[K2], const float (&w)[NR]) {
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            f32x16 D = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int mm = 0; mm < K2; ++mm) D = eq_mma<GFMT>(a[r][mm], f[mm], D);
            if constexpr (fam_is_expr<FAM>) {                                       // composite: all 16 entries factor by factor
                float sv[16], kv[16];
#pragma unroll
                for (int v = 0; v < 16; ++v) sv[v] = ISO ? fmaxf(D[v], 0.0f) : D[v];
                expr_value_block<float, ISO, 16>(sv, kp, kv);
#pragma unroll
                for (int v = 0; v < 16; ++v) D[v] = kv[v];
            } else {
                mfma_profile_block<FAM, ORD>(D, kp);
            }
#pragma unroll
            for (int v = 0; v < 8; ++v)
#pragma unroll
                for (int c = 0; c < NR; ++c) {
                    if constexpr (PKS) acc[r][c][v] = pk_fma(pk_splat(w[c]), (f32x2){D[2 * v], D[2 * v + 1]}, acc[r][c][v]);
                    else { acc[r][c][v][0] = __builtin_fmaf(w[c], D[2 * v], acc[r][c][v][0]); acc[r][c][v][1] = __builtin_fmaf(w[c], D[2 * v + 1], acc[r][c][v][1]); }
                }
        }
    };
    if constexpr (LDS == 2) {
        __shared__ uint4 tfA[K2][64], tfB[K2][64];
        __shared__ float twA[NR][32], twB[NR][32];
        typedef __attribute__((address_space(1))) const void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        float gw[NR];
#define CG_DMA2(tile, TF)                                                                       \
        {                                                                                       \
            const int ti_ = (tile);                                                             \
            const int tc_ = ti_ < nt ? ti_ : nt - 1;                                            \
            _Pragma("unroll") for (int q = 0; q < (K2 + WPB - 1) / WPB; ++q) {                  \
                const int mm = wv + q * WPB;                                                    \
                if (mm < K2) __builtin_amdgcn_global_load_lds((gptr_t)(pbase + (tc_ * K2 + mm) * 64 + l), (lptr_t)&TF[mm][0], 16, 0, 0); \
            }                                                                                   \
            if (wv == WPB - 1) {                                                                \
                _Pragma("unroll") for (int c = 0; c < NR; ++c) gw[c] = wbase[(tc_ * NR + c) * 32 + t] * (ti_ < nt ? 1.0f : 0.0f); \
            }                                                                                   \
        }
#define CG_PUTW(TW) if (wv == WPB - 1 && h == 0) { _Pragma("unroll") for (int c = 0; c < NR; ++c) TW[c][t] = gw[c]; }
#define CG_TILE2(TF, TW)                                                                        \
        {                                                                                       \
            Frag f[K2];                                                                         \
            float w[NR];                                                                        \
            _Pragma("unroll") for (int mm = 0; mm < K2; ++mm) f[mm].u = TF[mm][l];              \
            _Pragma("unroll") for (int c = 0; c < NR; ++c) w[c] = TW[c][t];                     \
            process(f, w);                                                                      \
        }
        CG_DMA2(0, tfA)
        CG_PUTW(twA)
        __syncthreads();
        for (int ti = 0; ti < nt; ti += 2) {
            CG_DMA2(ti + 1, tfB)                                    // past the chunk: a re-fetch nobody reads
            CG_TILE2(tfA, twA)
            CG_PUTW(twB)
            __syncthreads();
            if (ti + 1 >= nt) break;
            CG_DMA2(ti + 2, tfA)
            CG_TILE2(tfB, twB)
            CG_PUTW(twA)
            __syncthreads();
        }
#undef CG_DMA2
#undef CG_PUTW
#undef CG_TILE2
    } else {
    Frag f0[K2], f1[K2];
    float w0[NR], w1[NR];
    load_tile(0, f0, w0);
    for (int ti = 0; ti < nt; ti += 2) {
        load_tile(ti + 1, f1, w1);
        process(f0, w0);
        load_tile(ti + 2, f0, w0);
        if (ti + 1 < nt) process(f1, w1);
    }
    }

    const int vsel = (t & 3) + 4 * (t >> 3);
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int64_t i = i0 + 32 * r + t;
        const bool mine = ((t >> 2) & 1) == h && i < n;
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            float tot = 0.0f;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float s = acc[r][c][v >> 1][v & 1];
                s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8); s += __shfl_xor(s, 16);
                tot = (vsel == v) ? s : tot;
            }
            if (!mine) continue;
            if (final_store) {
                if (c < nrhs) {
                    float* yp = out + i + (int64_t)c * ldy;
                    float v = alpha * tot;
                    if (beta != 0.0f) v = __builtin_fmaf(beta, *yp, v);
                    *yp = v;
                }
            } else {
                out[((int64_t)blockIdx.y * NR + c) * npad + i] = tot;
            }
        }
    }
}

// ---- many right-hand sides: mul!(B, G, A) with A m x p, p >= 12 (src/gramian.jl:89-99) -----------------------------------------
// The accumulation B[i, :] += E[i, j] A[j, :] is a GEMM with the freshly evaluated 32 x 32 tile E as one factor, so it runs on the
// matrix cores too — in fp32, bit-for-bit the fmaf chain of the VALU form: v_mfma_f32_32x32x2_f32.  The first MFMA is issued with
// its operands swapped (column fragments as A, row fragments as B), so the tile comes out TRANSPOSED: lane (i, h) holds, in
// register v, the entry for row i and column j = 8 (v / 4) + 4 h + v % 4 — exactly the B operand E^T[k = h][n = i] of step v of
//     acc[c][i] += sum_{k = 0, 1} A^T[c][j(v, k)] * E^T[j(v, k)][i]            (32 right-hand sides c per accumulator tile),
// whose A operand — lane (c, h): a[j(v, h)][c] — is pre-packed per MVM in that order (mfma_pack_rhs_kernel: 16 floats per lane and
// tile, contiguous).  16 fp32 MFMAs (64 cycles each) per column tile and 32 right-hand sides against 16 x 32 v_fma_f32 (64 cycles per
// FOUR right-hand sides) in dense_mfma_gen_kernel<.., NR = 4>, which also re-evaluates the kernel once per four.
// One wave per workgroup, one row tile, NB = 1 or 2 blocks of 32 right-hand sides per pass.
template <int FAM, int K2, int NB>
__global__ __launch_bounds__(64) void dense_mfma_mrhs_kernel(const float* __restrict__ X, int64_t n, int32_t d,
                                                             const uint4* __restrict__ PB, const float* __restrict__ AP, int64_t ntile,
                                                             float* __restrict__ out, int64_t npad, int64_t ldy, int32_t nrhs,
                                                             int64_t tchunk, float alpha, float beta, int32_t final_store,
                                                             const float* __restrict__ Cn, const typename ParamsOf<FAM, float>::type kp) {
    constexpr bool ISO = fam_is_iso<FAM>;
    const int l = threadIdx.x & 63, t = l & 31, h = l >> 5;
    const int64_t i0 = (int64_t)blockIdx.x * 32;
    const float g = kp.gamma;
    Frag a[K2];
    {
        int64_t row = i0 + t;
        if (row >= n) row = n - 1;                               // clamp: computed, never stored
        const float* __restrict__ xr = X + row * (int64_t)d;
        float nx = 0.0f;
        if constexpr (ISO)
            for (int cc = 0; cc < d; ++cc) { const float xc = g * (xr[cc] - Cn[cc]); nx = __builtin_fmaf(xc, xc, nx); }
#pragma unroll
        for (int mm = 0; mm < K2; ++mm) {
            const int c = 2 * mm + h;
            uint4 f = make_uint4(0, 0, 0, 0);
            if (c < d) {
                unsigned x1, x2, x3;
                split3(ISO ? g * (xr[c] - Cn[c]) : g * xr[c], x1, x2, x3);
                f = make_uint4(x1 | (x1 << 16), x2 | (x1 << 16), x2 | (x3 << 16), x2 | (x3 << 16));
            } else if (ISO && c == d) {
                unsigned n1, n2, n3;
                split3(nx, n1, n2, n3);
                f = make_uint4(n1 | (n2 << 16), n3 | (BF16_ONE << 16), BF16_ONE | (BF16_ONE << 16), 0);
            }
            a[mm].u = f;
        }
    }
    const int64_t T0 = (int64_t)blockIdx.y * tchunk;
    const int64_t T1 = (T0 + tchunk < ntile) ? (T0 + tchunk) : ntile;
    const int nt = (int)(T1 - T0);
    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint4* __restrict__ pbase = PB + (T0 * K2) * 64;
    const float4* __restrict__ abase = (const float4*)AP + (T0 * NB) * 64 * 4;      // [tile][block][lane][16 floats]
    struct Tile { Frag f[K2]; float4 w[NB][4]; };
    auto load_tile = [&](int ti, Tile& tl) {
        const int tc = ti < nt ? ti : nt - 1;
#pragma unroll
        for (int mm = 0; mm < K2; ++mm) tl.f[mm].u = pbase[(tc * K2 + mm) * 64 + l];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) tl.w[b][q] = abase[((tc * NB + b) * 64 + l) * 4 + q];
    };
    auto process = [&](const Tile& tl) {
        f32x16 D = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int mm = 0; mm < K2; ++mm) D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tl.f[mm].v, a[mm].v, D, 0, 0, 0);   // transposed tile
        float kv[16];
        if constexpr (fam_is_expr<FAM>) {
            float sv[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) sv[v] = ISO ? fmaxf(D[v], 0.0f) : D[v];
            expr_value_block<float, ISO, 16>(sv, kp, kv);
        } else {
            mfma_profile_block<FAM>(D, kp);
#pragma unroll
            for (int v = 0; v < 16; ++v) kv[v] = D[v];
        }
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float4 wq = tl.w[b][v >> 2];
                const float wv = (v & 3) == 0 ? wq.x : ((v & 3) == 1 ? wq.y : ((v & 3) == 2 ? wq.z : wq.w));
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, kv[v], acc[b], 0, 0, 0);
            }
    };
    Tile t0, t1;
    load_tile(0, t0);
    for (int ti = 0; ti < nt; ti += 2) {
        load_tile(ti + 1, t1);
        process(t0);
        load_tile(ti + 2, t0);
        if (ti + 1 < nt) process(t1);
    }
    // acc[b][v]: lane (i = t, h), right-hand side c = 32 b + 8 (v / 4) + 4 h + v % 4
    const int64_t i = i0 + t;
    if (i >= n) return;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int c = 32 * b + 8 * (v >> 2) + 4 * h + (v & 3);
            if (final_store) {
                if (c < nrhs) {
                    float* yp = out + i + (int64_t)c * ldy;
                    float r = alpha * acc[b][v];
                    if (beta != 0.0f) r = __builtin_fmaf(beta, *yp, r);
                    *yp = r;
                }
            } else {
                out[((int64_t)blockIdx.y * (32 * NB) + c) * npad + i] = acc[b][v];
            }
        }
}

// ---- symmetric Gramians: upper triangle once (design notes: dense_mfma.hip, "Symmetric Gramian") ----------------------------
// FAM = FAM_EQFAST: the EQ form of dense_mfma.hip (exponent straight from the MFMA, norms in the weights); any other family:
// the generic form above (the MFMA yields the profile argument s, the norms ride in a pseudo-coordinate, weights are a_j).
constexpr int FAM_EQFAST = 1000;
constexpr int FAM_EQFAST_H = 1001;      // the same with the fp16 two-way split of the coordinates (one MFMA per four coordinates; dense_mfma.hip, "Which split")
template <int FAM> struct SymParamsOf { using type = typename ParamsOf<FAM, float>::type; };
template <> struct SymParamsOf<FAM_EQFAST> { using type = KParams<float>; };
template <> struct SymParamsOf<FAM_EQFAST_H> { using type = KParams<float>; };

// NW = waves per workgroup = row tiles per panel: 8 (two workgroups per CU = 4 waves per SIMD, 128 VGPRs) or 4 (three workgroups per CU = 3 waves
// per SIMD, 168 VGPRs: the profiles whose tile body needs them — mfma_sym_nw; the tile body itself runs fastest at 3 waves per SIMD,
// profiles/r05_tile_body_probe.txt).  NOT 6: a workgroup's waves go round the CU's four SIMDs, so two 6-wave workgroups load them 4 / 2 / 4 / 2 and
// the barriers pace every wave by the fullest SIMD (measured: MaternP(2) at C2 size 3.35 ms, VALU active 53 % of the time).
// ORD: MaternP's order at compile time (mfma_profile_block)
template <int FAM> constexpr int mfma_sym_nw = (FAM == COVGRAM_MATERNP || FAM == COVGRAM_RQ || FAM == FAM_SUM_ISO) ? 4 : 8;
template <int FAM, int K2, int NW_ = 8, int ORD = 0, int GFMT = 0>
__global__ __launch_bounds__(64 * NW_) __attribute__((amdgpu_waves_per_eu(NW_ == 8 ? 4 : 3, NW_ == 8 ? 4 : 3))) void dense_mfma_sym_kernel(
    const float* __restrict__ X, int64_t n, int32_t d, const uint4* __restrict__ PB, const float* __restrict__ W, int64_t ntile,
    float* __restrict__ R, float* __restrict__ S, int64_t npad, int32_t tchunk, float g, const float* __restrict__ Cn,
    int32_t pfirst, int32_t pstride, const int32_t* __restrict__ wgmap, const typename SymParamsOf<FAM>::type kp,
    const float* __restrict__ EF) {
    // weights: W[j], padded with zeros by the per-MVM pack (generic form), or, when EF is given (EQ form), a_j * EF[j] with
    // EF[j] = exp2(f_j) the fraction factor of the point's half-norm, cached beside the fragments in the points handle (0 for
    // padding), and W = the caller's a itself: no per-MVM pack kernel at all
    // (the guarded form measured 4-5 % faster than a clamped unconditional load here: tools/eq_k2_ab.py, gpurun r2c/r2d vs r2e/r2f)
    auto wt = [&](int64_t j) { return EF ? (j < n ? W[j] * EF[j] : 0.0f) : W[j]; };
    constexpr bool FAST = (FAM == FAM_EQFAST || FAM == FAM_EQFAST_H);
    constexpr int FMT = (FAM == FAM_EQFAST_H || (!FAST && GFMT == 1)) ? 1 : 0;   // fp16 operands: the EQ form's two-way split / the generic form's (gen_row_fragments)
    constexpr bool PK = K2 <= 2 || mfma_pk_sums<FAM, K2>;   // packed fmas for the weighted sums (see process below)
    constexpr bool ISO = FAST || fam_is_iso<FAM>;
    // 8 waves x ONE row tile each (the 16 row weights u cost as many registers as the accumulators: one row tile per wave
    // keeps 4 waves per SIMD); stages of ST = 4 column tiles, fetched by waves 0..3
    constexpr int NW = NW_, ST = 4;
    // Column chunks sit at ABSOLUTE multiples of tchunk (the panel's first one is cut at its own first tile 8 p), so the
    // workgroups in flight — consecutive panels of the same chunk index — walk the same ~1 MB of fragments, which stays in L2
    // (chunks relative to 8 p made every panel's range different: 620 MB of L2 misses per C2 launch instead of ~40).
    // panels pfirst, pfirst + pstride, ...: all of them on one GPU (0, 1); rank g of P GPUs takes (g, P) — cyclic, so that
    // every rank gets the same share of the triangle — and S is indexed by the LOCAL panel number blockIdx.x
    // wgmap[blockIdx.x] = (local panel << 12) | absolute chunk: the host lists only the (panel, chunk) pairs that exist — a
    // rectangular (panel, chunk) grid is half empty, and an empty 512-thread workgroup still waits for a full slot (LDS,
    // registers) in dispatch order before it can exit, which left the chip 40 % idle on short launches (1/8 of C2).
    const int32_t wm = wgmap[blockIdx.x];
    const int64_t lp = wm >> 12;
    const int64_t cabs = wm & 4095;
    const int64_t p = pfirst + (int64_t)pstride * lp;
    const int64_t T1a = (cabs + 1) * tchunk;
    const int64_t T0 = (cabs * tchunk > NW * p) ? cabs * tchunk : NW * p;
    if (T1a <= NW * p || T0 >= ntile) return;                      // (never for a listed pair; whole workgroup, before any barrier)
    const int64_t T1 = T1a < ntile ? T1a : ntile;
    const int nt = (int)(T1 - T0);
    const int l = threadIdx.x & 63, t = l & 31, h = l >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t I0 = NW * p + wv;                                // this wave's row tile
    const int64_t i0 = I0 * 32;
    Frag a[K2];
    float er = 1.0f;                                               // EQ form: exp2 of the fraction of the row's half-norm
    float u[16];                                                   // a_i e_i of the 16 rows this lane's accumulators belong to
    {
        int64_t row = i0 + t;
        if (row >= n) row = n - 1;                                 // clamp: computed, never stored, weight 0 below
        const float* __restrict__ xr = X + row * (int64_t)d;
        float part = 0.0f;
        if constexpr (FAST) {
            er = eq_row_fragments_fmt<K2, FMT>(xr, Cn, d, g, h, a);
        } else {
            gen_row_fragments<K2, GFMT, ISO>(xr, Cn, d, kp.gamma, h, a);
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {                             // MFMA 32x32 output: register v of half h is row 8 (v / 4) + 4 h + v % 4
            int64_t ri = i0 + 8 * (v >> 2) + 4 * h + (v & 3);
            const float keep = ri < n ? 1.0f : 0.0f;
            if (ri >= n) ri = n - 1;
            u[v] = wt(ri) * keep;
        }
    }
    // accumulators and row weights as register PAIRS from the start (round 5): as 16 scalars re-paired around every packed fma they were copied
    // between register sets at every transition between the four copies of the stage loop (16-32 v_mov per stage)
    f32x2 acc2[8], u2[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) { acc2[v] = (f32x2){0.0f, 0.0f}; u2[v] = (f32x2){u[2 * v], u[2 * v + 1]}; }

    const uint4* __restrict__ pbase = PB + (T0 * K2) * 64;
    __shared__ uint4 sfA[ST][K2][64], sfB[ST][K2][64];
    __shared__ float swA[ST][32], swB[ST][32];
    __shared__ float csA[NW][ST][64], csB[NW][ST][64];             // [wave][tile of the stage][half-wave, column]: column sums per half-wave
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int nstage = (nt + ST - 1) / ST;
    float gw = 0.0f;
    // one tile: E once, row sums with the column weight w (masked below the diagonal), column sums with the row weights u
    // (masked: only the stages that touch the panel's own diagonal block carry the two wave-uniform masks)
    auto process = [&](auto masked, const Frag (&f)[K2], float w, int64_t J, float& cpart) {
        constexpr bool MASKED = decltype(masked)::value;
        f32x16 D = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int mm = 0; mm < K2; ++mm) D = eq_mma<FMT>(a[mm], f[mm], D);
        if constexpr (fam_is_expr<FAM>) {                                           // composite: all 16 entries factor by factor
            float sv[16], kv[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) sv[v] = ISO ? fmaxf(D[v], 0.0f) : D[v];
            expr_value_block<float, ISO, 16>(sv, kp, kv);
#pragma unroll
            for (int v = 0; v < 16; ++v) D[v] = kv[v];
        } else {
        if constexpr (FAST) {
#pragma unroll
            for (int v = 0; v < 16; ++v) D[v] = __builtin_amdgcn_exp2f(D[v]);
        } else mfma_profile_block<FAM, ORD>(D, kp);
        }
        const float wr = (!MASKED || J >= I0) ? w : 0.0f;          // wave-uniform masks: only inside the diagonal block
        float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f;
        if constexpr (PK) {
            // two fmas per instruction (v_pk_fma_f32 on register pairs): the same sums in the same order as below.  Beside a RUNNING MFMA a packed fma
            // costs more than the two it replaces (tools/pkfma_probe.hip: 32 of them +260 cycles next to back-to-back MFMAs), so only the
            // kernels with one or two MFMAs per tile take it (general kernel at d = 3: 1.44 -> 1.33 ms; four MFMAs per tile: +2-3 %)
            f32x2 c01 = {0.0f, 0.0f}, c23 = {0.0f, 0.0f};
#pragma unroll
            for (int v = 0; v < 8; v += 2) {
                const f32x2 d01 = {D[2 * v], D[2 * v + 1]}, d23 = {D[2 * v + 2], D[2 * v + 3]};
                acc2[v] = pk_fma((f32x2){wr, wr}, d01, acc2[v]);
                acc2[v + 1] = pk_fma((f32x2){wr, wr}, d23, acc2[v + 1]);
                c01 = pk_fma(u2[v], d01, c01);
                c23 = pk_fma(u2[v + 1], d23, c23);
            }
            c0 = c01[0]; c1 = c01[1]; c2 = c23[0]; c3 = c23[1];
        } else {
#pragma unroll
        for (int v = 0; v < 8; v += 2) {
            acc2[v][0] = __builtin_fmaf(wr, D[2 * v], acc2[v][0]);
            acc2[v][1] = __builtin_fmaf(wr, D[2 * v + 1], acc2[v][1]);
            acc2[v + 1][0] = __builtin_fmaf(wr, D[2 * v + 2], acc2[v + 1][0]);
            acc2[v + 1][1] = __builtin_fmaf(wr, D[2 * v + 3], acc2[v + 1][1]);
            c0 = __builtin_fmaf(u2[v][0], D[2 * v], c0);
            c1 = __builtin_fmaf(u2[v][1], D[2 * v + 1], c1);
            c2 = __builtin_fmaf(u2[v + 1][0], D[2 * v + 2], c2);
            c3 = __builtin_fmaf(u2[v + 1][1], D[2 * v + 3], c3);
        }
        }
        cpart = (!MASKED || J > I0) ? (c0 + c1) + (c2 + c3) : 0.0f;   // this half-wave's 16 rows; the halves meet in the flush
    };
#define CG_DMA(stage, SF)                                                                       \
        if (wv < ST) {                                                                          \
            const int ti_ = (stage) * ST + wv;                                                  \
            const int tc_ = ti_ < nt ? ti_ : nt - 1;                                            \
            _Pragma("unroll") for (int mm = 0; mm < K2; ++mm)                                   \
                __builtin_amdgcn_global_load_lds((gptr_t)(pbase + (tc_ * K2 + mm) * 64 + l), (lptr_t)&SF[wv][mm][0], 16, 0, 0); \
            gw = wt((T0 + tc_) * 32 + t) * (ti_ < nt ? 1.0f : 0.0f);   /* tiles past the chunk: weight 0 */ \
        }
    /* the one-pass Sum takes the stage's tiles one at a time: a short loop body the register allocator handles without scratch (two at a time: 36-52 B) */ \
#define CG_STAGE_M(M_, st_, SF, SW, CS)                                                         \
        if constexpr (FAM == FAM_SUM_ISO || K2 > 4) {   /* (long fragments: two tiles' worth would not fit 128 registers) */ \
            _Pragma("unroll 1") for (int k = 0; k < ST; ++k) {                                  \
                Frag f0[K2];                                                                    \
                _Pragma("unroll") for (int mm = 0; mm < K2; ++mm) f0[mm].u = SF[k][mm][l];      \
                const float w0 = SW[k][t];                                                      \
                float cp0;                                                                      \
                process(std::integral_constant<bool, M_>(), f0, w0, T0 + (int64_t)(st_) * ST + k, cp0); \
                CS[wv][k][l] = cp0;                                                             \
            }                                                                                   \
        } else                                                                                  \
        _Pragma("unroll 1") for (int k = 0; k < ST; k += 2) {                                   \
            Frag f0[K2], f1[K2];                                                                \
            _Pragma("unroll") for (int mm = 0; mm < K2; ++mm) f0[mm].u = SF[k][mm][l];          \
            const float w0 = SW[k][t];                                                          \
            _Pragma("unroll") for (int mm = 0; mm < K2; ++mm) f1[mm].u = SF[k + 1][mm][l];      \
            const float w1 = SW[k + 1][t];                                                      \
            float cp0, cp1;                                                                     \
            process(std::integral_constant<bool, M_>(), f0, w0, T0 + (int64_t)(st_) * ST + k, cp0);     \
            process(std::integral_constant<bool, M_>(), f1, w1, T0 + (int64_t)(st_) * ST + k + 1, cp1); \
            CS[wv][k][l] = cp0; CS[wv][k + 1][l] = cp1;                                         \
        }
    // a stage whose first tile lies beyond the panel's diagonal block (tiles NW p .. NW p + NW - 1) needs no masks
#define CG_STAGE(st_, SF, SW, CS)                                                               \
        if (T0 + (int64_t)(st_) * ST >= NW * p + NW) { CG_STAGE_M(false, st_, SF, SW, CS) }     \
        else { CG_STAGE_M(true, st_, SF, SW, CS) }
    // after the stage's barrier: wave w < ST adds the 8 waves x 2 half-waves' column sums of tile w of that stage (fixed order)
#define CG_FLUSH(st_, CS)                                                                       \
        if (wv < ST) {                                                                          \
            const int64_t J_ = T0 + (int64_t)(st_) * ST + wv;                                   \
            if (h == 0 && J_ < T1) {                                                            \
                float s_ = 0.0f;                                                                \
                _Pragma("unroll") for (int w_ = 0; w_ < NW; ++w_) s_ += CS[w_][wv][t] + CS[w_][wv][32 + t]; \
                S[lp * npad + 32 * J_ + t] = s_;                                                \
            }                                                                                   \
        }
    CG_DMA(0, sfA)
    if (wv < ST && h == 0) swA[wv][t] = gw;
    __syncthreads();
    for (int st = 0; st < nstage; st += 2) {
        CG_DMA(st + 1 < nstage ? st + 1 : st, sfB)                  // past the last stage: a re-fetch nobody reads
        if (st > 0) CG_FLUSH(st - 1, csB)
        CG_STAGE(st, sfA, swA, csA)
        if (wv < ST && h == 0) swB[wv][t] = gw;
        __syncthreads();
        if (st + 1 >= nstage) { CG_FLUSH(st, csA) break; }
        CG_DMA(st + 2 < nstage ? st + 2 : st + 1, sfA)
        CG_FLUSH(st, csA)
        CG_STAGE(st + 1, sfB, swB, csB)
        if (wv < ST && h == 0) swA[wv][t] = gw;
        __syncthreads();
        if (st + 2 >= nstage) { CG_FLUSH(st + 1, csB) }
    }
#undef CG_DMA
#undef CG_STAGE
#undef CG_STAGE_M
#undef CG_FLUSH

    const int vsel = (t & 3) + 4 * (t >> 3);
    float tot = 0.0f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        float s = acc2[v >> 1][v & 1];
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8); s += __shfl_xor(s, 16);
        tot = (vsel == v) ? s : tot;
    }
    const int64_t i = i0 + t;
    if (((t >> 2) & 1) != h || i >= n) return;
    R[cabs * npad + i] = FAST ? er * tot : tot;
}


// ---- symmetric form, long fragments (K2 > 4: d = 9 .. 32): 4 waves x ONE row tile = a 128-row panel; ONE column tile per stage,
// its K2 fragment slices fetched by the four waves in turn (the split-tile staging of dense_mfma_eq_kernel<.., LDS = 2>), one
// barrier per tile.  Same chunks / slabs / masks / workgroup list as dense_mfma_sym_kernel, with 4 tiles per panel.
template <int FAM, int K2, int GFMT = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void dense_mfma_sym_wide_kernel(
    const float* __restrict__ X, int64_t n, int32_t d, const uint4* __restrict__ PB, const float* __restrict__ W, int64_t ntile,
    float* __restrict__ R, float* __restrict__ S, int64_t npad, int32_t tchunk, float g, const float* __restrict__ Cn,
    int32_t pfirst, int32_t pstride, const int32_t* __restrict__ wgmap, const typename SymParamsOf<FAM>::type kp,
    const float* __restrict__ EF) {
    // weights: W[j], padded with zeros by the per-MVM pack (generic form), or, when EF is given (EQ form), a_j * EF[j] with
    // EF[j] = exp2(f_j) the fraction factor of the point's half-norm, cached beside the fragments in the points handle (0 for
    // padding), and W = the caller's a itself: no per-MVM pack kernel at all
    // (the guarded form measured 4-5 % faster than a clamped unconditional load here: tools/eq_k2_ab.py, gpurun r2c/r2d vs r2e/r2f)
    auto wt = [&](int64_t j) { return EF ? (j < n ? W[j] * EF[j] : 0.0f) : W[j]; };
    constexpr bool FAST = (FAM == FAM_EQFAST || FAM == FAM_EQFAST_H);
    constexpr int FMT = (FAM == FAM_EQFAST_H || (!FAST && GFMT == 1)) ? 1 : 0;   // fp16 operands: the EQ form's split / the generic form's (gen_row_fragments)
    constexpr bool ISO = FAST || fam_is_iso<FAM>;
    constexpr int NW = 4;
    const int32_t wm = wgmap[blockIdx.x];
    const int64_t lp = wm >> 12;
    const int64_t cabs = wm & 4095;
    const int64_t p = pfirst + (int64_t)pstride * lp;
    const int64_t T1a = (cabs + 1) * tchunk;
    const int64_t T0 = (cabs * tchunk > NW * p) ? cabs * tchunk : NW * p;
    if (T1a <= NW * p || T0 >= ntile) return;                      // (never for a listed pair; whole workgroup, before any barrier)
    const int64_t T1 = T1a < ntile ? T1a : ntile;
    const int nt = (int)(T1 - T0);
    const int l = threadIdx.x & 63, t = l & 31, h = l >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t I0 = NW * p + wv;
    const int64_t i0 = I0 * 32;
    Frag a[K2];
    float er = 1.0f;                                               // EQ form: exp2 of the fraction of the row's half-norm
    float u[16];
    {
        int64_t row = i0 + t;
        if (row >= n) row = n - 1;
        const float* __restrict__ xr = X + row * (int64_t)d;
        float part = 0.0f;
        if constexpr (FAST) {
            er = eq_row_fragments_fmt<K2, FMT>(xr, Cn, d, g, h, a);
        } else {
            gen_row_fragments<K2, GFMT, ISO>(xr, Cn, d, kp.gamma, h, a);
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            int64_t ri = i0 + 8 * (v >> 2) + 4 * h + (v & 3);
            const float keep = ri < n ? 1.0f : 0.0f;
            if (ri >= n) ri = n - 1;
            u[v] = wt(ri) * keep;
        }
    }
    float acc[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.0f;

    const uint4* __restrict__ pbase = PB + (T0 * K2) * 64;
    __shared__ uint4 tfA[K2][64], tfB[K2][64];
    __shared__ float twA[32], twB[32];
    __shared__ float csA[NW][64], csB[NW][64];
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    float gw = 0.0f, ge = 1.0f, gmask = 0.0f;
#define CG_GW (FAST ? gw * ge * gmask : gw * gmask)
#define CG_DMA2(tile, TF)                                                                       \
        {                                                                                       \
            const int ti_ = (tile);                                                             \
            const int tc_ = ti_ < nt ? ti_ : nt - 1;                                            \
            _Pragma("unroll") for (int q = 0; q < (K2 + NW - 1) / NW; ++q) {                    \
                const int mm = wv + q * NW;                                                     \
                if (mm < K2) __builtin_amdgcn_global_load_lds((gptr_t)(pbase + (tc_ * K2 + mm) * 64 + l), (lptr_t)&TF[mm][0], 16, 0, 0); \
            }                                                                                   \
            /* the next tile's column weights are LOADED here and combined (a_j e_j, masks) only where they are stored, after the tile's arithmetic: any \
               use here puts the s_waitcnt vmcnt(0) for them — and for the LDS-DMA beside them — in front of the tile, once per TILE in this kernel */ \
            if (wv == 0) {                                                                      \
                const int64_t j_ = (T0 + tc_) * 32 + t;                                         \
                gmask = (ti_ < nt && (!FAST || j_ < n)) ? 1.0f : 0.0f;                          \
                if constexpr (FAST) { const int64_t jc_ = j_ < n ? j_ : n - 1; gw = W[jc_]; ge = EF[jc_]; } \
                else gw = W[j_];                                                                \
            }                                                                                   \
        }
#define CG_TILE2(ti_, TF, TW, CS)                                                               \
        {                                                                                       \
            const int64_t J = T0 + (ti_);                                                       \
            f32x16 D = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                        \
            Frag f_[K2];                                                                        \
            _Pragma("unroll") for (int mm = 0; mm < K2; ++mm) f_[mm].u = TF[mm][l];             \
            const float w = TW[t];                                                              \
            _Pragma("unroll") for (int mm = 0; mm < K2; ++mm) D = eq_mma<FMT>(a[mm], f_[mm], D); \
            if constexpr (fam_is_expr<FAM>) {                                                   \
                float sv[16], kv[16];                                                           \
                _Pragma("unroll") for (int v = 0; v < 16; ++v) sv[v] = ISO ? fmaxf(D[v], 0.0f) : D[v]; \
                expr_value_block<float, ISO, 16>(sv, kp, kv);                                   \
                _Pragma("unroll") for (int v = 0; v < 16; ++v) D[v] = kv[v];                    \
            } else {                                                                            \
            if constexpr (FAST) { _Pragma("unroll") for (int v = 0; v < 16; ++v) D[v] = __builtin_amdgcn_exp2f(D[v]); } \
            else mfma_profile_block<FAM>(D, kp);                                                \
            }                                                                                   \
            const float wr = (J >= I0) ? w : 0.0f;                                              \
            float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f;                                   \
            if constexpr (mfma_pk_sums<FAM, K2>) {                                              \
                f32x2 c01 = {0.0f, 0.0f}, c23 = {0.0f, 0.0f};                                   \
                _Pragma("unroll") for (int v = 0; v < 16; v += 4) {                             \
                    f32x2 a01 = {acc[v], acc[v + 1]}, a23 = {acc[v + 2], acc[v + 3]};           \
                    const f32x2 d01 = {D[v], D[v + 1]}, d23 = {D[v + 2], D[v + 3]};             \
                    a01 = pk_fma(pk_splat(wr), d01, a01);                                       \
                    a23 = pk_fma(pk_splat(wr), d23, a23);                                       \
                    acc[v] = a01[0]; acc[v + 1] = a01[1]; acc[v + 2] = a23[0]; acc[v + 3] = a23[1]; \
                    c01 = pk_fma((f32x2){u[v], u[v + 1]}, d01, c01);                            \
                    c23 = pk_fma((f32x2){u[v + 2], u[v + 3]}, d23, c23);                        \
                }                                                                               \
                c0 = c01[0]; c1 = c01[1]; c2 = c23[0]; c3 = c23[1];                             \
            } else                                                                              \
            _Pragma("unroll") for (int v = 0; v < 16; v += 4) {                                 \
                acc[v] = __builtin_fmaf(wr, D[v], acc[v]);                                      \
                acc[v + 1] = __builtin_fmaf(wr, D[v + 1], acc[v + 1]);                          \
                acc[v + 2] = __builtin_fmaf(wr, D[v + 2], acc[v + 2]);                          \
                acc[v + 3] = __builtin_fmaf(wr, D[v + 3], acc[v + 3]);                          \
                c0 = __builtin_fmaf(u[v], D[v], c0);                                            \
                c1 = __builtin_fmaf(u[v + 1], D[v + 1], c1);                                    \
                c2 = __builtin_fmaf(u[v + 2], D[v + 2], c2);                                    \
                c3 = __builtin_fmaf(u[v + 3], D[v + 3], c3);                                    \
            }                                                                                   \
            CS[wv][l] = (J > I0) ? (c0 + c1) + (c2 + c3) : 0.0f;                                \
        }
    // after the tile's barrier: one wave (they take turns) adds the 4 waves x 2 half-waves' column sums in fixed order
#define CG_FLUSH2(ti_, CS)                                                                      \
        if (wv == ((ti_) & (NW - 1)) && h == 0 && T0 + (ti_) < T1) {                            \
            float s_ = 0.0f;                                                                    \
            _Pragma("unroll") for (int w_ = 0; w_ < NW; ++w_) s_ += CS[w_][t] + CS[w_][32 + t]; \
            S[lp * npad + 32 * (T0 + (ti_)) + t] = s_;                                          \
        }
    CG_DMA2(0, tfA)
    if (wv == 0 && h == 0) twA[t] = CG_GW;
    __syncthreads();
    for (int ti = 0; ti < nt; ti += 2) {
        CG_DMA2(ti + 1, tfB)
        if (ti > 0) CG_FLUSH2(ti - 1, csB)
        CG_TILE2(ti, tfA, twA, csA)
        if (wv == 0 && h == 0) twB[t] = CG_GW;
        __syncthreads();
        if (ti + 1 >= nt) { CG_FLUSH2(ti, csA) break; }
        CG_DMA2(ti + 2, tfA)
        CG_FLUSH2(ti, csA)
        CG_TILE2(ti + 1, tfB, twB, csB)
        if (wv == 0 && h == 0) twA[t] = CG_GW;
        __syncthreads();
        if (ti + 2 >= nt) { CG_FLUSH2(ti + 1, csB) }
    }
#undef CG_DMA2
#undef CG_TILE2
#undef CG_FLUSH2
#undef CG_GW

    const int vsel = (t & 3) + 4 * (t >> 3);
    float tot = 0.0f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        float s = acc[v];
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8); s += __shfl_xor(s, 16);
        tot = (vsel == v) ? s : tot;
    }
    const int64_t i = i0 + t;
    if (((t >> 2) & 1) != h || i >= n) return;
    R[cabs * npad + i] = FAST ? er * tot : tot;
}

}  // namespace covgram
#include "dense_mfma_sym2.hpp"
namespace covgram {

struct MfmaArgs {
    const float* X; int64_t n; int32_t d;
    const uint4* PB; const float* W; int64_t ntile;
    float* out; int64_t npad; int64_t ldy; int32_t nrhs;
    int64_t tchunk; float alpha, beta; int32_t final_store;
    int32_t K2, RT, NR;
    const HostKernel* hk;
    hipStream_t stream;
    dim3 grid;
    const float* Cn = nullptr;   // common centre of isotropic kernels (dense_mvm.hpp)
    // symmetric form (dense_mfma_sym_kernel): row-sum slab R, column-sum slab S, the explicit workgroup list
    int32_t fmt = 0;             // 1: the generic kernels' fp16 two-way split (gen_row_fragments: GFMT = 1; isotropic profiles, K2 <= 4)
    int32_t lds = 0;             // 1: dense_mfma_gen_kernel<.., LDS = 2> (grid.x counts workgroups of 4 waves)
    int32_t mrhs = 0;            // 1 / 2: dense_mfma_mrhs_kernel with that many blocks of 32 right-hand sides (W = the packed A operands)
    int32_t sym = 0;
    int32_t sym_rt = 1;          // 2: dense_mfma_sym2_kernel (two row tiles per wave, 8-tile panels) where the family has the instance (mfma_sym2_has)
    float* R = nullptr; float* S = nullptr;
    const int32_t* wgmap = nullptr;
    int32_t pfirst = 0, pstride = 1;
};

// returns the resident blocks per CU of the instance when `query` is set (no launch), COVGRAM_OK / error otherwise
template <int FAM, int K2, int RT, int NR, int ORD, int GFMT>
static void mfma_gen_launch_f(const MfmaArgs& a) {
    if (a.lds)
        hipLaunchKernelGGL((dense_mfma_gen_kernel<FAM, K2, RT, NR, 2, ORD, GFMT>), dim3((a.grid.x + 3) / 4, a.grid.y), dim3(256), 0, a.stream, a.X, a.n, a.d,
                           a.PB, a.W, a.ntile, a.out, a.npad, a.ldy, a.nrhs, a.tchunk, a.alpha, a.beta, a.final_store, a.Cn,
                           make_params<FAM, float>(*a.hk));
    else
        hipLaunchKernelGGL((dense_mfma_gen_kernel<FAM, K2, RT, NR, 0, ORD, GFMT>), a.grid, dim3(64), 0, a.stream, a.X, a.n, a.d, a.PB, a.W, a.ntile, a.out,
                           a.npad, a.ldy, a.nrhs, a.tchunk, a.alpha, a.beta, a.final_store, a.Cn, make_params<FAM, float>(*a.hk));
}
// the fp16 split's instances exist for the isotropic families up to eight MFMAs per tile (d <= 30) — the host asks for nothing else
template <int FAM, int K2> constexpr bool mfma_gen_has_f16 = fam_is_iso<FAM> && !fam_is_expr<FAM> && K2 <= 8;   // d + 2 positions of four per MFMA: d <= 30
template <int FAM, int K2, int RT, int NR, int ORD>
static void mfma_gen_launch(const MfmaArgs& a) {
    if constexpr (mfma_gen_has_f16<FAM, K2>) { if (a.fmt == 1) { mfma_gen_launch_f<FAM, K2, RT, NR, ORD, 1>(a); return; } }
    mfma_gen_launch_f<FAM, K2, RT, NR, ORD, 0>(a);
}
template <int FAM, int K2, int RT, int NR>
static int mfma_gen_one(const MfmaArgs& a, bool query) {
    if (query) {
        static int cached = 0;
        if (!cached) {
            int nb = 0;
            constexpr int QORD = (NR == 1 && (FAM == COVGRAM_MATERNP || FAM == FAM_SUM_ISO)) ? 2 : 0;   // the instance most launches take
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, dense_mfma_gen_kernel<FAM, K2, RT, NR, 0, QORD, 0>, 64, 0) != hipSuccess || nb <= 0) nb = 16;
            cached = nb;
        }
        return cached;
    }
    // MaternP(1), MaternP(2) (Matern 3/2, 5/2) with one right-hand side: the order is a compile-time constant of the instance; a Sum: its number of terms
    if constexpr (FAM == COVGRAM_MATERNP && NR == 1) {
        if (a.hk->k.p == 2 && a.hk->k.power == 1) { mfma_gen_launch<FAM, K2, RT, NR, 2>(a); return COVGRAM_OK; }
        if (a.hk->k.p == 1 && a.hk->k.power == 1) { mfma_gen_launch<FAM, K2, RT, NR, 1>(a); return COVGRAM_OK; }
    }
    if constexpr (FAM == FAM_SUM_ISO && NR == 1) {
        if (a.hk->nterms == 2) mfma_gen_launch<FAM, K2, RT, NR, 2>(a); else mfma_gen_launch<FAM, K2, RT, NR, 3>(a);
        return COVGRAM_OK;
    }
    mfma_gen_launch<FAM, K2, RT, NR, 0>(a);
    return COVGRAM_OK;
}

// the two-row-tile symmetric kernel's generic instances: isotropic single profiles at one or two MFMAs per tile (d <= 3 bf16, d <= 6 fp16).  NOT the
// one-pass Sum: two row tiles of its term-by-term body spill into the tile loop (measured 6.3 against 4.0 ms for three terms, tools/gen_sym_rt_ab.py)
template <int FAM, int K2> constexpr bool mfma_sym2_has = fam_is_iso<FAM> && !fam_is_expr<FAM> && FAM != FAM_SUM_ISO && K2 <= 2;
inline bool mfma_sym2_family(int launcher_family) {
    switch (launcher_family) { case COVGRAM_EQ: case COVGRAM_RQ: case COVGRAM_CAUCHY: case COVGRAM_IMQ: case COVGRAM_MATERNP: return true; default: return false; }
}
template <int FAM, int K2, int ORD>
static void mfma_sym_narrow(const MfmaArgs& a) {
    constexpr int NW = mfma_sym_nw<FAM>;
    if constexpr (mfma_sym2_has<FAM, K2>) {
        if (a.sym_rt == 2) {
            if (a.fmt == 1)
                hipLaunchKernelGGL((dense_mfma_sym2_kernel<FAM, K2, 8, ORD, 1>), a.grid, dim3(256), 0, a.stream, a.X, a.n, a.d, a.PB, a.W, a.ntile, a.R, a.S, a.npad,
                                   (int32_t)a.tchunk, 0.0f, a.Cn, a.pfirst, a.pstride, a.wgmap, make_params<FAM, float>(*a.hk), (const float*)nullptr);
            else
                hipLaunchKernelGGL((dense_mfma_sym2_kernel<FAM, K2, 8, ORD, 0>), a.grid, dim3(256), 0, a.stream, a.X, a.n, a.d, a.PB, a.W, a.ntile, a.R, a.S, a.npad,
                                   (int32_t)a.tchunk, 0.0f, a.Cn, a.pfirst, a.pstride, a.wgmap, make_params<FAM, float>(*a.hk), (const float*)nullptr);
            return;
        }
    }
    if constexpr (mfma_gen_has_f16<FAM, K2>) {
        if (a.fmt == 1) {
            hipLaunchKernelGGL((dense_mfma_sym_kernel<FAM, K2, NW, ORD, 1>), a.grid, dim3(64 * NW), 0, a.stream, a.X, a.n, a.d, a.PB, a.W, a.ntile, a.R, a.S, a.npad,
                               (int32_t)a.tchunk, 0.0f, a.Cn, a.pfirst, a.pstride, a.wgmap, make_params<FAM, float>(*a.hk), (const float*)nullptr);
            return;
        }
    }
    hipLaunchKernelGGL((dense_mfma_sym_kernel<FAM, K2, NW, ORD, 0>), a.grid, dim3(64 * NW), 0, a.stream, a.X, a.n, a.d, a.PB, a.W, a.ntile, a.R, a.S, a.npad,
                       (int32_t)a.tchunk, 0.0f, a.Cn, a.pfirst, a.pstride, a.wgmap, make_params<FAM, float>(*a.hk), (const float*)nullptr);
}
// which symmetric form serves (family, K2): the 8- or 6-wave panels with stages of four tiles (narrow), or four waves and one tile per stage
// (wide).  The one-pass Sum takes the wide form from K2 = 3 on (its 6-wave instances spill 12-40 B at K2 = 3, 4) and has no instance beyond K2 = 8.
// (round 5: six MFMAs per tile too for the 8-wave families — Cauchy, IMQ, EQ^p, dot products —, one tile at a time in registers; as the EQ form, dense_mfma.hip)
template <int FAM> constexpr int mfma_sym_narrow_maxk2 = FAM == FAM_SUM_ISO ? 2 : ((mfma_sym_nw<FAM> == 8 && !fam_is_expr<FAM>) ? 6 : MFMA_NARROW_MAXK2);
inline int mfma_sym_tiles_per_panel(int launcher_family, int k2, int sym_rt = 1) {
    if (sym_rt == 2 && k2 <= 2 && mfma_sym2_family(launcher_family)) return 8;   // dense_mfma_sym2_kernel: 4 waves x 2 row tiles
    const bool heavy = launcher_family == COVGRAM_MATERNP || launcher_family == COVGRAM_RQ || launcher_family == FAM_SUM_ISO;
    const bool expr = launcher_family == FAM_EXPR_ISO || launcher_family == FAM_EXPR_DOT;
    const int narrow_max = launcher_family == FAM_SUM_ISO ? 2 : ((heavy || expr) ? MFMA_NARROW_MAXK2 : 6);
    return (k2 > narrow_max || heavy) ? 4 : 8;
}
template <int FAM, int K2>
static int mfma_sym_one(const MfmaArgs& a) {
    if constexpr (FAM == FAM_SUM_ISO && K2 > 8) {
        set_error("dense_mfma_sym: the one-pass Sum has no instance at K2 = %d", K2); return COVGRAM_EUNSUPPORTED;
    } else
    if constexpr (K2 <= mfma_sym_narrow_maxk2<FAM>) {
        if constexpr (FAM == COVGRAM_MATERNP) {            // the order is a compile-time constant of the instance
            switch (a.hk->k.power == 1 ? a.hk->k.p : 0) {
                case 1: mfma_sym_narrow<FAM, K2, 1>(a); break;
                case 2: mfma_sym_narrow<FAM, K2, 2>(a); break;
                case 3: mfma_sym_narrow<FAM, K2, 3>(a); break;
                default: mfma_sym_narrow<FAM, K2, 0>(a); break;
            }
        } else if constexpr (FAM == FAM_SUM_ISO) {         // ... and so is a Sum's number of terms
            if (a.hk->nterms == 2) mfma_sym_narrow<FAM, K2, 2>(a); else mfma_sym_narrow<FAM, K2, 3>(a);
        } else mfma_sym_narrow<FAM, K2, 0>(a);
    } else {
        if constexpr (mfma_gen_has_f16<FAM, K2>) {
            if (a.fmt == 1) {
                hipLaunchKernelGGL((dense_mfma_sym_wide_kernel<FAM, K2, 1>), a.grid, dim3(256), 0, a.stream, a.X, a.n, a.d, a.PB, a.W, a.ntile, a.R, a.S,
                                   a.npad, (int32_t)a.tchunk, 0.0f, a.Cn, a.pfirst, a.pstride, a.wgmap, make_params<FAM, float>(*a.hk), (const float*)nullptr);
                return COVGRAM_OK;
            }
        }
        hipLaunchKernelGGL((dense_mfma_sym_wide_kernel<FAM, K2, 0>), a.grid, dim3(256), 0, a.stream, a.X, a.n, a.d, a.PB, a.W, a.ntile, a.R, a.S,
                           a.npad, (int32_t)a.tchunk, 0.0f, a.Cn, a.pfirst, a.pstride, a.wgmap, make_params<FAM, float>(*a.hk), (const float*)nullptr);
    }
    return COVGRAM_OK;
}

template <int FAM, int K2, int NB>
static int mfma_mrhs_one(const MfmaArgs& a) {
    hipLaunchKernelGGL((dense_mfma_mrhs_kernel<FAM, K2, NB>), a.grid, dim3(64), 0, a.stream, a.X, a.n, a.d, a.PB, a.W, a.ntile, a.out, a.npad,
                       a.ldy, a.nrhs, a.tchunk, a.alpha, a.beta, a.final_store, a.Cn, make_params<FAM, float>(*a.hk));
    return COVGRAM_OK;
}

template <int FAM, int K2>
static int mfma_gen_K(const MfmaArgs& a, bool query) {
    if (a.sym) return mfma_sym_one<FAM, K2>(a);
    if (a.mrhs == 1) return mfma_mrhs_one<FAM, K2, 1>(a);
    if (a.mrhs == 2) return mfma_mrhs_one<FAM, K2, 2>(a);
    if (a.NR == 4) return mfma_gen_one<FAM, K2, 1, 4>(a, query);
    if constexpr (K2 <= 4) { if (a.RT == 2) return mfma_gen_one<FAM, K2, 2, 1>(a, query); }
    return mfma_gen_one<FAM, K2, 1, 1>(a, query);
}

template <int FAM>
int launch_mfma_family(const MfmaArgs& a, bool query) {
    switch (a.K2) {
        case 1: return mfma_gen_K<FAM, 1>(a, query);
        case 2: return mfma_gen_K<FAM, 2>(a, query);
        case 3: return mfma_gen_K<FAM, 3>(a, query);
        case 4: return mfma_gen_K<FAM, 4>(a, query);
        case 6: return mfma_gen_K<FAM, 6>(a, query);
        case 8: return mfma_gen_K<FAM, 8>(a, query);
        case 12: return mfma_gen_K<FAM, 12>(a, query);
        case 16: return mfma_gen_K<FAM, 16>(a, query);
        default: set_error("dense_mfma: K2 = %d not compiled", a.K2); return COVGRAM_EUNSUPPORTED;
    }
}

typedef int (*mfma_launch_fn)(const MfmaArgs&, bool query);
mfma_launch_fn mfma_launcher(int family);

}  // namespace covgram
