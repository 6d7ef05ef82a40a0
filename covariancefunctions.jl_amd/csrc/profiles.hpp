// profiles.hpp — device functors for the scalar kernel profiles phi(s) and (phi, phi', phi'').
// One functor per covgram_family; T = float uses bare gfx950 transcendental instructions
// (v_exp_f32, v_log_f32, v_sqrt_f32, v_rcp_f32, v_rsq_f32 — 1 ulp), T = double uses the device libm.
// Reference definitions: src/stationary.jl:42,53,60,71,132-158,224,235; src/mercer.jl:9,22;
// src/algebra.jl:61-62 (Power); src/transformation.jl:19 (Lengthscale, folded into the coordinate
// pre-scale gamma = 1/l by the host); derivatives: closed forms of src/gradient.jl:584-600.
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "pack.hpp"
#include "exp2_table.hpp"

namespace covgram {

// ---- math shims --------------------------------------------------------------------------------
__device__ __forceinline__ float cg_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ double cg_exp2(double x) { return exp2(x); }
__device__ __forceinline__ float cg_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ double cg_exp(double x) { return exp(x); }
__device__ __forceinline__ float cg_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ double cg_log2(double x) { return log2(x); }
__device__ __forceinline__ float cg_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
// fp64 square root, reciprocal and reciprocal square root WITHOUT the library's range scaling and correct-rounding steps (hipcc's
// sqrt is 18 instructions: scale test + two ldexp around v_rsq_f64 and three refinements; 1.0 / x is the IEEE division sequence): the
// hardware seed (v_rsq_f64 / v_rcp_f64, ~2^-23 relative, denormal arguments accepted) and a coupled Goldschmidt / cubic Newton
// refinement to <= 1 ulp + one rounding, then ONE class test that hands back the seed's own answer where the refinement cannot be
// formed (0, inf, NaN; a reciprocal that overflowed).  Arguments are squared distances / radii here: >= 0 or NaN.
__device__ __forceinline__ double cg_sqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);                       // x = 0: inf, x = inf: 0, x < 0 or NaN: NaN
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);                                    // sqrt(x) to ~2^-45
    h = __builtin_fma(h, r, h);                                    // 1 / (2 sqrt(x))
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    return __builtin_amdgcn_class(y, 0x264) ? x : g;               // seed inf (x = 0) or 0 (x = inf): the argument itself (g is NaN there)
}
// the same for a STRICTLY POSITIVE argument (or NaN / inf, which come out as NaN): no class test.  For callers that add a tiny constant
// to a sum of squares and whose value at inf is NaN anyway (MaternP: q(inf) exp(-inf) = inf * 0, as in the reference)
__device__ __forceinline__ float cg_sqrt_pos(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ double cg_sqrt_pos(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    return __builtin_fma(__builtin_fma(-g, g, x), h, g);
}
__device__ __forceinline__ float cg_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double cg_rcp(double x) {
    const double y = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, y, 1.0);                    // 2^-23
    const double r = __builtin_fma(y, __builtin_fma(e, e, e), y);  // y (1 + e + e^2): error e^3
    return __builtin_amdgcn_class(y, 0x267) ? y : r;               // seed 0 / inf / NaN (x = inf, 0, a denormal whose reciprocal overflows, NaN)
}
__device__ __forceinline__ float cg_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ double cg_rsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * y), y, 1.0);              // 1 - x y^2, 2^-22
    const double r = __builtin_fma(y * e, __builtin_fma(e, 0.375, 0.5), y);   // y (1 + e/2 + 3 e^2 / 8): error ~e^3
    return __builtin_amdgcn_class(y, 0x267) ? y : r;
}
// u^e for u > 0 (u == 0 handled by callers where it can occur)
__device__ __forceinline__ float cg_pow(float u, float e) { return cg_exp2(e * cg_log2(u)); }
__device__ __forceinline__ double cg_pow(double u, double e) { return pow(u, e); }
template <typename T>
__device__ __forceinline__ T cg_fma(T a, T b, T c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float cg_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <typename T>
__device__ __forceinline__ T horner(const T* h, int deg, T r) {
    T acc = h[deg];
    for (int m = deg - 1; m >= 0; --m) acc = cg_fma(acc, r, h[m]);
    return acc;
}

// degree-3 Horner on a table that is zero beyond its true degree
template <typename T>
__device__ __forceinline__ T horner3(const T* h, T r) {
    return cg_fma(cg_fma(cg_fma(h[3], r, h[2]), r, h[1]), r, h[0]);
}

// integer power by repeated multiplication (uniform exponent)
template <typename T>
__device__ __forceinline__ T ipow(T v, int q) {
    T r = v;
    for (int i = 1; i < q; ++i) r *= v;
    return r;
}

// ---- value functors: s is already divided by l^2 (and, for folded EQ, times log2(e)/2) ---------
template <int FAM, typename T, bool FOLDED>
struct Phi;

// exp2(s c) in fp64 for s >= 0 and a negative constant c = chi + clo (NaN propagates, s = inf gives 0): exp(-s/2) of the EQ
// profile and exp(-r) of the Matern / exponential profiles in the gradient and direct-difference kernels.  The library exp /
// exp2 cost 34 instructions there (hipcc expands each Horner step of their polynomial into v_mov_b64 + v_fmac_f64, both ends of
// the range are checked, and in the gradient kernel the factor kp.c0 was re-read from the kernarg segment inside the column
// loop — an s_load whose lgkmcnt(0) also cut the software-pipelined record stream short); this is 23: n = rint(s chi),
// r = fma(s, chi, -n) + s clo (the product never rounded), y = r ln 2, a degree-11 polynomial for exp(y) on |y| <= ln(2)/2
// (interpolation at Chebyshev nodes computed with mpmath, max relative error 1.7e-17 before rounding; c0 = c1 = 1 exactly) as
// three-address v_fma_f64, ldexp, one underflow select.
__device__ __forceinline__ double exp2_scaled_nonpos(double s, double chi, double clo) {
    const double x = s * chi;
    const double n = __builtin_rint(x);
    const double y = __builtin_fma(s, clo, __builtin_fma(s, chi, -n)) * 0.69314718055994530942;
    // a Horner step as ONE three-address v_fma_f64 (the compiler's two-address form copies the coefficient first)
    auto step = [](double q, double yy, double c) { double r; asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(q), "v"(yy), "v"(c)); return r; };
    double p = 0x1.af631d0059becp-26;
    p = __builtin_fma(p, y, 0x1.28b4057f44145p-22);
    p = step(p, y, 0x1.71ddf5749d126p-19);
    p = step(p, y, 0x1.a01991ac8730ap-16);
    p = step(p, y, 0x1.a01a01b14378fp-13);
    p = step(p, y, 0x1.6c16c187fbe02p-10);
    p = step(p, y, 0x1.111111110f225p-7);
    p = step(p, y, 0x1.555555554f0cfp-5);
    p = step(p, y, 0x1.555555555555ap-3);
    p = step(p, y, 0x1.0000000000011p-1);
    p = __builtin_fma(p, y, 1.0);
    p = __builtin_fma(p, y, 1.0);
    const double e = __builtin_ldexp(p, (int)n);
    return x < -1100.0 ? 0.0 : e;
}
// exp2(-t) in fp64 for t >= 0 in the dense kernels' Matern profile, whose polynomial factor carries NaN and inf (q(NaN) exp2(..) = NaN,
// q(inf) * 0 = NaN as in the reference's (1 + r + ...) exp(-r)): the argument is CLAMPED at 1100 (one v_min_f64; a NaN t gives 0 here)
// instead of the library's compare + two selects on the result, ln 2 is one constant (|f| <= 1/2 is exact, so the product's rounding is
// 2^-54 absolute), and the Horner steps are plain fmas so that the coefficients stay in SGPRs in kernels that have them to spare:
// 17 instructions against the library's 21.  Same polynomial as exp2_scaled_nonpos.
__device__ __forceinline__ double exp2_neg_clamped(double t) {
    const double tc = __builtin_fmin(t, 1100.0);
    const double n = __builtin_rint(-tc);
    const double y = (-tc - n) * 0.69314718055994530942;
    double p = 0x1.af631d0059becp-26;
    p = __builtin_fma(p, y, 0x1.28b4057f44145p-22);
    p = __builtin_fma(p, y, 0x1.71ddf5749d126p-19);
    p = __builtin_fma(p, y, 0x1.a01991ac8730ap-16);
    p = __builtin_fma(p, y, 0x1.a01a01b14378fp-13);
    p = __builtin_fma(p, y, 0x1.6c16c187fbe02p-10);
    p = __builtin_fma(p, y, 0x1.111111110f225p-7);
    p = __builtin_fma(p, y, 0x1.555555554f0cfp-5);
    p = __builtin_fma(p, y, 0x1.555555555555ap-3);
    p = __builtin_fma(p, y, 0x1.0000000000011p-1);
    p = __builtin_fma(p, y, 1.0);
    p = __builtin_fma(p, y, 1.0);
    return __builtin_ldexp(p, (int)n);
}
__device__ __forceinline__ float exp2_neg_clamped(float t) { return __builtin_amdgcn_exp2f(-t); }
// exp2(-t) in fp64 for t >= 0 by table: -256 t = n + r with n an integer and |r| <= 1/2, exp2(-t) = 2^(n >> 8) T[n & 255] (1 + p(r)),
// T[j] = 2^(j/256) correctly rounded (exp2_table.hpp: 2 KB, read per lane — L1-resident), p the degree-4 Taylor polynomial of
// exp(r ln2 / 256) - 1 (0.17 ulp on the interval).  n comes out of the low word of fma(t, -256, 1.5 * 2^52), so the reduction is
// three instructions and exact (the same product is rounded once to an integer and once not at all).  NaN propagates (the clamp only
// replaces the high word of t > 1100, by that of 1100: inf and huge t give 2^-1100 = 0).  14 instructions + one load against the
// 21 of the library's exp2 and the 17 of exp2_neg_clamped; <= 1.3 ulp (table 0.5, final fma 0.5, polynomial 0.2, its evaluation).
// `tab`: the table in global memory (read through L1) or a copy in LDS (exp_tab_lds, for kernels that fill it: a per-lane gather of
// 8-byte entries from 16 cache lines costs the CU's one texture unit ~16 clocks a wave — the bound of the dense fp64 loop once its
// arithmetic was down to 20 instructions per pair; the same gather from LDS is a few clocks)
__device__ __forceinline__ double* exp_tab_lds() { __shared__ double t[256]; return t; }
// (a kernel that evaluates through the LDS copy calls this first, all threads)
__device__ __forceinline__ void exp_tab_lds_fill() {
    double* t = exp_tab_lds();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) t[i] = EXP2_TAB256[i];
    __syncthreads();
}
template <bool KEEP_NAN = true>
__device__ __forceinline__ double exp2_neg_tab(double t, const double* __restrict__ tab = EXP2_TAB256) {
    // KEEP_NAN = false: one v_min_f64 (a NaN argument gives 0) for callers whose NaN travels in another factor (MaternP: q(NaN) * 0 = NaN)
    const double tc = KEEP_NAN ? __hiloint2double(t > 1100.0 ? 0x40913000 : __double2hiint(t), __double2loint(t)) : __builtin_fmin(t, 1100.0);
    // opaque registers: with literals the compiler forms a two-address v_fmac and re-materialises the addend per call (two v_mov)
    double magic = 0x1.8p52, c256 = -256.0;
    asm("" : "+v"(magic));
    asm("" : "+s"(c256));
    const double nb = __builtin_fma(tc, c256, magic);
    const int ni = __double2loint(nb);                            // round(-256 t), two's complement
    const double n = nb - magic;
    const double r = __builtin_fma(tc, c256, -n);
    const double tj = tab[ni & 255];
    double q = __builtin_fma(r, 0x1.3b2ab6fba4e77p-39, 0x1.c6b08d704a0c0p-29);
    q = __builtin_fma(q, r, 0x1.ebfbdff82c58fp-19);
    q = __builtin_fma(q, r, 0x1.62e42fefa39efp-9);
    return __builtin_ldexp(__builtin_fma(tj, r * q, tj), ni >> 8);
}
template <bool KEEP_NAN = true>
__device__ __forceinline__ float exp2_neg_tab(float t, const double* = nullptr) { return __builtin_amdgcn_exp2f(-t); }
// exp(-r) in fp64 for r >= 0 on the same table: -256 log2(e) = chi + clo, n from the low word of fma(r, chi, 1.5 * 2^52), the
// reduced argument fma(r, chi, -n) + r clo (the product never rounded).  The clamp is at r = 763 (exp(-763) < 2^-1100 = 0).
__device__ __forceinline__ double exp_neg_tab(double r, const double* __restrict__ tab = EXP2_TAB256) {
    const double rc = __hiloint2double(r > 763.0 ? 0x4087D800 : __double2hiint(r), __double2loint(r));
    const double clo = -0x1.777d0ffda0d24p-48;
    double magic = 0x1.8p52, chi = -0x1.71547652b82fep+8;
    asm("" : "+v"(magic));
    asm("" : "+s"(chi));
    const double nb = __builtin_fma(rc, chi, magic);
    const int ni = __double2loint(nb);
    const double n = nb - magic;
    const double f = __builtin_fma(rc, clo, __builtin_fma(rc, chi, -n));
    const double tj = tab[ni & 255];
    double q = __builtin_fma(f, 0x1.3b2ab6fba4e77p-39, 0x1.c6b08d704a0c0p-29);
    q = __builtin_fma(q, f, 0x1.ebfbdff82c58fp-19);
    q = __builtin_fma(q, f, 0x1.62e42fefa39efp-9);
    return __builtin_ldexp(__builtin_fma(tj, f * q, tj), ni >> 8);
}
__device__ __forceinline__ float exp_neg_tab(float r, const double* = nullptr) { return __builtin_amdgcn_exp2f(r * -1.44269504088896340736f); }
// u^(-a) in fp64 for u >= 1, a > 0 (the rational-quadratic profile: u = 1 + s / (2 alpha); NaN propagates, u = inf gives 0).  The
// library pow is a general function (sign / zero / infinity cases, ~150 instructions with hipcc's mov + fmac Horner steps); here
// log2(u) = e + 2 z q(z^2) / ln 2 with u = m 2^e, m in [1/sqrt 2, sqrt 2), z = (m - 1) / (m + 1) (|z| <= 0.1716; no cancellation as
// u -> 1: m - 1 is exact) and q a degree-7 polynomial for atanh(z) / z (interpolation at Chebyshev nodes computed with mpmath,
// 3e-18 before rounding); the reciprocal of m + 1 is v_rcp_f64 + two Newton steps; then exp2_scaled_nonpos(log2 u, -a).
__device__ __forceinline__ double log2_ge1(double u) {
    double m = __builtin_amdgcn_frexp_mant(u);                      // in [1/2, 1)
    int e = __builtin_amdgcn_frexp_exp(u);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;                                            // in [1/sqrt 2, sqrt 2)
    e = lo ? e - 1 : e;
    const double f = m - 1.0, dn = m + 1.0;
    double r = __builtin_amdgcn_rcp(dn);
    r = __builtin_fma(__builtin_fma(-dn, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-dn, r, 1.0), r, r);
    const double z = f * r, z2 = z * z;
    auto step = [](double q, double yy, double c) { double o; asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(q), "v"(yy), "v"(c)); return o; };
    double q = 0x1.2f4d88c4fad06p-4;
    q = __builtin_fma(q, z2, 0x1.3999bf614de60p-4);
    q = step(q, z2, 0x1.7466994f20a7cp-4);
    q = step(q, z2, 0x1.c71c50004ffddp-4);
    q = step(q, z2, 0x1.2492494513f76p-3);
    q = step(q, z2, 0x1.999999997aeaap-3);
    q = step(q, z2, 0x1.55555555555aep-2);
    q = __builtin_fma(q, z2, 1.0);
    return __builtin_fma((z + z) * q, 0x1.71547652b82fep+0, (double)e);      // e + ln(m) log2(e)
}
__device__ __forceinline__ double rq_pow(double u, double a) {
    const double v = exp2_scaled_nonpos(log2_ge1(u), -a, 0.0);
    return u <= 1.7e308 ? v : (u > 1.7e308 ? 0.0 : u);            // u = inf: 0; NaN: NaN
}
__device__ __forceinline__ float rq_pow(float u, float a) { return cg_pow(u, -a); }
// s^g in fp64 for s >= 0, g > 0 (gamma-exponential profile: s^(gamma/2)); 0^g = 0, NaN propagates, inf^g = inf.  log2_ge1's
// reduction is valid for every positive normal or denormal argument; exp2_scaled_nonpos takes either sign of the exponent.
__device__ __forceinline__ double pow_pos(double s, double g) {
    const double v = exp2_scaled_nonpos(log2_ge1(s), g, 0.0);
    return (s > 0.0 && s <= 1.7e308) ? v : (s == 0.0 ? 0.0 : s);  // 0 -> 0, inf -> inf, NaN -> NaN
}
__device__ __forceinline__ float pow_pos(float s, float g) { return cg_pow(s, g); }
// ---- LDS-table forms for kernels that fill the tables first (exp_tab_lds_fill / log_tab_lds_fill): dense fp64 value kernels, gradient jets ----
// log2(u) for positive finite u (normal or denormal) by table, read from an LDS copy (exp2_table.hpp: LOG2_TAB128): u = m 2^E with m in
// [1/2, 1) (v_frexp_mant / _exp), j = the top 7 mantissa bits of m, r = m * RN(1 / c_j) - 1 in one fma (|r| <= 2^-8, exact), and
// log2 m = -log2(RN(1 / c_j)) + log2(1 + r) with the degree-6 series of log(1 + r) (remainder 2.8e-18).  Absolute error ~1e-16 + one
// rounding at the magnitude of the result — what 2^(c log2 u) needs (the entry-wise tests of the RQ / gamma-exponential profiles hold
// it to (4 + |c log2 u|) ulp).  15 instructions + one ds_read_b128 against the 28 of log2_ge1 (reciprocal with two Newton steps +
// an atanh series).  0, inf, NaN: garbage — the callers' own tests on their argument replace those.
__device__ __forceinline__ double (*log_tab_lds())[2] { __shared__ __attribute__((aligned(16))) double t[128][2]; return t; }
__device__ __forceinline__ void log_tab_lds_fill() {
    double (*t)[2] = log_tab_lds();
    for (int i = threadIdx.x; i < 128; i += blockDim.x) { t[i][0] = LOG2_TAB128[i][0]; t[i][1] = LOG2_TAB128[i][1]; }
    __syncthreads();
}
__device__ __forceinline__ double log2_lds(double u) {
    const double m = __builtin_amdgcn_frexp_mant(u);
    const int e = __builtin_amdgcn_frexp_exp(u);
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 ent = *reinterpret_cast<const d2*>(log_tab_lds()[(__double2hiint(m) >> 13) & 127]);
    const double r = __builtin_fma(m, ent.x, -1.0);
    double q = __builtin_fma(r, -0x1.5555555555555p-3, 0x1.999999999999ap-3);      // -1/6, 1/5
    q = __builtin_fma(q, r, -0.25);
    q = __builtin_fma(q, r, 0x1.5555555555555p-2);                                  // 1/3
    q = __builtin_fma(q, r, -0.5);
    q = __builtin_fma(q, r, 1.0);
    return __builtin_fma(r * q, 0x1.71547652b82fep+0, ent.y + (double)e);          // log2(e) ln(1 + r) + (log2 c_j + E)
}
// 2^(-L a) for L >= 0 and a uniform a > 0 on the LDS table (the rational-quadratic profile's u^(-alpha) with L = log2 u, 13 instructions
// against the 23 of exp2_scaled_nonpos): L a = n / 256 + r exactly (n from the low word of fma(L, -256 a, 1.5 * 2^52)); L is clamped at
// 1100 / a by ONE v_min_f64 — a NaN is restored by the caller's own test on u (rq_pow's).
__device__ __forceinline__ double exp2_neg_prod_lds(double L, double a) {
    const double lc = __builtin_fmin(L, 1100.0 / a);
    double magic = 0x1.8p52, c256 = -256.0 * a;
    asm("" : "+v"(magic));
    const double nb = __builtin_fma(lc, c256, magic);
    const int ni = __double2loint(nb);
    const double n = nb - magic;
    const double r = __builtin_fma(lc, c256, -n);
    const double tj = exp_tab_lds()[ni & 255];
    double q = __builtin_fma(r, 0x1.3b2ab6fba4e77p-39, 0x1.c6b08d704a0c0p-29);
    q = __builtin_fma(q, r, 0x1.ebfbdff82c58fp-19);
    q = __builtin_fma(q, r, 0x1.62e42fefa39efp-9);
    return __builtin_ldexp(__builtin_fma(tj, r * q, tj), ni >> 8);
}
// 2^(L c) for finite L, either sign, |L c| <= ~1100 (the gamma-exponential's s^(gamma/2) with L = log2 s in [-1074, 1024] and
// c = gamma / 2 <= 1): no clamp — ldexp takes the whole exponent range, the caller's own tests on s replace the specials
// (pow_pos's: 0 -> 0, inf -> inf, NaN -> NaN).  12 instructions against the 23 of exp2_scaled_nonpos.
__device__ __forceinline__ double exp2_prod_lds(double L, double c) {
    double magic = 0x1.8p52, c256 = 256.0 * c;
    asm("" : "+v"(magic));
    const double nb = __builtin_fma(L, c256, magic);
    const int ni = __double2loint(nb);
    const double n = nb - magic;
    const double r = __builtin_fma(L, c256, -n);
    const double tj = exp_tab_lds()[ni & 255];
    // (here r = 256 L c - n carries the POSITIVE sign convention: exp2(x) = 2^(n >> 8) T[n & 255] exp(r ln2 / 256))
    double q = __builtin_fma(r, 0x1.3b2ab6fba4e77p-39, 0x1.c6b08d704a0c0p-29);
    q = __builtin_fma(q, r, 0x1.ebfbdff82c58fp-19);
    q = __builtin_fma(q, r, 0x1.62e42fefa39efp-9);
    return __builtin_ldexp(__builtin_fma(tj, r * q, tj), ni >> 8);
}
// exp(-t / 2), t >= 0 or NaN, on the LDS table (the gamma-exponential profile's outer exponential): -128 log2(e) in two parts
__device__ __forceinline__ double exp_neg_half_lds(double t) {
    const double tc = __hiloint2double(t > 1525.0 ? 0x4097D400 : __double2hiint(t), __double2loint(t));
    const double clo = -0x1.777d0ffda0d24p-49;
    double magic = 0x1.8p52, chi = -0x1.71547652b82fep+7;
    asm("" : "+v"(magic));
    asm("" : "+s"(chi));
    const double nb = __builtin_fma(tc, chi, magic);
    const int ni = __double2loint(nb);
    const double n = nb - magic;
    const double f = __builtin_fma(tc, clo, __builtin_fma(tc, chi, -n));
    const double tj = exp_tab_lds()[ni & 255];
    double q = __builtin_fma(f, 0x1.3b2ab6fba4e77p-39, 0x1.c6b08d704a0c0p-29);
    q = __builtin_fma(q, f, 0x1.ebfbdff82c58fp-19);
    q = __builtin_fma(q, f, 0x1.62e42fefa39efp-9);
    return __builtin_ldexp(__builtin_fma(tj, f * q, tj), ni >> 8);
}

__device__ __forceinline__ double eq_exp_neg_half(double s) { return exp2_scaled_nonpos(s, -0x1.71547652b82fep-1, -0x1.777d0ffda0d24p-57); }   // -log2(e)/2
// exp(-t/2), t >= 0
__device__ __forceinline__ float cg_exp_neg_half(float t) { return cg_exp(-0.5f * t); }
__device__ __forceinline__ double cg_exp_neg_half(double t) { return eq_exp_neg_half(t); }
// exp(-r), r >= 0
__device__ __forceinline__ float cg_exp_neg(float r) { return cg_exp(-r); }
__device__ __forceinline__ double cg_exp_neg(double r) { return exp2_scaled_nonpos(r, -0x1.71547652b82fep+0, -0x1.777d0ffda0d24p-56); }      // -log2(e)

template <typename T, bool FOLDED>
struct Phi<COVGRAM_EQ, T, FOLDED> {
    static __device__ __forceinline__ T eval(T s, const KParams<T>& kp) {
        if constexpr (FOLDED) return exp2_neg_tab(s);   // fp32: one v_exp_f32 with a free neg modifier; fp64: the table form (14 + 1 load)
        else if constexpr (sizeof(T) == 8) return eq_exp_neg_half(s);
        else return cg_exp2(s * (T)-0.72134752044448170368);   // -log2(e)/2 as a literal: kp.c0 holds the same value, but a kernarg field can be
                                                                   // re-read inside a register-starved loop (grad_mvm.hpp)
    }
};
template <typename T, bool F>
struct Phi<COVGRAM_EXP, T, F> {
    static __device__ __forceinline__ T eval(T s, const KParams<T>&) { return exp_neg_tab(cg_sqrt(s)); }   // (fp64: the table form, 15 + 1 load; the polynomial cg_exp_neg measured 6 % slower than the library exp here)
};
// FOLDED, fp32 dense kernels (round 4): the host pre-scales the coordinates by log2(e) / l, so sqrt(s) IS r log2(e) and
// exp(-r) = exp2(-sqrt(s)) — v_sqrt_f32 + v_exp_f32 with a free negate, one multiplication fewer per pair (make_host_kernel)
template <>
struct Phi<COVGRAM_EXP, float, true> {
    static __device__ __forceinline__ float eval(float s, const KParams<float>&) { return cg_exp2(-cg_sqrt(s)); }
};
// which profiles the dense (value-only) kernels evaluate in their folded form: must agree with make_host_kernel(for_gradient = false)
template <int FAM, typename T> constexpr bool dense_folded = (FAM == COVGRAM_EQ || FAM == COVGRAM_MATERNP || (FAM == COVGRAM_EXP && sizeof(T) == 4));
template <typename T, bool F>
struct Phi<COVGRAM_RQ, T, F> {
    static __device__ __forceinline__ T eval(T s, const KParams<T>& kp) {
        T u = cg_fma(s, kp.c0, (T)1);                   // 1 + s/(2 alpha)
        return rq_pow(u, kp.param);
    }
};
template <typename T, bool F>
struct Phi<COVGRAM_GAMMAEXP, T, F> {
    static __device__ __forceinline__ T eval(T s, const KParams<T>& kp) {
        // s^(gamma/2); s == 0 -> 0 (gamma > 0) — log2(0) = -inf, exp2(-inf) = 0
        T t = (kp.param == (T)0) ? (T)1 : pow_pos(s, kp.param);
        return cg_exp_neg_half(t);
    }
};
template <typename T, bool F>
struct Phi<COVGRAM_CAUCHY, T, F> {
    static __device__ __forceinline__ T eval(T s, const KParams<T>&) { return cg_rcp((T)1 + s); }
};
template <typename T, bool F>
struct Phi<COVGRAM_IMQ, T, F> {
    static __device__ __forceinline__ T eval(T s, const KParams<T>& kp) { return cg_rsqrt(s + kp.param); }
};
// FOLDED (dense MVM kernels): the host pre-scales the coordinates by log2(e) sqrt(2p+1) / l, so sqrt(s) IS r log2(e);
// exp(-r) = exp2(-sqrt(s)) (negation is a free source modifier) and the polynomial / Taylor tables are rescaled to that
// argument (make_host_kernel): two multiplications fewer per pair.
template <typename T>
struct Phi<COVGRAM_MATERNP, T, true> {
    static __device__ __forceinline__ T eval(T s, const KParams<T>& kp) {
        // The reference's Taylor branch below eps^(1/p) (src/stationary.jl:135-146) exists for the DERIVATIVES at 0; the value
        // q(r) exp(-r) has no cancellation there and agrees with the truncated series to << eps at the bound (the first term
        // the series drops is r^(2p+1) <= eps^(1 + 1/(2p))), so the value-only kernels skip it: s is a sum of squares >= 0.
        // fp64: sqrt(s + 2^-1000) without the class test of cg_sqrt (s = 0 gives r = 2^-500: q(r) exp2(-r) = 1 exactly; any s the
        // addition changes is below 2^-947, where the value is 1 to far below eps; s = inf / NaN give NaN as q(inf) * 0 does)
        return eval_tab(s, kp, EXP2_TAB256);
    }
    // SHIFTED: the caller's s already carries the + 2^-1000 (the dense fp64 kernels start their sum of squares from it: free)
    template <bool SHIFTED = false>
    static __device__ __forceinline__ T eval_tab(T s, const KParams<T>& kp, const double* __restrict__ tab) {
        T rr;
        if constexpr (sizeof(T) == 8) rr = cg_sqrt_pos(SHIFTED ? s : s + (T)0x1p-1000); else rr = cg_sqrt(s);
        T e = exp2_neg_tab<false>(rr, tab);
        // the polynomial at its own degree where the order is a constant of the caller's copy of the loop (p = 1, 2: one / two fmas;
        // the zero-padded degree-3 form otherwise)
        T q = kp.p == 1 ? cg_fma(kp.h0[1], rr, kp.h0[0])
            : kp.p == 2 ? cg_fma(cg_fma(kp.h0[2], rr, kp.h0[1]), rr, kp.h0[0])
            : (kp.p <= 3) ? horner3(kp.h0, rr) : horner(kp.h0, kp.p, rr);
        return q * e;
    }
};
template <typename T>
struct Phi<COVGRAM_MATERNP, T, false> {
    static __device__ __forceinline__ T eval(T s, const KParams<T>& kp) {
        // value only: no Taylor branch (see the folded specialisation above; the derivative jets keep it)
        T r = cg_sqrt(kp.mp_c * s);
        T e = cg_exp(-r);
        // nu <= 7/2: fixed-degree Horner on the zero-padded table (same values, no loop, no indexed loads)
        T q = (kp.p <= 3) ? horner3(kp.h0, r) : horner(kp.h0, kp.p, r);
        return q * e;
    }
};
// ---- Matern with real nu (src/stationary.jl:87-114): modified Bessel function of the second kind -----------------------
// K_a(x), a >= 0, x > 0.  hc[0..6] = round(a), mu = a - round(a), and the mu-only constants of Temme's method
// (gam1, gam2, 1/Gamma(1+mu), 1/Gamma(1-mu), pi mu / sin(pi mu)), prepared by the host (api.hip).  x < 2: Temme's series
// for K_mu and K_mu+1; x >= 2: Steed's continued fraction (CF2); then the upward recurrence K_{b+1} = K_{b-1} + (2b/x) K_b,
// which is the stable direction for K.  The iteration counts depend on the lane's x; the loops stop at eps(T).
template <typename T>
__device__ __forceinline__ T cg_log(T x) { return cg_log2(x) * (T)0.69314718055994530942; }

template <typename T>
__device__ T besselk(T x, const T* __restrict__ hc) {
    const T EPS = (sizeof(T) == 8) ? (T)2.2e-16 : (T)1.2e-7;
    const int nl = (int)hc[0];
    const T mu = hc[1], gam1 = hc[2], gam2 = hc[3], gampl = hc[4], gammi = hc[5], fact = hc[6];
    const T mu2 = mu * mu, xi = (T)1 / x, xi2 = (T)2 * xi;
    T rkmu, rk1;
    if (x < (T)2) {
        const T b = (T)0.5 * x;
        T d = -cg_log(b);
        T e = mu * d;
        const T ee = cg_exp(e), ie = (T)1 / ee;
        const T ch = (T)0.5 * (ee + ie);
        const T fact2 = (e > -EPS && e < EPS) ? (T)1 : (T)0.5 * (ee - ie) / e;          // sinh(e) / e
        T ff = fact * (gam1 * ch + gam2 * fact2 * d);
        T sum = ff;
        T p = (T)0.5 * ee / gampl, q = (T)0.5 * ie / gammi, c = (T)1;
        d = b * b;
        T sum1 = p;
        for (int i = 1; i < 500; ++i) {
            const T fi = (T)i;
            ff = (fi * ff + p + q) / (fi * fi - mu2);
            c *= d / fi;
            p /= (fi - mu);
            q /= (fi + mu);
            const T del = c * ff;
            sum += del;
            sum1 += c * (p - fi * ff);
            if (!(del > sum * EPS || -del > sum * EPS)) break;                            // |del| <= |sum| eps (sum > 0)
        }
        rkmu = sum;
        rk1 = sum1 * xi2;
    } else {
        T b = (T)2 * ((T)1 + x), d = (T)1 / b, h = d, delh = d, q1 = (T)0, q2 = (T)1;
        const T a1 = (T)0.25 - mu2;
        T q = a1, c = a1, a = -a1;
        T s = (T)1 + q * delh;
        for (int i = 2; i < 500; ++i) {
            a -= (T)(2 * (i - 1));
            c = -a * c / (T)i;
            const T qnew = (q1 - b * q2) / a;
            q1 = q2; q2 = qnew;
            q += c * qnew;
            b += (T)2;
            d = (T)1 / (b + a * d);
            delh = (b * d - (T)1) * delh;
            h += delh;
            const T dels = q * delh;
            s += dels;
            const T rel = dels / s;
            if (!(rel > EPS || -rel > EPS)) break;
        }
        h = a1 * h;
        rkmu = cg_sqrt((T)1.57079632679489661923 * xi) * cg_exp(-x) / s;
        rk1 = rkmu * (mu + x + (T)0.5 - h) * xi;
    }
    for (int i = 1; i <= nl; ++i) {
        const T t = (mu + (T)i) * xi2 * rk1 + rkmu;
        rkmu = rk1; rk1 = t;
    }
    return rkmu;
}

// kp.param = nu, kp.mp_c = 2 nu, kp.c0 = 2^(1-nu)/Gamma(nu), kp.mp_bound = taylor_bound, kp.ty = its polynomial,
// kp.h0 / h1 / h2 = the Temme constants for the orders nu, |nu-1|, |nu-2|
template <typename T, bool F>
struct Phi<COVGRAM_MATERN, T, F> {
    static __device__ __forceinline__ T eval(T s, const KParams<T>& kp) {
        if (s < kp.mp_bound) return cg_fma(cg_fma(kp.ty[2], s, kp.ty[1]), s, (T)1);      // src/stationary.jl:100-110
        if (!(s > (T)0)) return (T)1;                                                     // k(x, x) = 1 (also for nu <= 1)
        const T r = cg_sqrt(kp.mp_c * s);
        return kp.c0 * cg_pow(r, kp.param) * besselk<T>(r, kp.h0);
    }
};

template <typename T, bool F>
struct Phi<COVGRAM_DOT, T, F> {
    static __device__ __forceinline__ T eval(T s, const KParams<T>&) { return s; }
};
template <typename T, bool F>
struct Phi<COVGRAM_EXPDOT, T, F> {
    static __device__ __forceinline__ T eval(T s, const KParams<T>&) { return cg_exp(s); }
};
// NeuralNetwork kernel on normalised augmented inputs (src/mercer.jl:82-85): (2/pi) asin(x^.y^)
template <typename T, bool F>
struct Phi<COVGRAM_ASINDOT, T, F> {
    static __device__ __forceinline__ T eval(T s, const KParams<T>&) { return (T)0.63661977236758134308 * asin(s); }
};

// ---- runtime-family evaluation (dense instantiation, composite factors): a wave-uniform switch ------------------
template <typename T>
__device__ __forceinline__ T phi_any(int family, T s, const KParams<T>& kp) {
    T v;
    switch (family) {
        case COVGRAM_EQ: v = Phi<COVGRAM_EQ, T, false>::eval(s, kp); break;
        case COVGRAM_EXP: v = Phi<COVGRAM_EXP, T, false>::eval(s, kp); break;
        case COVGRAM_RQ: v = Phi<COVGRAM_RQ, T, false>::eval(s, kp); break;
        case COVGRAM_GAMMAEXP: v = Phi<COVGRAM_GAMMAEXP, T, false>::eval(s, kp); break;
        case COVGRAM_CAUCHY: v = Phi<COVGRAM_CAUCHY, T, false>::eval(s, kp); break;
        case COVGRAM_IMQ: v = Phi<COVGRAM_IMQ, T, false>::eval(s, kp); break;
        case COVGRAM_MATERNP: v = Phi<COVGRAM_MATERNP, T, false>::eval(s, kp); break;
        case COVGRAM_DOT: v = s; break;
        case COVGRAM_EXPDOT: v = Phi<COVGRAM_EXPDOT, T, false>::eval(s, kp); break;
        case COVGRAM_MATERN: v = Phi<COVGRAM_MATERN, T, false>::eval(s, kp); break;
        case COVGRAM_ASINDOT: v = Phi<COVGRAM_ASINDOT, T, false>::eval(s, kp); break;
        default: v = (T)1; break;                         // COVGRAM_CONSTANT: the factor is its scale
    }
    if (kp.power != 1) v = ipow(v, kp.power);
    return v;
}

// composite value: sum over terms of coef_t * product of the term's profile factors; ISO factors see s / l_f^2
template <typename T, bool ISO>
__device__ __forceinline__ T expr_value(T s, const ExprParams<T>& ep) {
    T total = (T)0;
    int fi = 0;
    for (int t = 0; t < ep.nterms; ++t) {
        T prod = ep.coef[t];
        for (int f = 0; f < ep.nfac[t]; ++f, ++fi) {
            const KParams<T>& q = ep.f[fi];
            prod *= phi_any<T>(ep.fam[fi], ISO ? s * q.gamma2 : s, q);
        }
        total += prod;
    }
    return total;
}

// The same composite for a block of BG column groups, factor-outer: the factor's parameters are read (scalar loads)
// and its family dispatched (scalar branch) once per block; the inner loops are straight-line math over the block.
// Each finished term goes straight into the NR accumulators: acc[c] += w[g][c] * term[g], w[g][c] = wts[g * wstride + c].
template <typename T, bool ISO, int BG, int NR>
__device__ __forceinline__ void expr_accumulate_block(const typename Pk<T>::V (&s)[BG], const ExprParams<T>& ep,
                                                      const typename Pk<T>::V* __restrict__ wts, int wstride,
                                                      typename Pk<T>::V (&acc)[NR]) {
    using PK = Pk<T>;
    using V = typename PK::V;
    int fi = 0;
    for (int t = 0; t < ep.nterms; ++t) {
        V prod[BG];
        const V coef = PK::splat(ep.coef[t]);
#pragma unroll
        for (int g = 0; g < BG; ++g) prod[g] = coef;
        for (int f = 0; f < ep.nfac[t]; ++f, ++fi) {
            const KParams<T>& q = ep.f[fi];
            const V g2 = PK::splat(ISO ? q.gamma2 : (T)1);
            const int pw = q.power;
#define CG_EXPR_CASE(F)                                                                                                          \
    case F:                                                                                                                      \
        if (pw == 1) {                                                                                                           \
            _Pragma("unroll") for (int g = 0; g < BG; ++g)                                                                       \
                prod[g] = prod[g] * PK::map(s[g] * g2, [&](T sv) { return Phi<F, T, false>::eval(sv, q); });                     \
        } else {                                                                                                                 \
            _Pragma("unroll") for (int g = 0; g < BG; ++g)                                                                       \
                prod[g] = prod[g] * PK::map(s[g] * g2, [&](T sv) { return ipow(Phi<F, T, false>::eval(sv, q), pw); });           \
        }                                                                                                                        \
        break;
            switch (ep.fam[fi]) {
                CG_EXPR_CASE(COVGRAM_EQ)
                CG_EXPR_CASE(COVGRAM_EXP)
                CG_EXPR_CASE(COVGRAM_RQ)
                CG_EXPR_CASE(COVGRAM_GAMMAEXP)
                CG_EXPR_CASE(COVGRAM_CAUCHY)
                CG_EXPR_CASE(COVGRAM_IMQ)
                CG_EXPR_CASE(COVGRAM_MATERNP)
                CG_EXPR_CASE(COVGRAM_MATERN)
                CG_EXPR_CASE(COVGRAM_ASINDOT)
                CG_EXPR_CASE(COVGRAM_EXPDOT)
                default:                                           // COVGRAM_DOT
                    CG_EXPR_CASE(COVGRAM_DOT)
            }
#undef CG_EXPR_CASE
        }
#pragma unroll
        for (int g = 0; g < BG; ++g)
#pragma unroll
            for (int c = 0; c < NR; ++c) acc[c] = PK::fma(wts[g * wstride + c], prod[g], acc[c]);
    }
}

template <typename T, bool F>
struct Phi<FAM_EXPR_ISO, T, F> {
    static __device__ __forceinline__ T eval(T s, const ExprParams<T>& ep) { return expr_value<T, true>(s, ep); }
};
template <typename T, bool F>
struct Phi<FAM_EXPR_DOT, T, F> {
    static __device__ __forceinline__ T eval(T s, const ExprParams<T>& ep) { return expr_value<T, false>(s, ep); }
};

template <int FAM, typename T, bool FOLDED, bool POW>
__device__ __forceinline__ T phi_value(T s, const typename ParamsOf<FAM, T>::type& kp) {
    T v = Phi<FAM, T, FOLDED>::eval(s, kp);
    if constexpr (POW) v = ipow(v, kp.power);
    return v;   // Constant scale is folded into alpha by the host
}

// ---- (phi, phi', phi'') w.r.t. the pre-scaled argument s' = gamma^2 s ---------------------------
template <int FAM, typename T>
struct DPhi;

template <typename T>
struct DPhi<COVGRAM_EQ, T> {
    static __device__ __forceinline__ void eval(T s, const KParams<T>& kp, T& v, T& d1, T& d2) {
        if constexpr (sizeof(T) == 8) v = eq_exp_neg_half(s);
        else v = cg_exp2(s * (T)-0.72134752044448170368);
        d1 = (T)-0.5 * v; d2 = (T)0.25 * v;
    }
};
template <typename T>
struct DPhi<COVGRAM_EXP, T> {
    static __device__ __forceinline__ void eval(T s, const KParams<T>&, T& v, T& d1, T& d2) {
        T rt = cg_sqrt(s); v = cg_exp_neg(rt);
        T ir = cg_rcp(rt);                               // s == 0: inf, like the reference's ForwardDiff
        d1 = (T)-0.5 * v * ir;
        d2 = (T)0.25 * v * (ir * ir + ir * ir * ir);
    }
};
template <typename T>
struct DPhi<COVGRAM_RQ, T> {
    static __device__ __forceinline__ void eval(T s, const KParams<T>& kp, T& v, T& d1, T& d2) {
        T a = kp.param;
        T u = cg_fma(s, kp.c0, (T)1);
        T iu = cg_rcp(u);
        v = rq_pow(u, a);
        d1 = (T)-0.5 * v * iu;
        d2 = (a + (T)1) * ((T)0.5 * kp.c0) * v * iu * iu;   // (a+1)/(4a) u^(-a-2)
    }
};
template <typename T>
struct DPhi<COVGRAM_GAMMAEXP, T> {
    static __device__ __forceinline__ void eval(T s, const KParams<T>& kp, T& v, T& d1, T& d2) {
        T g = kp.param;                                   // gamma / 2
        T sg = (g == (T)0) ? (T)1 : pow_pos(s, g);
        T is = cg_rcp(s);
        v = cg_exp_neg_half(sg);
        T hg = (T)0.5 * g;
        d1 = -hg * sg * is * v;
        d2 = v * (hg * hg * sg * sg * is * is - hg * (g - (T)1) * sg * is * is);
    }
};
template <typename T>
struct DPhi<COVGRAM_CAUCHY, T> {
    static __device__ __forceinline__ void eval(T s, const KParams<T>&, T& v, T& d1, T& d2) {
        v = cg_rcp((T)1 + s); d1 = -v * v; d2 = (T)2 * v * v * v;
    }
};
template <typename T>
struct DPhi<COVGRAM_IMQ, T> {
    static __device__ __forceinline__ void eval(T s, const KParams<T>& kp, T& v, T& d1, T& d2) {
        T iu = cg_rcp(s + kp.param);
        v = cg_rsqrt(s + kp.param); d1 = (T)-0.5 * v * iu; d2 = (T)0.75 * v * iu * iu;
    }
};
template <typename T>
struct DPhi<COVGRAM_MATERNP, T> {
    // PFIX >= 0: the order is a compile-time constant (the callers branch on kp.p ONCE, outside their column loops, so that the
    // loop they run carries only that order's code and registers)
    // LDSTAB: exp(-r) on the exponential's table read from the kernel's LDS copy (the kernel fills it first; fp64 only)
    template <int PFIX = -1, bool LDSTAB = false>
    static __device__ __forceinline__ void eval(T s, const KParams<T>& kp, T& v, T& d1, T& d2) {
        const int p = PFIX >= 0 ? PFIX : kp.p;
        T r = cg_sqrt(kp.mp_c * s);
        T e;
        if constexpr (LDSTAB && sizeof(T) == 8) e = exp_neg_tab(r, exp_tab_lds()); else e = cg_exp_neg(r);
        if (p == 0) {                                     // Exp profile (singular at 0)
            T ir = cg_rcp(r);
            v = e;
            d1 = (T)-0.5 * e * ir; d2 = (T)0.25 * e * (ir * ir + ir * ir * ir);
            return;
        }
        if (p <= 3) {                                     // fixed-degree forms on the zero-padded tables
            v = horner3(kp.h0, r) * e;
            d1 = kp.mp_d1 * horner3(kp.h1, r) * e;        // d/dr[r^nu K_nu] = -r^nu K_{nu-1}
            if (p >= 2) d2 = kp.mp_d2 * horner3(kp.h2, r) * e;
            else d2 = -kp.mp_d1 * e * kp.mp_c * (T)0.5 * cg_rcp(r);
            if (s < kp.mp_bound) {                        // polynomial branch, differentiated termwise
                v = horner3(kp.ty, s);
                d1 = cg_fma(cg_fma((T)3 * kp.ty[3], s, (T)2 * kp.ty[2]), s, kp.ty[1]);
                d2 = cg_fma((T)6 * kp.ty[3], s, (T)2 * kp.ty[2]);
            }
            return;
        }
        v = horner(kp.h0, p, r) * e;
        d1 = kp.mp_d1 * horner(kp.h1, p - 1, r) * e;      // d/dr[r^nu K_nu] = -r^nu K_{nu-1}
        d2 = kp.mp_d2 * horner(kp.h2, p - 2, r) * e;
        if (s < kp.mp_bound) {                            // polynomial branch, differentiated termwise
            v = horner(kp.ty, p, s);
            T t1 = (T)0, t2 = (T)0;
            for (int i = p; i >= 1; --i) t1 = cg_fma(t1, s, kp.ty[i] * (T)i);
            for (int i = p; i >= 2; --i) t2 = cg_fma(t2, s, kp.ty[i] * (T)(i * (i - 1)));
            d1 = t1; d2 = t2;
        }
    }
};
// d/dr [r^a K_a(r)] = -r^a K_{a-1}(r) and dr/ds = nu / r  =>  phi' = -C nu r^(nu-1) K_{|nu-1|}(r),  phi'' = C nu^2 r^(nu-2) K_{|nu-2|}(r)
// (K_{-a} = K_a); below taylor_bound the derivatives of the reference's polynomial, as ForwardDiff sees it.
template <typename T>
struct DPhi<COVGRAM_MATERN, T> {
    static __device__ __forceinline__ void eval(T s, const KParams<T>& kp, T& v, T& d1, T& d2) {
        const T nu = kp.param;
        if (s < kp.mp_bound) {
            v = cg_fma(cg_fma(kp.ty[2], s, kp.ty[1]), s, (T)1);
            d1 = cg_fma((T)2 * kp.ty[2], s, kp.ty[1]);
            d2 = (T)2 * kp.ty[2];
            return;
        }
        const T r = cg_sqrt(kp.mp_c * s);
        const T lr = cg_log2(r);
        const T C = kp.c0;
        v = C * cg_exp2(nu * lr) * besselk<T>(r, kp.h0);
        d1 = -C * nu * cg_exp2((nu - (T)1) * lr) * besselk<T>(r, kp.h1);
        d2 = C * nu * nu * cg_exp2((nu - (T)2) * lr) * besselk<T>(r, kp.h2);
    }
};
template <typename T>
struct DPhi<COVGRAM_DOT, T> {
    static __device__ __forceinline__ void eval(T s, const KParams<T>&, T& v, T& d1, T& d2) {
        v = s; d1 = (T)1; d2 = (T)0;
    }
};
template <typename T>
struct DPhi<COVGRAM_ASINDOT, T> {   // f1, f2 of src/gradient.jl:192-194
    static __device__ __forceinline__ void eval(T s, const KParams<T>&, T& v, T& d1, T& d2) {
        const T c = (T)0.63661977236758134308;
        const T q = cg_rsqrt((T)1 - s * s);
        v = c * asin(s); d1 = c * q; d2 = c * s * q * q * q;
    }
};
template <typename T>
struct DPhi<COVGRAM_EXPDOT, T> {
    static __device__ __forceinline__ void eval(T s, const KParams<T>&, T& v, T& d1, T& d2) {
        v = cg_exp(s); d1 = v; d2 = v;
    }
};

// (phi^q, (phi^q)', (phi^q)'') from (phi, phi', phi'')
template <typename T>
__device__ __forceinline__ void power_jet(int q, T& v, T& d1, T& d2) {
    T vq2 = (q >= 2) ? ((q == 2) ? (T)1 : ipow(v, q - 2)) : (T)0;
    T vq1 = vq2 * v;
    if (q < 2) vq1 = (T)1;
    T n2 = (T)(q * (q - 1)) * vq2 * d1 * d1 + (T)q * vq1 * d2;
    d1 = (T)q * vq1 * d1;
    d2 = n2;
    v = vq1 * v;
}

// LIGHT: without Matern(nu) — its Bessel series, inlined into the per-pair interpreter loop of the lane-per-row gradient kernel, costs
// every composite registers and instruction cache; composites that do contain it take the panel path (api.hip)
template <typename T, bool LIGHT = false>
__device__ __forceinline__ void jet_any(int family, T s, const KParams<T>& kp, T& v, T& d1, T& d2) {
    switch (family) {
        case COVGRAM_EQ: DPhi<COVGRAM_EQ, T>::eval(s, kp, v, d1, d2); break;
        case COVGRAM_EXP: DPhi<COVGRAM_EXP, T>::eval(s, kp, v, d1, d2); break;
        case COVGRAM_RQ: DPhi<COVGRAM_RQ, T>::eval(s, kp, v, d1, d2); break;
        case COVGRAM_GAMMAEXP: DPhi<COVGRAM_GAMMAEXP, T>::eval(s, kp, v, d1, d2); break;
        case COVGRAM_CAUCHY: DPhi<COVGRAM_CAUCHY, T>::eval(s, kp, v, d1, d2); break;
        case COVGRAM_IMQ: DPhi<COVGRAM_IMQ, T>::eval(s, kp, v, d1, d2); break;
        case COVGRAM_MATERNP: DPhi<COVGRAM_MATERNP, T>::eval(s, kp, v, d1, d2); break;
        case COVGRAM_DOT: DPhi<COVGRAM_DOT, T>::eval(s, kp, v, d1, d2); break;
        case COVGRAM_EXPDOT: DPhi<COVGRAM_EXPDOT, T>::eval(s, kp, v, d1, d2); break;
        case COVGRAM_MATERN: if constexpr (!LIGHT) { DPhi<COVGRAM_MATERN, T>::eval(s, kp, v, d1, d2); } else { v = (T)1; d1 = (T)0; d2 = (T)0; } break;
        case COVGRAM_ASINDOT: DPhi<COVGRAM_ASINDOT, T>::eval(s, kp, v, d1, d2); break;
        default: v = (T)1; d1 = (T)0; d2 = (T)0; break;   // COVGRAM_CONSTANT
    }
    if (kp.power != 1) power_jet(kp.power, v, d1, d2);
}

// composite jet w.r.t. the raw s: chain rule through each factor's 1/l^2, product rule within a term, sum over terms
template <typename T, bool ISO, bool LIGHT = false>
__device__ __forceinline__ void expr_jet(T s, const ExprParams<T>& ep, T& v, T& d1, T& d2) {
    v = (T)0; d1 = (T)0; d2 = (T)0;
    int fi = 0;
    for (int t = 0; t < ep.nterms; ++t) {
        T p0 = ep.coef[t], p1 = (T)0, p2 = (T)0;
        for (int f = 0; f < ep.nfac[t]; ++f, ++fi) {
            const KParams<T>& q = ep.f[fi];
            const T g2 = ISO ? q.gamma2 : (T)1;
            T fv, f1, f2;
            jet_any<T, LIGHT>(ep.fam[fi], s * g2, q, fv, f1, f2);
            f1 *= g2; f2 *= g2 * g2;
            p2 = p2 * fv + (T)2 * p1 * f1 + p0 * f2;
            p1 = p1 * fv + p0 * f1;
            p0 = p0 * fv;
        }
        v += p0; d1 += p1; d2 += p2;
    }
}
// N composite values at once, factor-outer, for the matrix-core tiles (a lane's 16 entries of a 32 x 32 tile): parameters and
// family dispatch once per factor and tile.  Only the profiles that are smooth in s at 0 (the host admits no others here).
template <typename T, bool ISO, int N>
__device__ __forceinline__ void expr_value_block(const T (&s)[N], const ExprParams<T>& ep, T (&out)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = (T)0;
    int fi = 0;
    for (int t = 0; t < ep.nterms; ++t) {
        T prod[N];
#pragma unroll
        for (int i = 0; i < N; ++i) prod[i] = ep.coef[t];
        for (int f = 0; f < ep.nfac[t]; ++f, ++fi) {
            const KParams<T>& q = ep.f[fi];
            const T g2 = ISO ? q.gamma2 : (T)1;
            const int pw = q.power;
#define CG_EXPRV_CASE(F)                                                                                            \
    case F:                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < N; ++i) {                                                             \
            T v = Phi<F, T, false>::eval(s[i] * g2, q);                                                             \
            if (pw != 1) v = ipow(v, pw);                                                                           \
            prod[i] *= v;                                                                                           \
        }                                                                                                           \
        break;
            switch (ep.fam[fi]) {
                CG_EXPRV_CASE(COVGRAM_EQ)
                CG_EXPRV_CASE(COVGRAM_RQ)
                CG_EXPRV_CASE(COVGRAM_CAUCHY)
                CG_EXPRV_CASE(COVGRAM_IMQ)
                CG_EXPRV_CASE(COVGRAM_MATERNP)
                CG_EXPRV_CASE(COVGRAM_EXPDOT)
                default:
                    CG_EXPRV_CASE(COVGRAM_DOT)
            }
#undef CG_EXPRV_CASE
        }
#pragma unroll
        for (int i = 0; i < N; ++i) out[i] += prod[i];
    }
}
// The composite jet for a block of BG column groups, factor-outer (the gradient counterpart of expr_accumulate_block):
// parameters and family dispatch once per factor and block.
template <typename T, bool ISO, int BG>
__device__ __forceinline__ void expr_jet_block(const typename Pk<T>::V (&s)[BG], const ExprParams<T>& ep,
                                               typename Pk<T>::V (&v)[BG], typename Pk<T>::V (&d1)[BG], typename Pk<T>::V (&d2)[BG]) {
    using PK = Pk<T>;
    using V = typename PK::V;
#pragma unroll
    for (int g = 0; g < BG; ++g) { v[g] = PK::splat((T)0); d1[g] = PK::splat((T)0); d2[g] = PK::splat((T)0); }
    int fi = 0;
    for (int t = 0; t < ep.nterms; ++t) {
        V p0[BG], p1[BG], p2[BG];
#pragma unroll
        for (int g = 0; g < BG; ++g) { p0[g] = PK::splat(ep.coef[t]); p1[g] = PK::splat((T)0); p2[g] = PK::splat((T)0); }
        for (int f = 0; f < ep.nfac[t]; ++f, ++fi) {
            const KParams<T>& q = ep.f[fi];
            const T g2 = ISO ? q.gamma2 : (T)1;
            const int pw = q.power;
#define CG_JET_CASE(F)                                                                                              \
    case F:                                                                                                         \
        _Pragma("unroll") for (int g = 0; g < BG; ++g) {                                                            \
            V fv, f1, f2;                                                                                           \
            PK::map3(s[g], [&](T sv, T& a0, T& a1, T& a2) {                                                         \
                DPhi<F, T>::eval(sv * g2, q, a0, a1, a2);                                                           \
                if (pw != 1) power_jet(pw, a0, a1, a2);                                                             \
                a1 *= g2; a2 *= g2 * g2;                                                                            \
            }, fv, f1, f2);                                                                                         \
            p2[g] = p2[g] * fv + PK::splat((T)2) * p1[g] * f1 + p0[g] * f2;                                         \
            p1[g] = p1[g] * fv + p0[g] * f1;                                                                        \
            p0[g] = p0[g] * fv;                                                                                     \
        }                                                                                                           \
        break;
            switch (ep.fam[fi]) {
                CG_JET_CASE(COVGRAM_EQ)
                CG_JET_CASE(COVGRAM_EXP)
                CG_JET_CASE(COVGRAM_RQ)
                CG_JET_CASE(COVGRAM_GAMMAEXP)
                CG_JET_CASE(COVGRAM_CAUCHY)
                CG_JET_CASE(COVGRAM_IMQ)
                CG_JET_CASE(COVGRAM_MATERNP)
                CG_JET_CASE(COVGRAM_MATERN)
                CG_JET_CASE(COVGRAM_ASINDOT)
                CG_JET_CASE(COVGRAM_EXPDOT)
                default:
                    CG_JET_CASE(COVGRAM_DOT)
            }
#undef CG_JET_CASE
        }
#pragma unroll
        for (int g = 0; g < BG; ++g) { v[g] = v[g] + p0[g]; d1[g] = d1[g] + p1[g]; d2[g] = d2[g] + p2[g]; }
    }
}

template <typename T>
struct DPhi<FAM_EXPR_ISO, T> {
    static __device__ __forceinline__ void eval(T s, const ExprParams<T>& ep, T& v, T& d1, T& d2) { expr_jet<T, true, true>(s, ep, v, d1, d2); }
};
template <typename T>
struct DPhi<FAM_EXPR_DOT, T> {
    static __device__ __forceinline__ void eval(T s, const ExprParams<T>& ep, T& v, T& d1, T& d2) { expr_jet<T, false, true>(s, ep, v, d1, d2); }
};

// POW = false compiles the Power chain rule away: the gradient kernel's software pipeline needs the derivative
// evaluation to stay ONE basic block (a branch lets hipcc sink the prefetch loads past it, to their first use).
template <int FAM, typename T, bool POW, int PFIX = -1>
__device__ __forceinline__ void phi_jet(T s, const typename ParamsOf<FAM, T>::type& kp, T& v, T& d1, T& d2) {
    if constexpr (FAM == COVGRAM_MATERNP) DPhi<FAM, T>::template eval<PFIX>(s, kp, v, d1, d2);
    else DPhi<FAM, T>::eval(s, kp, v, d1, d2);
    if constexpr (POW) power_jet(kp.power, v, d1, d2);
}
template <int FAM, typename T, bool POW, int PFIX = -1>
__device__ __forceinline__ void phi_derivs(T s, const typename ParamsOf<FAM, T>::type& kp, T& d1, T& d2) {
    T v;
    phi_jet<FAM, T, POW, PFIX>(s, kp, v, d1, d2);
}

}  // namespace covgram
