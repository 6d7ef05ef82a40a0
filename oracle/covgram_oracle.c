/* covgram_oracle.c — plain-C CPU restatement of the reference's lazy-Gramian MVM hot loops.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * ("port": the timed CPU baseline printed next to the GPU number).  The product (libcovgram.so) never
 * links or calls it.  PARITY PIN STATUS: "parity unpinned" in the strict sense — Julia is unavailable
 * here and the reference ships no golden vectors; this file is pinned by the reference's own test
 * relations, closed forms and a 50-digit mpmath evaluation (see oracle/covgram_oracle.py header,
 * tests/test_oracle.py).  Citations: file:line relative to the reference root.
 *
 * Schedules mirror the reference: rows in parallel (`@threads for i`, src/gramian.jl:81,244) via
 * `#pragma omp parallel for`, inner column loop vectorised with a reduction (`@simd`, src/gramian.jl:82).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { F_EQ = 0, F_EXP, F_RQ, F_GAMMAEXP, F_CAUCHY, F_IMQ, F_MATERNP, F_DOT, F_EXPDOT };

typedef struct {
    int32_t family, trait, p, power;
    double param, lengthscale, scale;
} oracle_kernel;

#define MAXP 8

typedef struct {
    double h0[MAXP + 1], h1[MAXP + 1], h2[MAXP + 1], ty[MAXP + 1];
    double c, bound, d1, d2;
} maternp_tab;

static double fact(int n) { double f = 1; for (int i = 2; i <= n; ++i) f *= i; return f; }

/* normalised polynomial of stationary.jl:147-157: MaternP_q(s) = exp(-r) sum_m h[m] r^m */
static void mp_poly(int q, double* h) {
    const double nrm = fact(2 * q) / fact(q);                       /* stationary.jl:157 */
    for (int m = 0; m <= q; ++m) h[m] = fact(2 * q - m) / (fact(m) * fact(q - m)) * pow(2.0, m) / nrm; /* :184-191 */
}

static void mp_table(int p, double eps, maternp_tab* t) {
    memset(t, 0, sizeof(*t));
    t->c = 2.0 * p + 1.0;
    mp_poly(p, t->h0);
    if (p >= 1) mp_poly(p - 1, t->h1);
    if (p >= 2) mp_poly(p - 2, t->h2);
    t->ty[0] = 1.0;
    for (int i = 1; i <= p; ++i) {                                  /* stationary.jl:172-182, via the power series */
        long double coef = 0;
        for (int m = 0; m <= (p < 2 * i ? p : 2 * i); ++m) {
            int k = 2 * i - m;
            coef += (long double)t->h0[m] * ((k & 1) ? -1.0L : 1.0L) / (long double)fact(k);
        }
        t->ty[i] = (double)(coef * powl((long double)t->c, i));     /* = d_i / i! */
    }
    t->d1 = p >= 1 ? t->ty[1] : 0.0;
    t->d2 = p >= 2 ? 2.0 * t->ty[2] : 0.0;
    t->bound = p >= 1 ? pow(eps, 1.0 / p) : 0.0;                    /* stationary.jl:135 */
}

static inline double horner(const double* h, int deg, double r) {
    double acc = h[deg];
    for (int m = deg - 1; m >= 0; --m) acc = acc * r + h[m];
    return acc;
}

/* phi(s) — stationary.jl:42,53,60,71,132-158,224,235; mercer.jl:9,22; algebra.jl:61-62; transformation.jl:19 */
static inline double phi(const oracle_kernel* k, const maternp_tab* t, double s) {
    if (k->trait == 1 && k->lengthscale != 1.0) s /= k->lengthscale * k->lengthscale;
    double v;
    switch (k->family) {
        case F_EQ: v = exp(-s / 2); break;
        case F_EXP: v = exp(-sqrt(s)); break;
        case F_RQ: v = pow(1 + s / (2 * k->param), -k->param); break;
        case F_GAMMAEXP: v = exp(-pow(s, k->param / 2) / 2); break;
        case F_CAUCHY: v = 1.0 / (1.0 + s); break;
        case F_IMQ: v = 1.0 / sqrt(s + k->param * k->param); break;
        case F_MATERNP:
            if (k->p >= 1 && s < t->bound) v = horner(t->ty, k->p, s);
            else { double r = sqrt(t->c * s); v = horner(t->h0, k->p, r) * exp(-r); }
            break;
        case F_DOT: v = s; break;
        default: v = exp(s); break;
    }
    if (k->power != 1) { double b = v; for (int i = 1; i < k->power; ++i) v *= b; }
    return k->scale * v;
}

/* (phi', phi'') w.r.t. s — closed forms of gradient.jl:584-600 (see covgram_oracle.py::profile_derivatives) */
static inline void dphi(const oracle_kernel* k, const maternp_tab* t, double s, double* o1, double* o2) {
    double inner = 1.0;
    if (k->trait == 1 && k->lengthscale != 1.0) { inner = 1.0 / (k->lengthscale * k->lengthscale); s *= inner; }
    double v, d1, d2;
    switch (k->family) {
        case F_EQ: v = exp(-s / 2); d1 = -v / 2; d2 = v / 4; break;
        case F_EXP: { double rt = sqrt(s); v = exp(-rt); d1 = -v / (2 * rt); d2 = v * (1 / (4 * s) + 1 / (4 * s * rt)); } break;
        case F_RQ: { double a = k->param, u = 1 + s / (2 * a); v = pow(u, -a); d1 = -0.5 * pow(u, -a - 1); d2 = (a + 1) / (4 * a) * pow(u, -a - 2); } break;
        case F_GAMMAEXP: { double g = k->param / 2; v = exp(-pow(s, g) / 2); d1 = -(g / 2) * pow(s, g - 1) * v;
                           d2 = v * ((g / 2) * (g / 2) * pow(s, 2 * g - 2) - (g / 2) * (g - 1) * pow(s, g - 2)); } break;
        case F_CAUCHY: { double u = 1 + s; v = 1 / u; d1 = -1 / (u * u); d2 = 2 / (u * u * u); } break;
        case F_IMQ: { double u = s + k->param * k->param; v = pow(u, -0.5); d1 = -0.5 * pow(u, -1.5); d2 = 0.75 * pow(u, -2.5); } break;
        case F_MATERNP: {
            int p = k->p;
            double r = sqrt(t->c * s), e = exp(-r);
            if (p >= 1 && s < t->bound) {
                v = horner(t->ty, p, s); d1 = 0; d2 = 0;
                for (int i = p; i >= 1; --i) d1 = d1 * s + t->ty[i] * i;
                for (int i = p; i >= 2; --i) d2 = d2 * s + t->ty[i] * i * (i - 1);
            } else if (p == 0) {
                v = e; d1 = -e / (2 * r); d2 = e * (1 / (4 * s) + 1 / (4 * s * r));
            } else {
                v = horner(t->h0, p, r) * e;
                d1 = t->d1 * horner(t->h1, p - 1, r) * e;
                d2 = p >= 2 ? t->d2 * horner(t->h2, p - 2, r) * e : -t->d1 * e * t->c / (2 * r);
            }
        } break;
        case F_DOT: v = s; d1 = 1; d2 = 0; break;
        default: v = exp(s); d1 = v; d2 = v; break;
    }
    d1 *= inner; d2 *= inner * inner;
    if (k->power != 1) {
        int q = k->power;
        double vq2 = q >= 2 ? pow(v, q - 2) : 0.0, vq1 = pow(v, q - 1);
        double n2 = q * (q - 1) * vq2 * d1 * d1 + q * vq1 * d2;
        d1 = q * vq1 * d1; d2 = n2;
    }
    *o1 = k->scale * d1; *o2 = k->scale * d2;
}

/* ------------------------------------------------------------------------------------------------
 * mul!(y, G, a, alpha, beta) — gramian.jl:78-87 (+ matrix form :89-99).  X: n×d, Y: m×d point-major.
 * T = double.  a: m×nrhs (lda), y: n×nrhs (ldy), column-major.
 * ---------------------------------------------------------------------------------------------- */
#define DEFINE_MVM(NAME, T, EPS)                                                                                    \
    int NAME(const oracle_kernel* k, const T* X, int64_t n, const T* Y, int64_t m, int32_t d, const T* a,            \
             int64_t lda, T* y, int64_t ldy, int32_t nrhs, double alpha, double beta) {                              \
        maternp_tab tab;                                                                                             \
        if (k->family == F_MATERNP) { if (k->p < 0 || k->p > MAXP) return -2; mp_table(k->p, EPS, &tab); }           \
        const int iso = (k->trait == 1);                                                                             \
        for (int c = 0; c < nrhs; ++c) {                                                                             \
            const T* ac = a + (int64_t)c * lda;                                                                      \
            T* yc = y + (int64_t)c * ldy;                                                                            \
            _Pragma("omp parallel for schedule(static)")                                                             \
            for (int64_t i = 0; i < n; ++i) {                                                                        \
                const T* xi = X + i * (int64_t)d;                                                                    \
                T acc = (T)0;                          /* accumulate in eltype(y), like y[i] += ... */              \
                _Pragma("omp simd reduction(+ : acc)")                                                               \
                for (int64_t j = 0; j < m; ++j) {                                                                    \
                    const T* yj = Y + j * (int64_t)d;                                                                \
                    T s = (T)0;                                                                                      \
                    if (iso) for (int l = 0; l < d; ++l) { T r = xi[l] - yj[l]; s += r * r; }  /* util.jl:40-47 */  \
                    else     for (int l = 0; l < d; ++l) s += xi[l] * yj[l];                   /* mercer.jl:3 */     \
                    acc += (T)(alpha * phi(k, &tab, (double)s)) * ac[j];                                             \
                }                                                                                                    \
                yc[i] = (beta == 0.0 ? (T)0 : (T)beta * yc[i]) + acc;      /* gramian.jl:80 */                       \
            }                                                                                                        \
        }                                                                                                            \
        return 0;                                                                                                    \
    }

DEFINE_MVM(oracle_mvm_f64, double, 2.220446049250313e-16)
DEFINE_MVM(oracle_mvm_f32, float, 1.1920928955078125e-07)

/* Fast paths used by the timed CPU baseline: EQ with the profile inlined in the element type, exactly the
 * arithmetic Julia specialises gramian.jl:78-87 to for EQ on Float32 / Float64 points (expf / exp of -r²/2). */
int oracle_mvm_eq_f32(const float* X, int64_t n, const float* Y, int64_t m, int32_t d, const float* a, float* y,
                      int64_t row0, int64_t row1) {
    _Pragma("omp parallel for schedule(static)")
    for (int64_t i = row0; i < row1; ++i) {
        const float* xi = X + i * (int64_t)d;
        float acc = 0.f;
        _Pragma("omp simd reduction(+ : acc)")
        for (int64_t j = 0; j < m; ++j) {
            const float* yj = Y + j * (int64_t)d;
            float s = 0.f;
            for (int l = 0; l < d; ++l) { float r = xi[l] - yj[l]; s += r * r; }
            acc += expf(-s / 2) * a[j];
        }
        y[i - row0] = acc;
    }
    return 0;
}

int oracle_mvm_eq_f64(const double* X, int64_t n, const double* Y, int64_t m, int32_t d, const double* a, double* y,
                      int64_t row0, int64_t row1) {
    _Pragma("omp parallel for schedule(static)")
    for (int64_t i = row0; i < row1; ++i) {
        const double* xi = X + i * (int64_t)d;
        double acc = 0.;
        _Pragma("omp simd reduction(+ : acc)")
        for (int64_t j = 0; j < m; ++j) {
            const double* yj = Y + j * (int64_t)d;
            double s = 0.;
            for (int l = 0; l < d; ++l) { double r = xi[l] - yj[l]; s += r * r; }
            acc += exp(-s / 2) * a[j];
        }
        y[i - row0] = acc;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * blockmul!(y, G, a, alpha, beta) — gramian.jl:241-253 with the block mul! of gradient.jl:86-92 / 109-115.
 * ---------------------------------------------------------------------------------------------- */
int oracle_grad_mvm_f64(const oracle_kernel* k, const double* X, int64_t n, const double* Y, int64_t m, int32_t d,
                        const double* a, double* y, double alpha, double beta) {
    maternp_tab tab;
    if (k->family == F_MATERNP) { if (k->p < 0 || k->p > MAXP) return -2; mp_table(k->p, 2.220446049250313e-16, &tab); }
    const int iso = (k->trait == 1);
    _Pragma("omp parallel for schedule(static)")
    for (int64_t i = 0; i < n; ++i) {                               /* @threads for i (gramian.jl:244) */
        const double* xi = X + i * (int64_t)d;
        double* yi = y + i * (int64_t)d;
        for (int l = 0; l < d; ++l) yi[l] = (beta == 0.0) ? 0.0 : beta * yi[l];   /* gramian.jl:245 */
        for (int64_t j = 0; j < m; ++j) {                           /* gramian.jl:247-249 */
            const double* yj = Y + j * (int64_t)d;
            const double* aj = a + j * (int64_t)d;
            double k1, k2;
            if (iso) {                                              /* gradient.jl:86-92 */
                double r2 = 0, ra = 0;
                for (int l = 0; l < d; ++l) { double r = xi[l] - yj[l]; r2 += r * r; ra += r * aj[l]; }
                dphi(k, &tab, r2, &k1, &k2);
                for (int l = 0; l < d; ++l) yi[l] += alpha * -2.0 * (k1 * aj[l] + 2 * k2 * (xi[l] - yj[l]) * ra);
            } else {                                                /* gradient.jl:109-115 */
                double xy = 0, xa = 0;
                for (int l = 0; l < d; ++l) { xy += xi[l] * yj[l]; xa += xi[l] * aj[l]; }
                dphi(k, &tab, xy, &k1, &k2);
                for (int l = 0; l < d; ++l) yi[l] += alpha * (k1 * aj[l] + k2 * yj[l] * xa);
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Toeplitz MVM by circulant embedding (ToeplitzMatrices 0.7.1's algorithm), own radix-2 complex FFT.
 * T[i,j] = vc[i-j] (i>=j), vr[j-i] (i<j).  N = power of two >= n+m-1.
 * ---------------------------------------------------------------------------------------------- */
static void fft_inplace(double* re, double* im, int64_t N, int inverse) {
    for (int64_t i = 1, j = 0; i < N; ++i) {
        int64_t bit = N >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    for (int64_t len = 2; len <= N; len <<= 1) {
        const double ang = 2 * M_PI / (double)len * (inverse ? 1 : -1);
        const int64_t half = len >> 1;
        _Pragma("omp parallel for schedule(static) if (N >= 65536)")
        for (int64_t i = 0; i < N; i += len) {
            for (int64_t kx = 0; kx < half; ++kx) {
                const double wr = cos(ang * (double)kx), wi = sin(ang * (double)kx);
                const int64_t u = i + kx, v = u + half;
                const double xr = re[v] * wr - im[v] * wi, xi = re[v] * wi + im[v] * wr;
                re[v] = re[u] - xr; im[v] = im[u] - xi;
                re[u] += xr; im[u] += xi;
            }
        }
    }
    if (inverse) for (int64_t i = 0; i < N; ++i) { re[i] /= (double)N; im[i] /= (double)N; }
}

int oracle_toeplitz_mvm_f64(const double* vc, const double* vr, int64_t n, int64_t m, const double* a, double* y,
                            double alpha, double beta) {
    if (!vr) { vr = vc; m = n; }
    int64_t N = 1;
    while (N < n + m - 1 || N < 2) N <<= 1;
    double* buf = (double*)calloc((size_t)4 * N, sizeof(double));
    if (!buf) return -5;
    double *cr = buf, *ci = buf + N, *ar = buf + 2 * N, *ai = buf + 3 * N;
    for (int64_t i = 0; i < n; ++i) cr[i] = vc[i];
    for (int64_t kx = 1; kx < m; ++kx) cr[N - kx] = vr[kx];
    for (int64_t j = 0; j < m; ++j) ar[j] = a[j];
    fft_inplace(cr, ci, N, 0);
    fft_inplace(ar, ai, N, 0);
    for (int64_t i = 0; i < N; ++i) {
        const double r = cr[i] * ar[i] - ci[i] * ai[i], q = cr[i] * ai[i] + ci[i] * ar[i];
        ar[i] = r; ai[i] = q;
    }
    fft_inplace(ar, ai, N, 1);
    for (int64_t i = 0; i < n; ++i) y[i] = alpha * ar[i] + (beta == 0.0 ? 0.0 : beta * y[i]);
    free(buf);
    return 0;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
