// api.hip — C-ABI entry points of libcovgram.so (include/covgram.h): context / points handles,
// kernel-parameter construction, and the dense, dense-instantiate and gradient MVM drivers.
// Structured MVMs live in toeplitz.hip (Toeplitz / Circulant), kron.hip (Kronecker) and lowrank.hip (low rank).
#include <math.h>
#include <stdarg.h>

#include <algorithm>
#include <vector>

#include "dense_mvm.hpp"
#include "dense_sym32.hpp"
#include "dense_bcast.hpp"
#include "dense_wide.hpp"
#include "grad_mvm.hpp"
#include "grad_bcast.hpp"
#include "grad_wide.hpp"

namespace covgram {

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ------------------------------------------------------------------------------------------------
// compiled dimension set
// ------------------------------------------------------------------------------------------------
const int kDims[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64};
const int kNumDims = (int)(sizeof(kDims) / sizeof(kDims[0]));

int pad_dim(int d) {
    for (int i = 0; i < kNumDims; ++i)
        if (kDims[i] >= d) return kDims[i];
    return -1;
}

// per-family launchers (dense_fam.hip / grad_fam.hip, one translation unit per family)
#define CG_DECL(n)                                                 \
    int launch_dense_family_##n(const DenseArgs&, int dtype);      \
    int launch_dense_wide_family_##n(const DenseArgs&, int dtype); \
    int launch_grad_family_##n(const GradArgs&, int dtype);        \
    int launch_grad_wide_family_##n(const GradWideArgs&, int dtype);
CG_DECL(0) CG_DECL(1) CG_DECL(2) CG_DECL(3) CG_DECL(4) CG_DECL(5) CG_DECL(6) CG_DECL(7) CG_DECL(8) CG_DECL(9) CG_DECL(10) CG_DECL(11) CG_DECL(12)
#undef CG_DECL

dense_launch_fn dense_launcher(int family) {
    static const dense_launch_fn t[NUM_TU_FAMILIES] = {
        launch_dense_family_0, launch_dense_family_1, launch_dense_family_2, launch_dense_family_3, launch_dense_family_4,
        launch_dense_family_5, launch_dense_family_6, launch_dense_family_7, launch_dense_family_8, launch_dense_family_9,
        launch_dense_family_10, launch_dense_family_11, launch_dense_family_12};
    return (family >= 0 && family < NUM_TU_FAMILIES) ? t[family] : nullptr;
}
dense_launch_fn dense_wide_launcher(int family) {
    static const dense_launch_fn t[NUM_TU_FAMILIES] = {
        launch_dense_wide_family_0, launch_dense_wide_family_1, launch_dense_wide_family_2, launch_dense_wide_family_3,
        launch_dense_wide_family_4, launch_dense_wide_family_5, launch_dense_wide_family_6, launch_dense_wide_family_7,
        launch_dense_wide_family_8, launch_dense_wide_family_9, launch_dense_wide_family_10, launch_dense_wide_family_11, launch_dense_wide_family_12};
    return (family >= 0 && family < NUM_TU_FAMILIES) ? t[family] : nullptr;
}
grad_launch_fn grad_launcher(int family) {
    static const grad_launch_fn t[NUM_TU_FAMILIES] = {
        launch_grad_family_0, launch_grad_family_1, launch_grad_family_2, launch_grad_family_3, launch_grad_family_4,
        launch_grad_family_5, launch_grad_family_6, launch_grad_family_7, launch_grad_family_8, launch_grad_family_9,
        launch_grad_family_10, launch_grad_family_11, launch_grad_family_12};
    return (family >= 0 && family < NUM_TU_FAMILIES) ? t[family] : nullptr;
}

// ------------------------------------------------------------------------------------------------
// kernel parameters.  MaternP tables follow src/stationary.jl:117-191 (coefficients :184-191,
// normalisation :157, derivatives at zero :172-182 — derived from the power series, not SymEngine).
// ------------------------------------------------------------------------------------------------
static long double factl(int n) {
    long double f = 1;
    for (int i = 2; i <= n; ++i) f *= i;
    return f;
}

// normalised polynomial h[m], m = 0..q:  H_q(r) = exp(-r) sum_m h[m] r^m,  H_q(0) = 1
static void maternp_poly(int q, long double* h) {
    const long double nrm = factl(2 * q) / factl(q);
    for (int m = 0; m <= q; ++m) {
        // coefficient of (2r)^m: (2q-m)! / (m! (q-m)!)
        const long double c = factl(2 * q - m) / (factl(m) * factl(q - m));
        h[m] = c * powl(2.0L, m) / nrm;
    }
}

// d_i = d^i/ds^i MaternP_p(s) at s = 0 (i = 1..p): the r^(2i) series coefficient of exp(-r) q_p(r) times (2p+1)^i i!
static void maternp_derivs0(int p, long double* d /* d[1..p] */) {
    long double h[MAXP + 1];
    maternp_poly(p, h);
    for (int i = 1; i <= p; ++i) {
        long double coef = 0;
        for (int m = 0; m <= std::min(p, 2 * i); ++m) {
            const int k = 2 * i - m;
            coef += h[m] * (((k & 1) ? -1.0L : 1.0L) / factl(k));
        }
        d[i] = coef * powl((long double)(2 * p + 1), i) * factl(i);
    }
}

static int make_simple_kernel(const covgram_kernel* k, int dtype, bool for_gradient, HostKernel* out);

// Composite: every factor keeps its own parameter block with gamma = 1/l (unfolded EQ); rows and columns stay unscaled.
static int make_composite_kernel(const covgram_kernel_composite* c, int dtype, HostKernel* out) {
    const covgram_kernel& h = c->head;
    CG_REQUIRE(h.trait == COVGRAM_ISOTROPIC || h.trait == COVGRAM_DOTPRODUCT, COVGRAM_EINVAL, "composite: bad trait %d", h.trait);
    CG_REQUIRE(h.power == 1 && h.lengthscale == 1.0, COVGRAM_EINVAL, "composite head must have power == 1 and lengthscale == 1");
    CG_REQUIRE(c->nterms >= 1 && c->nterms <= EXPR_MAXT, COVGRAM_EUNSUPPORTED, "composite: %d terms (1..%d supported)", c->nterms, EXPR_MAXT);
    memset(out, 0, sizeof(*out));
    out->k = h;
    out->kp.gamma = 1.0; out->kp.gamma2 = 1.0; out->kp.scale = h.scale; out->kp.power = 1;
    out->eq_folded = false;
    out->tu_family = (h.trait == COVGRAM_ISOTROPIC) ? FAM_EXPR_ISO : FAM_EXPR_DOT;
    out->nterms = c->nterms;
    int nin = 0, nf = 0;   // factors consumed from the descriptor / profile factors kept for the device
    for (int t = 0; t < c->nterms; ++t) {
        CG_REQUIRE(c->nfactors[t] >= 1, COVGRAM_EINVAL, "composite: term %d has no factors", t);
        out->nfac[t] = 0;
        out->coef[t] = 1.0;
        for (int f = 0; f < c->nfactors[t]; ++f, ++nin) {
            CG_REQUIRE(nin < EXPR_MAXF, COVGRAM_EUNSUPPORTED, "composite: more than %d factors", EXPR_MAXF);
            const covgram_kernel& fk = c->factors[nin];
            if (fk.family == COVGRAM_CONSTANT) {   // Constants fold into the term's coefficient
                out->coef[t] *= fk.scale;
                continue;
            }
            CG_REQUIRE(fk.trait == h.trait, COVGRAM_EINVAL, "composite: factor %d has trait %d, head has %d (GenericInput is not a device path)",
                       nin, fk.trait, h.trait);
            HostKernel one;
            int rc = make_simple_kernel(&fk, dtype, true, &one);
            if (rc) return rc;
            double sc = fk.scale;                  // (scale * phi)^1 only: a factor's scale multiplies the term once
            out->coef[t] *= sc;
            out->ffam[nf] = fk.family;
            out->fkp[nf] = one.kp;
            out->fkp[nf].scale = 1.0;
            ++out->nfac[t]; ++nf;
        }
    }
    return COVGRAM_OK;
}

int make_host_kernel(const covgram_kernel* k, int dtype, bool for_gradient, HostKernel* out) {
    CG_REQUIRE(k != nullptr, COVGRAM_EINVAL, "kernel is NULL");
    if (k->family == COVGRAM_COMPOSITE) return make_composite_kernel((const covgram_kernel_composite*)k, dtype, out);
    return make_simple_kernel(k, dtype, for_gradient, out);
}

static int make_simple_kernel(const covgram_kernel* k, int dtype, bool for_gradient, HostKernel* out) {
    CG_REQUIRE(k->family >= 0 && k->family < COVGRAM_NFAMILY, COVGRAM_EUNSUPPORTED, "unknown kernel family %d", k->family);
    if (k->family == COVGRAM_MATERNP && k->p == 0 && !for_gradient) {
        // MaternP(0)(s) = exp(-sqrt(s)) IS the Exponential profile (src/stationary.jl:117-158 with p = 0; tests/golden closed forms): the
        // value-only kernels run it as such instead of through the general MaternP body (fp32 n = 32768: 436 -> 293 us, profiles/r04_sym32_sweep.txt)
        covgram_kernel e = *k;
        e.family = COVGRAM_EXP; e.p = 0; e.param = 0.0;
        return make_simple_kernel(&e, dtype, for_gradient, out);
    }
    const bool dotfam = (k->family == COVGRAM_DOT || k->family == COVGRAM_EXPDOT || k->family == COVGRAM_ASINDOT);
    const int trait = dotfam ? COVGRAM_DOTPRODUCT : COVGRAM_ISOTROPIC;
    CG_REQUIRE(k->trait == trait, COVGRAM_EINVAL, "kernel trait %d does not match family %d (input_trait would be %d)",
               k->trait, k->family, trait);
    CG_REQUIRE(k->power >= 1, COVGRAM_EINVAL, "Power exponent must be >= 1 (got %d)", k->power);
    CG_REQUIRE(k->lengthscale > 0, COVGRAM_EINVAL, "DomainError: l = %g is non-positive", k->lengthscale);
    CG_REQUIRE(!(dotfam && k->lengthscale != 1.0), COVGRAM_EINVAL, "Lengthscale applies to isotropic kernels only");
    memset(out, 0, sizeof(*out));
    out->k = *k;
    out->tu_family = k->family;
    KParams<double>& kp = out->kp;
    const double inv_l = 1.0 / k->lengthscale;
    kp.scale = k->scale;
    kp.power = k->power;
    kp.p = 0;
    kp.gamma = dotfam ? 1.0 : inv_l;
    kp.c0 = 0;
    out->eq_folded = false;
    const double LOG2E = 1.4426950408889634074;
    switch (k->family) {
        case COVGRAM_EQ:
            kp.c0 = -0.5 * LOG2E;  // exp(-s/2) = exp2(c0 s)
            if (!for_gradient) {     // dense path: fold sqrt(log2(e)/2) into the coordinate pre-scale
                kp.gamma = inv_l * sqrt(0.5 * LOG2E);
                out->eq_folded = true;
            }
            break;
        case COVGRAM_EXP:
            // fp32 dense path: fold log2(e) into the coordinate pre-scale, exp(-r) = exp2(-sqrt(s')) (profiles.hpp: dense_folded)
            if (!for_gradient && dtype == COVGRAM_F32) { kp.gamma = inv_l * LOG2E; out->eq_folded = true; }
            break;
        case COVGRAM_RQ:
            CG_REQUIRE(k->param > 0, COVGRAM_EINVAL, "DomainError: alpha not positive");
            kp.param = k->param;
            kp.c0 = 1.0 / (2.0 * k->param);
            break;
        case COVGRAM_GAMMAEXP:
            CG_REQUIRE(k->param >= 0 && k->param <= 2, COVGRAM_EINVAL, "DomainError: gamma not in [0,2]");
            kp.param = 0.5 * k->param;
            break;
        case COVGRAM_IMQ:
            kp.param = k->param * k->param;
            break;
        case COVGRAM_MATERNP: {
            CG_REQUIRE(k->p >= 0, COVGRAM_EINVAL, "DomainError: p = %d is negative", k->p);
            CG_REQUIRE(k->p <= MAXP, COVGRAM_EUNSUPPORTED, "MaternP order %d exceeds the compiled maximum %d", k->p, MAXP);
            const int p = k->p;
            kp.p = p;
            kp.mp_c = 2.0 * p + 1.0;
            long double h[MAXP + 1], d[MAXP + 1];
            maternp_poly(p, h);
            for (int m = 0; m <= p; ++m) kp.h0[m] = (double)h[m];
            if (p >= 1) { maternp_poly(p - 1, h); for (int m = 0; m <= p - 1; ++m) kp.h1[m] = (double)h[m]; }
            if (p >= 2) { maternp_poly(p - 2, h); for (int m = 0; m <= p - 2; ++m) kp.h2[m] = (double)h[m]; }
            kp.ty[0] = 1.0;
            if (p >= 1) {
                maternp_derivs0(p, d);
                for (int i = 1; i <= p; ++i) kp.ty[i] = (double)(d[i] / factl(i));
                kp.mp_d1 = (double)d[1];
                kp.mp_d2 = (p >= 2) ? (double)d[2] : 0.0;
                const double eps = (dtype == COVGRAM_F64) ? 2.220446049250313e-16 : 1.1920928955078125e-07;
                kp.mp_bound = pow(eps, 1.0 / p);     // src/stationary.jl:135
            } else {
                kp.mp_bound = 0.0;                   // eps^(1/0) = 0: never taken
            }
            if (!for_gradient) {
                // dense path: fold sqrt(2p+1) and log2(e) into the coordinate pre-scale, s' = (2p+1) log2(e)^2 s / l^2, so that
                // sqrt(s') = r log2(e): H_p(r) = exp2(-sqrt(s')) sum_m (h_m / log2(e)^m) sqrt(s')^m, Taylor in s'
                const double f2 = kp.mp_c * LOG2E * LOG2E;
                kp.gamma = inv_l * sqrt(f2);
                double li = 1.0, fi = 1.0;
                for (int i = 0; i <= p; ++i) { kp.h0[i] *= li; kp.ty[i] *= fi; li /= LOG2E; fi /= f2; }
                kp.mp_bound *= f2;
                out->eq_folded = true;
            }
            break;
        }
        case COVGRAM_MATERN: {
            // Matern(nu), real nu > 0 (src/stationary.jl:87-114): phi = C r^nu K_nu(r), r = sqrt(2 nu s), C = 2^(1-nu)/Gamma(nu);
            // below taylor_bound the reference's polynomial (:100-110).  K_a by Temme's series (x < 2) / Steed's continued fraction
            // (x >= 2) at order mu = a - round(a) and upward recurrence; the mu-only constants are computed here, per order:
            // h0 <- order nu (value), h1 <- |nu - 1| (phi'), h2 <- |nu - 2| (phi'').
            const double nu = k->param;
            CG_REQUIRE(nu > 0, COVGRAM_EINVAL, "DomainError: nu = %g is negative", nu);
            CG_REQUIRE(nu <= 64, COVGRAM_EUNSUPPORTED, "Matern: nu = %g exceeds 64", nu);
            kp.param = nu;
            kp.mp_c = 2.0 * nu;
            const double eps = (dtype == COVGRAM_F64) ? 2.220446049250313e-16 : 1.1920928955078125e-07;
            kp.mp_bound = (nu > 2) ? sqrt(eps) : ((nu > 1) ? eps : 0.0);
            kp.ty[0] = 1.0;
            kp.ty[1] = (nu > 1) ? nu / (2 * (1 - nu)) : 0.0;
            kp.ty[2] = (nu > 2) ? nu * nu / (8 * (2 - 3 * nu + nu * nu)) : 0.0;
            kp.c0 = exp2(1.0 - nu) / tgamma(nu);                      // C
            auto order = [&](double a, double* h) {                   // h[0..6] = nl, mu, gam1, gam2, 1/Gamma(1+mu), 1/Gamma(1-mu), pi mu / sin(pi mu)
                const double nl = floor(a + 0.5);
                const long double mu = (long double)a - nl;
                const long double gp = 1.0L / tgammal(1.0L + mu), gm = 1.0L / tgammal(1.0L - mu);
                long double g1;
                if (fabsl(mu) < 0.01L) {                              // (1/Gamma(1-mu) - 1/Gamma(1+mu)) / (2 mu) from the series of 1/Gamma
                    const long double m2 = mu * mu;
                    g1 = -(0.57721566490153286061L + m2 * (-0.042002635034095235529L + m2 * (-0.042197734555544336748L + m2 * 0.0072189432466630995424L)));
                } else {
                    g1 = (gm - gp) / (2 * mu);
                }
                const long double pimu = 3.14159265358979323846264338327950288L * mu;
                h[0] = nl; h[1] = (double)mu; h[2] = (double)g1; h[3] = (double)((gm + gp) / 2); h[4] = (double)gp; h[5] = (double)gm;
                h[6] = (fabsl(pimu) < 1e-18L) ? 1.0 : (double)(pimu / sinl(pimu));
            };
            order(nu, kp.h0); order(fabs(nu - 1), kp.h1); order(fabs(nu - 2), kp.h2);
            break;
        }
        default: break;
    }
    kp.gamma2 = kp.gamma * kp.gamma;
    return COVGRAM_OK;
}

int ws_reserve(covgram_ctx* ctx, int slot, size_t bytes, void** out) {
    Workspace& w = ctx->ws[slot];
    if (w.bytes < bytes) {
        if (w.ptr) {
            CG_CHECK_HIP(hipStreamSynchronize(ctx->stream));
            CG_CHECK_HIP(hipFree(w.ptr));
            w.ptr = nullptr; w.bytes = 0;
        }
        size_t want = std::max(bytes, (size_t)1 << 20);
        want = (want + 255) & ~(size_t)255;
        hipError_t e = hipMalloc(&w.ptr, want);
        if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return COVGRAM_ENOMEM; }
        w.bytes = want;
    }
    *out = w.ptr;
    return COVGRAM_OK;
}

int tickets_reserve(covgram_ctx* ctx, size_t count, unsigned** out) {
    if (ctx->tickets_cap < count) {
        if (ctx->tickets) {
            CG_CHECK_HIP(hipStreamSynchronize(ctx->stream));
            CG_CHECK_HIP(hipFree(ctx->tickets));
            ctx->tickets = nullptr; ctx->tickets_cap = 0;
        }
        const size_t want = std::max<size_t>(count, 16384);
        hipError_t e = hipMalloc((void**)&ctx->tickets, want * sizeof(unsigned));
        if (e != hipSuccess) { ctx->tickets = nullptr; set_error("hipMalloc(%zu) failed: %s", want * sizeof(unsigned), hipGetErrorString(e)); return COVGRAM_ENOMEM; }
        CG_CHECK_HIP(hipMemsetAsync(ctx->tickets, 0, want * sizeof(unsigned), ctx->stream));
        ctx->tickets_cap = want;
    }
    *out = ctx->tickets;
    return COVGRAM_OK;
}

// dense instantiation: generic over family via a uniform switch (HBM-write-bound, n*m*sizeof(T) out)
grad_wide_launch_fn grad_wide_launcher(int family) {
    static const grad_wide_launch_fn t[NUM_TU_FAMILIES] = {
        launch_grad_wide_family_0, launch_grad_wide_family_1, launch_grad_wide_family_2, launch_grad_wide_family_3,
        launch_grad_wide_family_4, launch_grad_wide_family_5, launch_grad_wide_family_6, launch_grad_wide_family_7,
        launch_grad_wide_family_8, launch_grad_wide_family_9, launch_grad_wide_family_10, launch_grad_wide_family_11, launch_grad_wide_family_12};
    return (family >= 0 && family < NUM_TU_FAMILIES) ? t[family] : nullptr;
}

std::pair<hipEvent_t, hipEvent_t>* timer_next(covgram_ctx* ctx) {
    if (!ctx->time_kernels) return nullptr;
    if (ctx->timers_used == ctx->timers.size()) {
        if (ctx->timers.size() >= 8192) return nullptr;
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return nullptr;
        ctx->timers.emplace_back(a, b);
    }
    return &ctx->timers[ctx->timers_used++];
}

template <typename T>
__global__ __launch_bounds__(256) void matrix_kernel(const T* __restrict__ X, int64_t n, const T* __restrict__ Y, int64_t m,
                                                     int32_t d, T* __restrict__ out, int64_t ldo, int family, T scale,
                                                     const KParams<T> kp) {
    // block = 256 rows × 1 column strip of 16 columns; lanes along rows -> coalesced column-major stores
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t jb = (int64_t)blockIdx.y * 16;
    if (i >= n) return;
    const bool iso = (family != COVGRAM_DOT && family != COVGRAM_EXPDOT && family != COVGRAM_ASINDOT);
    const T* xi = X + i * (int64_t)d;
    for (int64_t j = jb; j < jb + 16 && j < m; ++j) {
        const T* yj = Y + j * (int64_t)d;
        T s = (T)0;
        for (int l = 0; l < d; ++l) {
            if (iso) { T r = (xi[l] - yj[l]) * kp.gamma; s = cg_fma(r, r, s); }
            else s = cg_fma(xi[l], yj[l], s);
        }
        out[i + j * ldo] = scale * phi_any<T>(family, s, kp);
    }
}

// The same entries with the row x_i held in registers (d <= DM) and a strip of 64 columns per thread: the generic kernel above re-reads
// x_i from memory for every entry — at d = 32 that is 32 vector loads per entry (62 GB/s written for a 16384^2 fp32 tile, against
// 1.8 TB/s at d = 3).  Same arithmetic in the same order (entries are bit-identical); y_j is wave-uniform (scalar loads).
// EXPR: composite kernels (ExprParams), else one profile through phi_any.
template <typename T, int DM, bool EXPR, typename PT, int VR = 1>
__global__ __launch_bounds__(256) void matrix_reg_kernel(const T* __restrict__ X, int64_t n, const T* __restrict__ Y, int64_t m,
                                                         int32_t d, T* __restrict__ out, int64_t ldo, int family_or_iso, T scale, const PT kp) {
    // VR consecutive rows per thread (round 5): one 16-byte streaming store per column instead of VR narrow ones — Matrix(G) is a gigabyte written once
    // (fp32, n = 16384: 3.7 -> TB/s figures in profiles/r05_matrix_bench.txt); VR > 1 needs n, ldo multiples of VR and a 16-byte aligned out
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VR;
    const int64_t jb = (int64_t)blockIdx.y * 64;
    if (i >= n) return;
    const bool iso = EXPR ? (family_or_iso != 0)
                          : (family_or_iso != COVGRAM_DOT && family_or_iso != COVGRAM_EXPDOT && family_or_iso != COVGRAM_ASINDOT);
    T x[VR][DM];
#pragma unroll
    for (int r = 0; r < VR; ++r)
#pragma unroll
        for (int l = 0; l < DM; ++l) x[r][l] = (l < d) ? X[(i + r) * (int64_t)d + l] : (T)0;
    const int64_t jend = (jb + 64 < m) ? jb + 64 : m;
    T gam = (T)1;
    if constexpr (!EXPR) gam = kp.gamma;
    // the padded dimensions l >= d contribute exact zeros (x = y = 0: r = 0, fma(0, 0, s) = s): no branch inside the entry loop;
    // their y is a scalar select after an in-bounds (clamped) load
    auto entry = [&](const T* __restrict__ yj, auto full, T (&s)[VR]) {
#pragma unroll
        for (int r = 0; r < VR; ++r) s[r] = (T)0;
#pragma unroll
        for (int l = 0; l < DM; ++l) {
            T yl;
            if constexpr (decltype(full)::value) yl = yj[l];
            else { const int lc = l < d ? l : d - 1; const T yv = yj[lc]; yl = l < d ? yv : (T)0; }
#pragma unroll
            for (int r = 0; r < VR; ++r) {
                if (iso) { T q = x[r][l] - yl; if constexpr (!EXPR) q *= gam; s[r] = cg_fma(q, q, s[r]); }
                else s[r] = cg_fma(x[r][l], yl, s[r]);
            }
        }
    };
    typedef T VT __attribute__((ext_vector_type(VR)));
    // the family switch sits OUTSIDE the column loop: each case is a short loop with one profile inlined (the switch inside — phi_any per
    // entry — put every profile, the Bessel series included, into one loop body: 180 SGPR spills, y fetched a dword at a time)
    auto strip = [&](auto famc, auto full) {
        constexpr int F = decltype(famc)::value;
        for (int64_t j = jb; j < jend; ++j) {
            T s[VR], v[VR];
            entry(Y + j * (int64_t)d, full, s);
#pragma unroll
            for (int r = 0; r < VR; ++r) {
                if constexpr (EXPR) v[r] = scale * (iso ? expr_value<T, true>(s[r], kp) : expr_value<T, false>(s[r], kp));
                else {
                    T w;
                    if constexpr (F == COVGRAM_DOT) w = s[r];
                    else if constexpr (F == COVGRAM_CONSTANT) w = (T)1;
                    else w = Phi<F, T, false>::eval(s[r], kp);
                    if (kp.power != 1) w = ipow(w, kp.power);
                    v[r] = scale * w;
                }
            }
            if constexpr (VR == 1) __builtin_nontemporal_store(v[0], out + i + j * ldo);
            else {
                VT vv;
#pragma unroll
                for (int r = 0; r < VR; ++r) vv[r] = v[r];
                __builtin_nontemporal_store(vv, reinterpret_cast<VT*>(out + i + j * ldo));
            }
        }
    };
    auto run = [&](auto famc) { if (d == DM) strip(famc, std::true_type()); else strip(famc, std::false_type()); };
    if constexpr (EXPR) run(std::integral_constant<int, 0>());
    else switch (family_or_iso) {
#define CG_FAMCASE(F) case F: run(std::integral_constant<int, F>()); break;
        CG_FAMCASE(COVGRAM_EQ) CG_FAMCASE(COVGRAM_EXP) CG_FAMCASE(COVGRAM_RQ) CG_FAMCASE(COVGRAM_GAMMAEXP) CG_FAMCASE(COVGRAM_CAUCHY)
        CG_FAMCASE(COVGRAM_IMQ) CG_FAMCASE(COVGRAM_MATERNP) CG_FAMCASE(COVGRAM_DOT) CG_FAMCASE(COVGRAM_EXPDOT) CG_FAMCASE(COVGRAM_MATERN)
        CG_FAMCASE(COVGRAM_ASINDOT)
#undef CG_FAMCASE
    }
}

template <typename T>
__global__ __launch_bounds__(256) void matrix_expr_kernel(const T* __restrict__ X, int64_t n, const T* __restrict__ Y, int64_t m,
                                                          int32_t d, T* __restrict__ out, int64_t ldo, int iso, T scale,
                                                          const ExprParams<T> ep) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t jb = (int64_t)blockIdx.y * 16;
    if (i >= n) return;
    const T* xi = X + i * (int64_t)d;
    for (int64_t j = jb; j < jb + 16 && j < m; ++j) {
        const T* yj = Y + j * (int64_t)d;
        T s = (T)0;
        for (int l = 0; l < d; ++l) {
            if (iso) { T r = xi[l] - yj[l]; s = cg_fma(r, r, s); }
            else s = cg_fma(xi[l], yj[l], s);
        }
        out[i + j * ldo] = scale * (iso ? expr_value<T, true>(s, ep) : expr_value<T, false>(s, ep));
    }
}

// staging helpers for loc == HOST ------------------------------------------------------------------
struct Staged {
    void* dev = nullptr;
    void* host = nullptr;
    size_t bytes = 0;
};

}  // namespace covgram

using namespace covgram;

// ================================================================================================
// C ABI
// ================================================================================================
// the automatic choice of the broadcast kernel by padded dimension (tools/c4_bcast_ab.py, profiles/r04_c4_bcast_ab.txt: n = 16384, EQ,
// scalar stream -> broadcast, four waves per workgroup: d = 8 x0.91, 16 x0.95, 24 x1.23, 32 (C4) x1.25, 48 x1.83)
#ifndef GRAD_BCAST_AUTO
#define GRAD_BCAST_AUTO(D) ((D) >= 24 ? 4 : 0)
#endif

extern "C" {

int covgram_version(void) { return COVGRAM_VERSION; }
int covgram_sizeof_kernel(void) { return (int)sizeof(covgram_kernel); }
int covgram_sizeof_composite(void) { return (int)sizeof(covgram_kernel_composite); }
const char* covgram_last_error(void) { return g_err; }

int covgram_device_count(int* count) {
    CG_REQUIRE(count != nullptr, COVGRAM_EINVAL, "count is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { c = 0; (void)hipGetLastError(); }
    *count = c;
    return COVGRAM_OK;
}

int covgram_ctx_create(covgram_ctx** out, int device_id, void* hip_stream) {
    CG_REQUIRE(out != nullptr, COVGRAM_EINVAL, "ctx out pointer is NULL");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device visible (hipGetDeviceCount: %s) — libcovgram has no CPU fallback",
                  e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        return COVGRAM_ENODEVICE;
    }
    CG_REQUIRE(device_id >= 0 && device_id < count, COVGRAM_EINVAL, "device_id %d out of range [0,%d)", device_id, count);
    ::covgram::DeviceGuard _cg_dev(device_id);
    CG_REQUIRE(_cg_dev.ok(), COVGRAM_EHIP, "hipSetDevice(%d) failed", device_id);
    hipDeviceProp_t prop;
    CG_CHECK_HIP(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; libcovgram is built for gfx950 (MI355X) only", device_id, prop.gcnArchName);
        return COVGRAM_ENODEVICE;
    }
    covgram_ctx* c = new covgram_ctx();
    c->device = device_id;
    c->num_cus = prop.multiProcessorCount;
    c->stream = (hipStream_t)hip_stream;   // NULL = the device's default (null) stream, like any HIP API
    c->own_stream = false;
    *out = c;
    return COVGRAM_OK;
}

int covgram_ctx_destroy(covgram_ctx* ctx) {
    if (!ctx) return COVGRAM_OK;
    CG_REQUIRE(ctx->live_handles == 0, COVGRAM_EINVAL, "ctx destroyed with %d live handles", ctx->live_handles);
    ::covgram::DeviceGuard _cg_dev(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx_blas_destroy(ctx);
    (void)comm_destroy(ctx);
    for (auto& w : ctx->ws) if (w.ptr) (void)hipFree(w.ptr);
    if (ctx->sym_map) (void)hipFree(ctx->sym_map);
    if (ctx->stamp_buf) (void)hipFree(ctx->stamp_buf);
    if (ctx->tickets) (void)hipFree(ctx->tickets);
    for (auto& t : ctx->timers) { (void)hipEventDestroy(t.first); (void)hipEventDestroy(t.second); }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return COVGRAM_OK;
}

int covgram_ctx_set_stream(covgram_ctx* ctx, void* hip_stream) {
    CG_REQUIRE(ctx != nullptr, COVGRAM_EINVAL, "ctx is NULL");
    // workspaces are reused across calls: order the new stream after everything queued on the old one
    if ((hipStream_t)hip_stream != ctx->stream) CG_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stream = (hipStream_t)hip_stream;
    return COVGRAM_OK;
}

int covgram_ctx_get_stream(covgram_ctx* ctx, void** hip_stream) {
    CG_REQUIRE(ctx && hip_stream, COVGRAM_EINVAL, "NULL argument");
    *hip_stream = (void*)ctx->stream;
    return COVGRAM_OK;
}

int covgram_ctx_set_option(covgram_ctx* ctx, const char* key, int64_t value) {
    CG_REQUIRE(ctx && key, COVGRAM_EINVAL, "NULL argument");
    if (!strcmp(key, "dense_variant")) ctx->dense_variant = value;
    else if (!strcmp(key, "toeplitz_fused")) ctx->toeplitz_fused = value;
    else if (!strcmp(key, "toeplitz_colfft")) ctx->toeplitz_colfft = value;
    else if (!strcmp(key, "toeplitz_persist")) ctx->toeplitz_persist = value;
    else if (!strcmp(key, "kron_fill")) ctx->kron_fill = value;
    else if (!strcmp(key, "mfma_f16")) ctx->mfma_f16 = value;
    else if (!strcmp(key, "mfma_gate_pct")) { CG_REQUIRE(value >= 1 && value <= 100, COVGRAM_EINVAL, "mfma_gate_pct = %lld: 1..100", (long long)value); ctx->mfma_gate_pct = value; }
    else if (!strcmp(key, "mfma_fuse_w")) ctx->mfma_fuse_w = value;
    else if (!strcmp(key, "toeplitz_real_spectrum")) ctx->toeplitz_real_spectrum = value;
    else if (!strcmp(key, "rows_per_lane")) ctx->rows_per_lane = value;
    else if (!strcmp(key, "jsplit")) ctx->jsplit = value;
    else if (!strcmp(key, "target_wgs")) ctx->target_wgs = value;
    else if (!strcmp(key, "grad_keep_r")) ctx->grad_keep_r = value;
    else if (!strcmp(key, "grad_expand")) ctx->grad_expand = value;
    else if (!strcmp(key, "grad_bcast")) ctx->grad_bcast = value;
    else if (!strcmp(key, "lds_pad")) ctx->lds_pad = value;
    else if (!strcmp(key, "mfma_lds")) ctx->mfma_lds = value;
    else if (!strcmp(key, "mfma_sym")) ctx->mfma_sym = value;
    else if (!strcmp(key, "dense_sym")) ctx->dense_sym = value;
    else if (!strcmp(key, "dense_bcast")) ctx->dense_bcast = value;
    else if (!strcmp(key, "mfma_stamp")) ctx->mfma_stamp = value;
    else if (!strcmp(key, "inkernel_reduce")) ctx->inkernel_reduce = value;
    else if (!strcmp(key, "matrix_variant")) ctx->matrix_variant = value;
    else if (!strcmp(key, "mfma_mrhs")) ctx->mfma_mrhs = value;
    else if (!strcmp(key, "composite_termwise")) ctx->composite_termwise = value;
    else if (!strcmp(key, "sum_fused")) ctx->sum_fused = value;
    else if (!strcmp(key, "lowrank_reverse")) ctx->lowrank_reverse = value;
    else if (!strcmp(key, "mfma_sym_rt")) ctx->mfma_sym_rt = value;
    else if (!strcmp(key, "mfma_sym_st")) ctx->mfma_sym_st = value;
    else if (!strcmp(key, "lowrank_wgs")) ctx->lowrank_wgs = value;
    else if (!strcmp(key, "time_kernels")) { ctx->time_kernels = value; ctx->timers_used = 0; }
    else { set_error("unknown option '%s'", key); return COVGRAM_EINVAL; }
    return COVGRAM_OK;
}

int covgram_ctx_get_info(covgram_ctx* ctx, const char* key, int64_t* value) {
    CG_REQUIRE(ctx && key && value, COVGRAM_EINVAL, "NULL argument");
    if (!strcmp(key, "last_dense_path")) *value = ctx->last_dense_path;
    else if (!strcmp(key, "last_jsplit")) *value = ctx->last_jsplit;
    else if (!strcmp(key, "last_kron_path")) *value = ctx->last_kron_path;
    else if (!strcmp(key, "last_mfma_lds")) *value = ctx->last_mfma_lds;
    else if (!strcmp(key, "last_mfma_sym")) *value = ctx->last_mfma_sym;
    else if (!strcmp(key, "last_dense_sym")) *value = ctx->last_dense_sym;
    else if (!strcmp(key, "last_dense_bcast")) *value = ctx->last_dense_bcast;
    else if (!strcmp(key, "last_mfma_f16")) *value = ctx->last_mfma_f16;
    else if (!strcmp(key, "last_inkernel_reduce")) *value = ctx->last_inkernel_reduce;
    else if (!strcmp(key, "last_grad_expand")) *value = ctx->last_grad_expand;
    else if (!strcmp(key, "last_grad_bcast")) *value = ctx->last_grad_bcast;
    else if (!strcmp(key, "last_sum_fused")) *value = ctx->last_sum_fused;
    else if (!strcmp(key, "last_mfma_instance")) *value = ctx->last_mfma_instance;
    else if (!strcmp(key, "last_mfma_sym_rt")) *value = ctx->last_mfma_sym_rt;
    else if (!strcmp(key, "num_cus")) *value = ctx->num_cus;
    else if (!strcmp(key, "last_clock_khz")) {
        // median over the workgroups of the last stamped launch of (shader cycles) / (100 MHz ticks) x 100 MHz, in kHz; 0: none
        *value = 0;
        if (ctx->stamp_count > 0 && ctx->stamp_buf) {
            CG_DEVICE(ctx);
            CG_CHECK_HIP(hipStreamSynchronize(ctx->stream));
            std::vector<unsigned long long> h(ctx->stamp_count * 4);
            CG_CHECK_HIP(hipMemcpy(h.data(), ctx->stamp_buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            std::vector<double> khz;
            for (size_t w = 0; w < ctx->stamp_count; ++w) {
                const double dc = (double)(h[4 * w + 2] - h[4 * w]), dr = (double)(h[4 * w + 3] - h[4 * w + 1]);
                if (h[4 * w + 3] > h[4 * w + 1] && dr >= 100.0) khz.push_back(dc / dr * 1.0e5);   // >= 1 us of work
            }
            if (!khz.empty()) { std::nth_element(khz.begin(), khz.begin() + khz.size() / 2, khz.end()); *value = (int64_t)khz[khz.size() / 2]; }
        }
    }
    else { set_error("unknown info key '%s'", key); return COVGRAM_EINVAL; }
    return COVGRAM_OK;
}

int covgram_ctx_kernel_time(covgram_ctx* ctx, double* total_ms, int64_t* launches, int32_t reset) {
    CG_REQUIRE(ctx && total_ms && launches, COVGRAM_EINVAL, "NULL argument");
    CG_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    double tot = 0;
    for (size_t i = 0; i < ctx->timers_used; ++i) {
        float ms = 0;
        CG_CHECK_HIP(hipEventElapsedTime(&ms, ctx->timers[i].first, ctx->timers[i].second));
        tot += ms;
    }
    *total_ms = tot;
    *launches = (int64_t)ctx->timers_used;
    if (reset) ctx->timers_used = 0;
    return COVGRAM_OK;
}

int covgram_sync(covgram_ctx* ctx) {
    CG_REQUIRE(ctx != nullptr, COVGRAM_EINVAL, "ctx is NULL");
    CG_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return COVGRAM_OK;
}

int covgram_points_create(covgram_ctx* ctx, covgram_points** out, const void* x, int64_t n, int32_t d, int32_t dtype,
                          int32_t loc) {
    CG_REQUIRE(ctx && out, COVGRAM_EINVAL, "NULL argument");
    CG_REQUIRE(n >= 0 && d >= 1, COVGRAM_EINVAL, "points: need n >= 0 and d >= 1 (got n=%lld d=%d)", (long long)n, d);
    CG_REQUIRE(dtype == COVGRAM_F32 || dtype == COVGRAM_F64, COVGRAM_EINVAL, "unknown dtype %d", dtype);
    CG_REQUIRE(x != nullptr || n == 0, COVGRAM_EINVAL, "points pointer is NULL");
    CG_DEVICE(ctx);
    covgram_points* p = new covgram_points();
    p->ctx = ctx; p->n = n; p->d = d; p->dtype = dtype;
    if (loc == COVGRAM_DEVICE) { p->dptr = const_cast<void*>(x); p->owns = false; }
    else {
        const size_t bytes = (size_t)n * d * dtype_size(dtype);
        if (bytes) {
            hipError_t e = hipMalloc(&p->dptr, bytes);
            if (e != hipSuccess) { delete p; set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return COVGRAM_ENOMEM; }
            e = hipMemcpyAsync(p->dptr, x, bytes, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { (void)hipFree(p->dptr); delete p; set_error("H2D copy failed: %s", hipGetErrorString(e)); return COVGRAM_EHIP; }
        }
        p->owns = true;
    }
    int rc = points_max_norm2(p);      // one reduction + 4-byte read-back: gates the matrix-core EQ path (dense_mfma.hip)
    if (rc) { if (p->owns && p->dptr) (void)hipFree(p->dptr); delete p; return rc; }
    ctx->live_handles++;
    *out = p;
    return COVGRAM_OK;
}

int covgram_points_slice(const covgram_points* parent, int64_t first, int64_t count, covgram_points** out) {
    CG_REQUIRE(parent && out, COVGRAM_EINVAL, "NULL argument");
    CG_REQUIRE(first >= 0 && count >= 0 && first + count <= parent->n, COVGRAM_EINVAL,
               "slice [%lld, %lld) outside [0, %lld)", (long long)first, (long long)(first + count), (long long)parent->n);
    covgram_points* p = new covgram_points(*parent);
    p->owns = false;
    p->owns_center = false;            // the centre buffer stays the parent's
    for (auto& f : p->frag) f = covgram_points::FragSlot{};   // a slice packs its own fragments
    p->frag_clock = 0;
    p->n = count;
    p->dptr = (char*)parent->dptr + (size_t)first * parent->d * dtype_size(parent->dtype);
    parent->ctx->live_handles++;
    *out = p;
    return COVGRAM_OK;
}

int covgram_points_destroy(covgram_points* p) {
    if (!p) return COVGRAM_OK;
    {
        ::covgram::DeviceGuard _cg_dev(p->ctx->device);            // (a finalizer may call this from any thread state)
        bool frags = false;
        for (const auto& f : p->frag) frags = frags || f.ptr;
        if (frags || (p->owns && p->dptr) || (p->owns_center && p->center_buf)) (void)hipStreamSynchronize(p->ctx->stream);
        for (auto& f : p->frag) if (f.ptr) (void)hipFree(f.ptr);
        if (p->owns && p->dptr) (void)hipFree(p->dptr);
        if (p->owns_center && p->center_buf) (void)hipFree(p->center_buf);
    }
    p->ctx->live_handles--;
    delete p;
    return COVGRAM_OK;
}

int covgram_points_info(const covgram_points* p, int64_t* n, int32_t* d, int32_t* dtype) {
    CG_REQUIRE(p != nullptr, COVGRAM_EINVAL, "points is NULL");
    if (n) *n = p->n;
    if (d) *d = p->d;
    if (dtype) *dtype = p->dtype;
    return COVGRAM_OK;
}

// ------------------------------------------------------------------------------------------------
static int check_pair(const covgram_ctx* ctx, const covgram_points* X, const covgram_points* Y) {
    CG_REQUIRE(ctx && X && Y, COVGRAM_EINVAL, "NULL argument");
    CG_REQUIRE(X->ctx == ctx && Y->ctx == ctx, COVGRAM_EINVAL, "points belong to a different ctx");
    CG_REQUIRE(X->dtype == Y->dtype, COVGRAM_EINVAL, "x and y have different dtypes");
    CG_REQUIRE(X->d == Y->d, COVGRAM_EINVAL, "DimensionMismatch: inputs have to have the same length: %d, %d", X->d, Y->d);
    return COVGRAM_OK;
}

// The direct-difference symmetric kernels (dense_sym_kernel: fp64, 64-row blocks; dense_sym32_kernel: fp32, 64 R-row blocks) on
// gramian(k, x) for `world` ranks (1: the whole triangle in one launch): a single profile without a Power wrapper, d <= 64, and the
// column-sum slab — ceil(row blocks / world) x npad scalars, held in workspace slot 4 for the life of the ctx — within `cap_gib` GiB.
// ONE predicate for covgram_mvm, covgram_mvm_sym_supported and covgram_mvm_sym_partial (the three disagreed on the cap).
struct DenseSymShape { int64_t blk, blocks, mine, npad; size_t slab_bytes; };
static bool dense_sym_shape(const HostKernel& hk, const covgram_points* X, int world, int cap_gib, DenseSymShape* out) {
    if (X->n <= 0 || hk.tu_family >= COVGRAM_NFAMILY || hk.k.power != 1 || world < 1) return false;
    if (X->d > kDims[kNumDims - 1]) return false;
    const int D = pad_dim(X->d);
    if (D < 0 || rows_per_lane_for(D) != 1) return false;
    DenseSymShape sh;
    const bool f64 = X->dtype == COVGRAM_F64;
    sh.blk = f64 ? 64 : 64 * dense_sym32_rows(D);
    sh.blocks = (X->n + sh.blk - 1) / sh.blk;
    sh.mine = (sh.blocks + world - 1) / world;
    sh.npad = f64 ? ((X->n + 63) / 64) * 64 : ((X->n + 511) / 512) * 512;
    sh.slab_bytes = (size_t)sh.mine * (size_t)sh.npad * dtype_size(X->dtype);
    if (sh.slab_bytes > ((size_t)cap_gib << 30)) return false;
    if (out) *out = sh;
    return true;
}

// choose the J split: enough workgroups to fill the chip, chunks aligned to the inner accumulation block
static void choose_split(const covgram_ctx* ctx, int64_t rowblocks, int64_t m, int64_t align, int64_t* jchunk, int* jsplit,
                         int64_t default_target = 0) {
    // ~128 single-wave workgroups per CU = 4 rounds of resident waves: measured on C2 (profiles/r01_quickbench_wg64.txt)
    // 3.63 ms at 8 waves per CU, 2.76 ms at 32, 2.63 ms at 64, 2.56 ms at 128 — dynamic rounds even out per-CU speed
    int64_t target = ctx->target_wgs > 0 ? ctx->target_wgs : (default_target > 0 ? default_target : (int64_t)ctx->num_cus * 128);
    int64_t js = ctx->jsplit > 0 ? ctx->jsplit : (target + rowblocks - 1) / std::max<int64_t>(rowblocks, 1);
    const int64_t maxsplit = std::max<int64_t>(1, m / align);
    js = std::max<int64_t>(1, std::min(js, maxsplit));
    int64_t jc = (m + js - 1) / js;
    jc = ((jc + align - 1) / align) * align;
    if (jc <= 0) jc = align;
    js = std::max<int64_t>(1, (m + jc - 1) / jc);
    *jchunk = jc;
    *jsplit = (int)js;
}

// A composite whose every term is ONE profile (times constants) is a plain Sum (src/algebra.jl:5-14): G = sum_t c_t G_t, so
// every MVM is the sum of the terms' MVMs, each on its own single-profile path (matrix cores, folded constants, lane-per-row
// gradient kernel) instead of the per-block interpreter of the composite kernels.  Fills the terms; false = not such a sum.
// Constant-only terms (k = c: G = c 1 1', src/stationary.jl:27-34) add up in *constant: their MVM is c * sum(a) on every row.
// Product terms (several profiles) stay composite, one single-term composite each, and run on the interpreter by themselves;
// when EVERY term is a product the whole composite stays on the interpreter (one pass shares the distances).
struct SumTerm {
    bool simple;
    covgram_kernel k;                     // simple
    covgram_kernel_composite c;           // product term
    const covgram_kernel* ptr() const { return simple ? &k : &c.head; }
};
static bool composite_sum_terms(const covgram_ctx* ctx, const covgram_kernel* k, int32_t loc, SumTerm* terms, int* nterms,
                                double* constant) {
    if (k == nullptr || k->family != COVGRAM_COMPOSITE || ctx->composite_termwise == 0) return false;
    const covgram_kernel_composite* c = (const covgram_kernel_composite*)k;
    if (c->nterms < 1 || c->nterms > COVGRAM_COMPOSITE_MAX_TERMS) return false;
    int nin = 0, nt = 0, nsimple = 0;
    *constant = 0.0;
    for (int t = 0; t < c->nterms; ++t) {
        double coef = c->head.scale;
        int profiles = 0;
        SumTerm& st = terms[nt];
        memset(&st.c, 0, sizeof(st.c));
        for (int f = 0; f < c->nfactors[t]; ++f, ++nin) {
            if (nin >= COVGRAM_COMPOSITE_MAX_FACTORS) return false;
            const covgram_kernel& fk = c->factors[nin];
            if (fk.family == COVGRAM_CONSTANT) { coef *= fk.scale; continue; }
            st.c.factors[profiles++] = fk;
        }
        if (profiles == 0) { *constant += coef; continue; }
        st.simple = profiles == 1;
        if (st.simple) { st.k = st.c.factors[0]; st.k.scale *= coef; ++nsimple; }
        else {
            st.c.head = c->head; st.c.head.scale = coef;
            st.c.nterms = 1; st.c.nfactors[0] = profiles;
        }
        ++nt;
    }
    if (nt == 0) return false;                                 // a profile term goes first: it applies beta
    if (nsimple == 0 && *constant == 0.0) return false;        // products only: one interpreter pass
    if (nsimple == 0 && nt > 1) return false;
    if (*constant != 0.0 && loc != COVGRAM_DEVICE) return false;   // the constant term works on device vectors only
    *nterms = nt;
    return true;
}

extern "C++" {
// y[i * ys + r * ldy] += coef * sum_j a[j * as + r * lda]   (as = ys = 1: dense; d + 1: the value row of value-gradient blocks)
template <typename T>
__global__ __launch_bounds__(1024) void const_sum_kernel(const T* __restrict__ a, int64_t m, int64_t lda, int64_t as, T* __restrict__ S) {
    const T* __restrict__ ar = a + (int64_t)blockIdx.x * lda;
    T s = 0;
    for (int64_t j = threadIdx.x; j < m; j += 1024) s += ar[j * as];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ T part[16];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        T tot = 0;
        for (int w = 0; w < 16; ++w) tot += part[w];           // fixed order: deterministic
        S[blockIdx.x] = tot;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void const_add_kernel(T* __restrict__ y, int64_t n, int64_t ldy, int64_t ys, const T* __restrict__ S, T coef) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[(int64_t)blockIdx.y * ldy + i * ys] += coef * S[blockIdx.y];
}
}  // extern "C++"
static int constant_term_mvm(covgram_ctx* ctx, int dtype, const void* a, int64_t m, int64_t lda, int64_t as, void* y, int64_t n, int64_t ldy,
                             int64_t ys, int nrhs, double coef) {
    if (n == 0 || m == 0 || coef == 0.0) return COVGRAM_OK;
    void* S;
    int rc = ws_reserve(ctx, 0, (size_t)nrhs * 8, &S);
    if (rc) return rc;
    const dim3 ag((unsigned)((n + 255) / 256), (unsigned)nrhs);
    if (dtype == COVGRAM_F32) {
        hipLaunchKernelGGL(const_sum_kernel<float>, dim3(nrhs), dim3(1024), 0, ctx->stream, (const float*)a, m, lda, as, (float*)S);
        hipLaunchKernelGGL(const_add_kernel<float>, ag, dim3(256), 0, ctx->stream, (float*)y, n, ldy, ys, (const float*)S, (float)coef);
    } else {
        hipLaunchKernelGGL(const_sum_kernel<double>, dim3(nrhs), dim3(1024), 0, ctx->stream, (const double*)a, m, lda, as, (double*)S);
        hipLaunchKernelGGL(const_add_kernel<double>, ag, dim3(256), 0, ctx->stream, (double*)y, n, ldy, ys, (const double*)S, coef);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("constant-term kernels failed: %s", hipGetErrorString(e)); return COVGRAM_EHIP; }
    return COVGRAM_OK;
}

int covgram_mvm(covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, const covgram_points* Y, const void* a,
                int64_t lda, void* y, int64_t ldy, int32_t nrhs, double alpha, double beta, int32_t loc) {
    int rc = check_pair(ctx, X, Y);
    if (rc) return rc;
    ctx->last_sum_fused = 0;
    {
        SumTerm terms[COVGRAM_COMPOSITE_MAX_TERMS];
        int nt = 0;
        double constant = 0.0;
        // (a Sum the one-pass matrix-core kernels take is NOT split: every term shares the pair's distance, as in the reference's
        //  per-pair evaluation src/algebra.jl:27-47 — dense_mfma.hip: sum_fused_applies)
        if (!sum_fused_applies(ctx, k, X, Y, nrhs) && composite_sum_terms(ctx, k, loc, terms, &nt, &constant)) {
            HostKernel chk;
            rc = make_host_kernel(k, X->dtype, false, &chk);   // the composite's own validation (traits, limits) still applies
            if (rc) return rc;
            for (int t = 0; t < nt; ++t) {
                rc = covgram_mvm(ctx, terms[t].ptr(), X, Y, a, lda, y, ldy, nrhs, alpha, t == 0 ? beta : 1.0, loc);
                if (rc) return rc;
            }
            return constant_term_mvm(ctx, X->dtype, a, Y->n, lda, 1, y, X->n, ldy, 1, nrhs, alpha * constant);
        }
    }
    CG_REQUIRE(nrhs >= 1, COVGRAM_EINVAL, "nrhs must be >= 1");
    const int64_t n = X->n, m = Y->n;
    CG_REQUIRE(lda >= m && ldy >= n, COVGRAM_EINVAL, "DimensionMismatch: lda=%lld < m=%lld or ldy=%lld < n=%lld",
               (long long)lda, (long long)m, (long long)ldy, (long long)n);
    CG_REQUIRE((a != nullptr || m == 0) && (y != nullptr || n == 0), COVGRAM_EINVAL, "a or y is NULL");
    const int dtype = X->dtype;
    const size_t ts = dtype_size(dtype);
    HostKernel hk;
    rc = make_host_kernel(k, dtype, false, &hk);
    if (rc) return rc;
    // d <= 64: x_i lives in registers (dense_mvm.hpp); beyond that the chunked kernel of dense_wide.hpp (any d)
    const bool wide = X->d > kDims[kNumDims - 1];
    const int D = wide ? ((X->d + 31) / 32) * 32 : pad_dim(X->d);
    dense_launch_fn launch = wide ? dense_wide_launcher(hk.tu_family) : dense_launcher(hk.tu_family);
    CG_DEVICE(ctx);
    if (n == 0) return COVGRAM_OK;

    // host staging ---------------------------------------------------------------------------------
    const void* a_dev = a;
    void* y_dev = y;
    int64_t lda_d = lda, ldy_d = ldy;
    if (loc == COVGRAM_HOST) {
        void *sa, *sy;
        rc = ws_reserve(ctx, 2, (size_t)std::max<int64_t>(m, 1) * nrhs * ts, &sa); if (rc) return rc;
        rc = ws_reserve(ctx, 3, (size_t)n * nrhs * ts, &sy); if (rc) return rc;
        if (m > 0)
            CG_CHECK_HIP(hipMemcpy2DAsync(sa, (size_t)m * ts, a, (size_t)lda * ts, (size_t)m * ts, nrhs, hipMemcpyHostToDevice, ctx->stream));
        if (beta != 0.0)
            CG_CHECK_HIP(hipMemcpy2DAsync(sy, (size_t)n * ts, y, (size_t)ldy * ts, (size_t)n * ts, nrhs, hipMemcpyHostToDevice, ctx->stream));
        a_dev = sa; y_dev = sy; lda_d = m; ldy_d = n;
    }

    const double alpha_eff = alpha * hk.kp.scale;
    const int R = rows_per_lane_for(D);
    const int PKN = (dtype == COVGRAM_F32) ? 2 : 1;      // fp32 packs two columns per stream element
    const int64_t rows_per_wg = (int64_t)DENSE_THREADS * R;
    const int64_t rowblocks = (n + rows_per_wg - 1) / rows_per_wg;
    const int64_t npad = rowblocks * rows_per_wg;

    // isotropic kernels work relative to the column side's centre (common.hpp: covgram_points::center)
    const void* Cn = (hk.k.trait == COVGRAM_ISOTROPIC) ? Y->center : nullptr;
    // Gramian(Dot(), x, y) = X Y' (src/gramian.jl:23,150-151): the reference goes entry by entry through the generic loop; the
    // product is X (Y' a), two O((n + m) d) streaming passes (lowrank.hip).  dense_variant = 1 / 2 keep the entry-by-entry kernels.
    const bool dotfac = m > 0 && hk.tu_family == COVGRAM_DOT && hk.k.power == 1 && ctx->dense_variant == 0 && ctx->sym_part_world == 0 &&
                        (size_t)X->d * (size_t)nrhs * ts <= 65536;
    if (dotfac) {
        rc = mvm_dot_factored(ctx, hk, X, Y, a_dev, lda_d, y_dev, ldy_d, nrhs, alpha, beta); if (rc) return rc;
        ctx->last_dense_path = 4;
    }
    bool mfma = !dotfac && m > 0 && mfma_eq_eligible(ctx, hk, X, Y, nrhs);
    const bool sym = mfma && mfma_eq_sym_eligible(ctx, hk, X, Y, nrhs);
    ctx->last_mfma_sym = sym ? 1 : 0;
    if (dotfac) mfma = true;                                     // (handled: skip the kernels below)
    else if (sym) { rc = mvm_eq_mfma_sym(ctx, hk, X, (const float*)a_dev, (float*)y_dev, alpha, beta); if (rc) return rc; }
    else if (mfma) { rc = mvm_eq_mfma(ctx, hk, X, Y, (const float*)a_dev, (float*)y_dev, alpha, beta); if (rc) return rc; }
    else if (m > 0 && mfma_gen_eligible(ctx, hk, X, Y, nrhs)) {
        mfma = true;
        if (mfma_gen_sym_eligible(ctx, hk, X, Y, nrhs)) {              // gramian(k, x): upper triangle once, any matrix-core profile
            ctx->last_mfma_sym = 1;
            rc = mvm_eq_mfma_sym(ctx, hk, X, (const float*)a_dev, (float*)y_dev, alpha, beta, 0, 1, k); if (rc) return rc;
        } else {
            rc = mvm_mfma_gen(ctx, k, X, Y, (const float*)a_dev, lda_d, (float*)y_dev, ldy_d, nrhs, alpha, beta); if (rc) return rc;
        }
    }
    if (m > 0 && !dotfac) ctx->last_dense_path = mfma ? 2 : (wide ? 3 : 1);
    // fp64 gramian(k, x) * a on the direct-difference path: the upper triangle once (dense_sym_kernel, dense_mvm.hpp).  The same point
    // set on both sides, one right-hand side, a single (non-composite) profile without a Power wrapper, d <= 64; from n = 6144 (16384 for
    // the profiles that cost a reciprocal or less: below that the launch is latency and the triangle's imbalance, not arithmetic —
    // tools/fp64_sym_sweep.py: break-even at n ~ 6000, x1.1-1.25 at 8192, x1.3-1.5 at 16384, x1.5-1.75 from 32768) while the column-sum
    // slab n^2 / 8 bytes — held in workspace slot 4 for the life of the ctx — stays within 1 GiB (n <= 92681) when the choice is automatic,
    // 2 GiB (n <= 131072) when option "dense_sym" = 1 asks for it; R = 1 row per lane (64-row blocks) is what the kernel is written for.
    const bool cheap_profile = hk.tu_family == COVGRAM_CAUCHY || hk.tu_family == COVGRAM_IMQ || hk.tu_family == COVGRAM_DOT;
    // covgram_mvm_sym_partial (direct-difference form): rank r of P evaluates the row blocks r, r + P, ... only (sym_part_world > 0)
    // fp32 (dense_sym32_kernel, round 4): the same form for what the matrix-core path refuses — Exponential, gamma-exponential, MaternP(0),
    // clouds that fail its radius gate, dense_variant = 1 — from n = 24576 (32768 for the cheap profiles).
    const int sp_world = ctx->sym_part_world, sp_rank = ctx->sym_part_rank;
    const bool sym_forced = ctx->dense_sym == 1 || sp_world > 0;
    DenseSymShape symsh{};
    // fp32 break-even measured at n ~ 22000 (profiles/r04_sym32_sweep.txt: x0.85-0.9 at 16384, x1.2-1.5 at 32768, x1.5-1.85 from 65536)
    const int64_t sym_min_n = dtype == COVGRAM_F64 ? (cheap_profile ? 16384 : 6144) : (cheap_profile ? 32768 : 24576);   // fp64, d = 3 (tools/c1_sym_ab.py): n = 4096 31 against 29 us, 6144 42 against 48, 8192 62 against 77
    const bool dsym = !mfma && !wide && m > 0 && nrhs == 1 && (ctx->dense_sym != 0 || sp_world > 0) && X->dptr == Y->dptr && n == m &&
                      (sym_forced || n >= sym_min_n) && R == 1 &&
                      dense_sym_shape(hk, X, sp_world > 0 ? sp_world : 1, sym_forced ? 2 : 1, &symsh);
    if (sp_world > 0 && !dsym) { set_error("the symmetric direct-difference kernel does not apply to this kernel / point set"); return COVGRAM_EUNSUPPORTED; }
    const bool dsym32 = dsym && dtype == COVGRAM_F32;
    // fp64, wide points (round 4, dense_bcast.hpp): |x - y|^2 expanded around cached norms — one v_fmac_f64 per dimension and pair instead of
    // a subtraction and an fma, the column records in VGPRs (DPP broadcast) instead of the scalar stream — for the profiles that are smooth
    // in s at 0, one right-hand side, no Power wrapper, inside the expanded form's radius gate (as the gradient kernel's: grad_mvm.hpp).
    // Where the symmetric form applies too (gramian(k, x), n from 6144 / 16384, or covgram_mvm_sym_partial's cyclic row blocks) the two
    // combine: dense_bcast_sym_kernel evaluates the upper triangle once WITH the one-fmac distance (profiles/r04_fp64_dense_d_sweep.txt).
    const bool bc_family = hk.tu_family == COVGRAM_EQ || hk.tu_family == COVGRAM_RQ || hk.tu_family == COVGRAM_CAUCHY || hk.tu_family == COVGRAM_IMQ ||
                           (hk.tu_family == COVGRAM_MATERNP && hk.k.p >= 1);
    const bool dbc = !mfma && !wide && m > 0 && dtype == COVGRAM_F64 && nrhs == 1 && bc_family && hk.k.power == 1 && dense_bcast_ok(D) &&
                     ctx->dense_bcast != 0 && ctx->dense_variant != 1 &&
                     (ctx->dense_bcast == 1 || (D >= 16 && gate_radius2(X, Y) / (hk.k.lengthscale * hk.k.lengthscale) <= GRAD_EXPAND_GATE));
    ctx->last_dense_bcast = dbc ? 1 : 0;
    if (dbc) {
        const int64_t CB = DENSE_BCAST_CB;
        const int64_t mpad = ((m + CB - 1) / CB) * CB + CB;               // whole blocks + one prefetch-only block
        void* Pb;
        rc = ws_reserve(ctx, 0, (size_t)mpad * (D + 2) * ts, &Pb); if (rc) return rc;
        double* Exb = (double*)Pb + (size_t)mpad * D;
#define CG_BPL(DLV) hipLaunchKernelGGL((dense_bcast_pack_lanes_kernel<DLV>), dim3((unsigned)((mpad * DLV + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)Y->dptr, m, Y->d, \
                                       (const double*)a_dev, (double*)Pb, Exb, hk.kp.gamma, (const double*)Cn, mpad)
        if (D == 16) CG_BPL(16); else if (D == 32) CG_BPL(32); else if (D == 64) CG_BPL(64);
        else
        hipLaunchKernelGGL(dense_bcast_pack_kernel<double>, dim3((unsigned)((mpad + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)Y->dptr, m, Y->d,
                           (const double*)a_dev, (double*)Pb, Exb, D, hk.kp.gamma, (const double*)Cn, mpad);
#undef CG_BPL
        DenseArgs da;
        da.C = Cn; da.X = X->dptr; da.n = n; da.d = X->d; da.P = Pb; da.Ex = Exb; da.m = m; da.ldy = ldy_d; da.nrhs = 1;
        da.Dpad = D; da.NRpad = 1; da.rows_per_lane = 1; da.variant = 0; da.bcast = 1;
        da.alpha = alpha_eff; da.beta = beta; da.hk = &hk; da.stream = ctx->stream;
        ctx->last_dense_path = 1;
        int64_t jchunk; int jsplit;
        auto* tmb = timer_next(ctx);
        if (dsym) {
            // gramian(k, x): the upper triangle once on dense_bcast_sym_kernel — 64-row blocks, chunks of whole 64-column blocks, the column-sum
            // slab and the reduce kernel of dense_sym_kernel (covgram_mvm_sym_partial's cyclic blocks included)
            choose_split(ctx, rowblocks, m, 64, &jchunk, &jsplit);
            const int64_t blocks = (m + 63) / 64;
            const int64_t want = std::max<int64_t>(1, std::min<int64_t>(blocks, 2 * (int64_t)jsplit));
            const int64_t per = (blocks + want - 1) / want;
            da.jchunk = jchunk = per * 64;
            da.jsplit = jsplit = (int)((blocks + per - 1) / per);
            da.npad = npad; da.sym = 1;
            if (sp_world > 0) { da.sym_first = sp_rank; da.sym_stride = sp_world; }
            rc = ws_reserve(ctx, 1, (size_t)jsplit * npad * ts, &da.out); if (rc) return rc;
            rc = ws_reserve(ctx, 4, symsh.slab_bytes, &da.colslab); if (rc) return rc;
            if (tmb) (void)hipEventRecord(tmb->first, ctx->stream);
            rc = launch(da, dtype); if (rc) return rc;
            if (tmb) (void)hipEventRecord(tmb->second, ctx->stream);
            hipLaunchKernelGGL(dense_sym_reduce_kernel<double>, dim3((unsigned)rowblocks), dim3(1024), 0, ctx->stream, (const double*)da.out,
                               (const double*)da.colslab, npad, jsplit, (double*)y_dev, n, alpha_eff, beta, da.sym_first, da.sym_stride);
        } else {
            const int64_t rb256 = (n + 255) / 256, npadb = rb256 * 256;
            choose_split(ctx, rb256, m, 64, &jchunk, &jsplit, (int64_t)ctx->num_cus * 32);
            da.npad = npadb; da.jchunk = jchunk; da.jsplit = jsplit;
            if (jsplit == 1) da.out = y_dev;
            else { rc = ws_reserve(ctx, 1, (size_t)jsplit * npadb * ts, &da.out); if (rc) return rc; }
            if (tmb) (void)hipEventRecord(tmb->first, ctx->stream);
            rc = launch(da, dtype); if (rc) return rc;
            if (tmb) (void)hipEventRecord(tmb->second, ctx->stream);
            if (jsplit > 1)
                hipLaunchKernelGGL(dense_reduce_kernel<double>, dim3((unsigned)((n + 63) / 64), 1), dim3(256), 0, ctx->stream, (const double*)da.out, npadb, 1, jsplit,
                                   (double*)y_dev, n, ldy_d, 1, alpha_eff, beta);
        }
        ctx->last_jsplit = jsplit;
    }
    ctx->last_dense_sym = dsym ? 1 : 0;
    for (int c0 = 0; c0 < nrhs && !mfma && !dbc; c0 += 4) {
        const int nr = std::min(4, nrhs - c0);
        const int NRpad = (nr == 1) ? 1 : 4;
        const char* a_c = (const char*)a_dev + (size_t)c0 * lda_d * ts;
        char* y_c = (char*)y_dev + (size_t)c0 * ldy_d * ts;
        if (m == 0) {  // empty sum: y <- beta * y
            int64_t jc; int js;
            (void)jc; (void)js;
            if (dtype == COVGRAM_F32)
                hipLaunchKernelGGL(dense_reduce_kernel<float>, dim3((unsigned)((n + 63) / 64), nr), dim3(256), 0, ctx->stream,
                                   (const float*)nullptr, npad, NRpad, 0, (float*)y_c, n, ldy_d, nr, 0.0f, (float)beta);
            else
                hipLaunchKernelGGL(dense_reduce_kernel<double>, dim3((unsigned)((n + 63) / 64), nr), dim3(256), 0, ctx->stream,
                                   (const double*)nullptr, npad, NRpad, 0, (double*)y_c, n, ldy_d, nr, 0.0, beta);
            continue;
        }
        void* P;
        const int PADTO = dsym32 ? 8 : PKN;                 // dense_sym32_kernel reduces eight columns at a time
        int64_t mp = ((m + PADTO - 1) / PADTO) * PADTO;     // stream padded to whole column groups
        if (wide) {                                         // ... or to whole column blocks of the wide kernel
            const int64_t bc = 32 * PKN;
            mp = ((m + bc - 1) / bc) * bc;
        }
        rc = ws_reserve(ctx, 0, (size_t)mp * (D + NRpad) * ts, &P); if (rc) return rc;
        if (wide) {
            const int64_t tot = mp * (int64_t)(D + NRpad);
            if (dtype == COVGRAM_F32)
                hipLaunchKernelGGL(dense_wide_pack_kernel<float>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream,
                                   (const float*)Y->dptr, m, Y->d, D, (const float*)a_c, lda_d, nr, 0, (float*)P, NRpad, PKN, 32, (float)hk.kp.gamma, (const float*)Cn);
            else
                hipLaunchKernelGGL(dense_wide_pack_kernel<double>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream,
                                   (const double*)Y->dptr, m, Y->d, D, (const double*)a_c, lda_d, nr, 0, (double*)P, NRpad, PKN, 32, hk.kp.gamma, (const double*)Cn);
        } else if (dtype == COVGRAM_F32)
            hipLaunchKernelGGL(dense_pack_kernel<float>, dim3((unsigned)((mp + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const float*)Y->dptr, m, Y->d, (const float*)a_c, lda_d, nr, 0, (float*)P, D, NRpad, PKN, (float)hk.kp.gamma, (const float*)Cn, PADTO);
        else
            hipLaunchKernelGGL(dense_pack_kernel<double>, dim3((unsigned)((mp + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const double*)Y->dptr, m, Y->d, (const double*)a_c, lda_d, nr, 0, (double*)P, D, NRpad, PKN, hk.kp.gamma, (const double*)Cn, PADTO);
        int64_t jchunk; int jsplit;
        // chunks are multiples of 64 columns (whole packed pairs); small problems split finer than the 512-column
        // inner accumulation block so that they still expose thousands of waves
        choose_split(ctx, rowblocks, m, 64, &jchunk, &jsplit);
        ctx->last_jsplit = jsplit;
        DenseArgs da;
        da.C = Cn;
        da.X = X->dptr; da.n = n; da.d = X->d; da.P = P; da.m = m; da.npad = npad; da.ldy = ldy_d; da.nrhs = nr;
        da.Dpad = D; da.NRpad = NRpad; da.jchunk = jchunk; da.jsplit = jsplit; da.rows_per_lane = R;
        da.variant = (int)ctx->dense_variant; da.lds_pad = (int)ctx->lds_pad; da.alpha = alpha_eff; da.beta = beta; da.hk = &hk; da.stream = ctx->stream;
        if (dsym32) {
            // fp32: 64 R-row blocks (R rows per lane), chunks of whole 512-column inner blocks; ~64 k workgroups, half of them (left of
            // the diagonal) leave at once, so that the triangle's long and short rows even out over several rounds of resident waves
            const int64_t want = std::max<int64_t>(1, std::min<int64_t>((m + 511) / 512, (65536 + symsh.mine - 1) / symsh.mine));
            const int64_t per = ((m + 511) / 512 + want - 1) / want;
            da.jchunk = jchunk = per * 512;
            da.jsplit = jsplit = (int)((((m + 7) & ~(int64_t)7) + jchunk - 1) / jchunk);
            da.npad = symsh.npad;
            da.sym = 1;
            if (sp_world > 0) { da.sym_first = sp_rank; da.sym_stride = sp_world; }
            ctx->last_jsplit = jsplit;
            rc = ws_reserve(ctx, 1, (size_t)jsplit * symsh.npad * ts, &da.out); if (rc) return rc;
            rc = ws_reserve(ctx, 4, symsh.slab_bytes, &da.colslab); if (rc) return rc;
            auto* tms = timer_next(ctx);
            if (tms) (void)hipEventRecord(tms->first, ctx->stream);
            rc = launch(da, dtype); if (rc) return rc;
            if (tms) (void)hipEventRecord(tms->second, ctx->stream);
            hipLaunchKernelGGL(dense_sym32_reduce_kernel<float>, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, ctx->stream, (const float*)da.out,
                               (const float*)da.colslab, symsh.npad, jsplit, (float*)y_c, n, (float)alpha_eff, (float)beta, da.sym_first, da.sym_stride,
                               (int32_t)symsh.blk);
            continue;
        }
        if (dsym) {
            // chunks of whole 64-column blocks; about twice the workgroups of the all-entries launch, half of them (left of the
            // diagonal) leave at once
            int64_t blocks = (m + 63) / 64;
            int64_t want = std::max<int64_t>(1, std::min<int64_t>(blocks, 2 * (int64_t)jsplit));
            int64_t per = (blocks + want - 1) / want;
            da.jchunk = jchunk = per * 64;
            da.jsplit = jsplit = (int)((blocks + per - 1) / per);
            da.sym = 1;
            if (sp_world > 0) { da.sym_first = sp_rank; da.sym_stride = sp_world; }
            rc = ws_reserve(ctx, 1, (size_t)jsplit * npad * ts, &da.out); if (rc) return rc;
            rc = ws_reserve(ctx, 4, symsh.slab_bytes, &da.colslab); if (rc) return rc;
            auto* tms = timer_next(ctx);
            if (tms) (void)hipEventRecord(tms->first, ctx->stream);
            rc = launch(da, dtype); if (rc) return rc;
            if (tms) (void)hipEventRecord(tms->second, ctx->stream);
            hipLaunchKernelGGL(dense_sym_reduce_kernel<double>, dim3((unsigned)rowblocks), dim3(1024), 0, ctx->stream, (const double*)da.out,
                               (const double*)da.colslab, npad, jsplit, (double*)y_c, n, alpha_eff, beta, da.sym_first, da.sym_stride);
            continue;
        }
        if (jsplit == 1) da.out = y_c;
        else { rc = ws_reserve(ctx, 1, (size_t)jsplit * NRpad * npad * ts, &da.out); if (rc) return rc; }
        // the lane-per-row kernel can sum its own split-J slab (last-arriving workgroup of each 64-row block, fixed order: pack.hpp) — only
        // on request: its column splits are fine (64 at C1's size), and 64 tickets on one counter plus the uncached re-read cost more than
        // the launch they save (profiles/r04_inkernel_reduce_ab.txt: C1 27.1 -> 39.0 us, n = 16384 259 -> 262)
        const bool ikr = !wide && inkernel_reduce_on(ctx, false, n, jsplit);
        ctx->last_inkernel_reduce = ikr ? 1 : 0;
        if (ikr) { rc = tickets_reserve(ctx, (size_t)rowblocks, &da.tickets); if (rc) return rc; da.yfinal = y_c; }
        auto* tm = timer_next(ctx);
        if (tm) (void)hipEventRecord(tm->first, ctx->stream);
        rc = launch(da, dtype); if (rc) return rc;
        if (tm) (void)hipEventRecord(tm->second, ctx->stream);
        if (jsplit > 1 && !ikr) {
            if (dtype == COVGRAM_F32)
                hipLaunchKernelGGL(dense_reduce_kernel<float>, dim3((unsigned)((n + 63) / 64), nr), dim3(256), 0, ctx->stream,
                                   (const float*)da.out, npad, NRpad, jsplit, (float*)y_c, n, ldy_d, nr, (float)alpha_eff, (float)beta);
            else
                hipLaunchKernelGGL(dense_reduce_kernel<double>, dim3((unsigned)((n + 63) / 64), nr), dim3(256), 0, ctx->stream,
                                   (const double*)da.out, npad, NRpad, jsplit, (double*)y_c, n, ldy_d, nr, alpha_eff, beta);
        }
    }
    CG_CHECK_HIP(hipGetLastError());
    if (loc == COVGRAM_HOST) {
        CG_CHECK_HIP(hipMemcpy2DAsync(y, (size_t)ldy * ts, y_dev, (size_t)n * ts, (size_t)n * ts, nrhs, hipMemcpyDeviceToHost, ctx->stream));
        CG_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return COVGRAM_OK;
}

int covgram_matrix(covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, const covgram_points* Y, void* out,
                   int64_t ldo, int32_t loc) {
    int rc = check_pair(ctx, X, Y);
    if (rc) return rc;
    const int64_t n = X->n, m = Y->n;
    CG_REQUIRE(ldo >= n, COVGRAM_EINVAL, "ldo=%lld < n=%lld", (long long)ldo, (long long)n);
    CG_REQUIRE(out != nullptr || n * m == 0, COVGRAM_EINVAL, "out is NULL");
    const int dtype = X->dtype;
    const size_t ts = dtype_size(dtype);
    HostKernel hk;
    rc = make_host_kernel(k, dtype, true, &hk);   // gamma = 1/l, unfolded EQ
    if (rc) return rc;
    CG_DEVICE(ctx);
    if (n == 0 || m == 0) return COVGRAM_OK;
    void* o = out;
    int64_t ld = ldo;
    if (loc == COVGRAM_HOST) { rc = ws_reserve(ctx, 3, (size_t)n * m * ts, &o); if (rc) return rc; ld = n; }
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)((m + 15) / 16));
    if (X->d <= 64 && ctx->matrix_variant != 1) {                  // x_i in registers, 64-column strips (option matrix_variant = 1: the generic kernel)
        const dim3 g2((unsigned)((n + 255) / 256), (unsigned)((m + 63) / 64));
        const int dd = X->d;
        const bool expr = hk.tu_family >= COVGRAM_NFAMILY;
        const int fam_or_iso = expr ? (hk.tu_family == FAM_EXPR_ISO ? 1 : 0) : k->family;
        // rows per thread: 16-byte streaming stores (4 fp32 / 2 fp64 rows) when the points are short (registers), n and the leading dimension are
        // multiples of it and the output is 16-byte aligned; a single profile only (the composite interpreter keeps one row per thread)
#define CG_MAT(TT, DMV)                                                                                                                   \
        do {                                                                                                                              \
            constexpr int VRV = 16 / (int)sizeof(TT);                                                                                     \
            const bool vr = !expr && DMV <= 16 && n % VRV == 0 && ld % VRV == 0 && ((uintptr_t)o % 16) == 0;                                \
            const dim3 g3((unsigned)((n / VRV + 255) / 256), (unsigned)((m + 63) / 64));                                                  \
            if (expr) hipLaunchKernelGGL((matrix_reg_kernel<TT, DMV, true, ExprParams<TT>>), g2, dim3(256), 0, ctx->stream, (const TT*)X->dptr, n, \
                                         (const TT*)Y->dptr, m, dd, (TT*)o, ld, fam_or_iso, (TT)hk.kp.scale, make_params<FAM_EXPR_ISO, TT>(hk)); \
            else if (vr) hipLaunchKernelGGL((matrix_reg_kernel<TT, (DMV <= 16 ? DMV : 16), false, KParams<TT>, VRV>), g3, dim3(256), 0, ctx->stream, (const TT*)X->dptr, n,  \
                                    (const TT*)Y->dptr, m, dd, (TT*)o, ld, fam_or_iso, (TT)hk.kp.scale, cast_params<TT>(hk.kp));              \
            else hipLaunchKernelGGL((matrix_reg_kernel<TT, DMV, false, KParams<TT>>), g2, dim3(256), 0, ctx->stream, (const TT*)X->dptr, n,  \
                                    (const TT*)Y->dptr, m, dd, (TT*)o, ld, fam_or_iso, (TT)hk.kp.scale, cast_params<TT>(hk.kp));              \
        } while (0)
#define CG_MAT_D(TT) do { if (dd <= 4) CG_MAT(TT, 4); else if (dd <= 8) CG_MAT(TT, 8); else if (dd <= 16) CG_MAT(TT, 16); \
                          else if (dd <= 32) CG_MAT(TT, 32); else CG_MAT(TT, 64); } while (0)
        if (dtype == COVGRAM_F32) CG_MAT_D(float); else CG_MAT_D(double);
#undef CG_MAT_D
#undef CG_MAT
    } else if (hk.tu_family >= COVGRAM_NFAMILY) {
        const int iso = hk.tu_family == FAM_EXPR_ISO;
        if (dtype == COVGRAM_F32)
            hipLaunchKernelGGL(matrix_expr_kernel<float>, grid, dim3(256), 0, ctx->stream, (const float*)X->dptr, n, (const float*)Y->dptr,
                               m, X->d, (float*)o, ld, iso, (float)hk.kp.scale, make_params<FAM_EXPR_ISO, float>(hk));
        else
            hipLaunchKernelGGL(matrix_expr_kernel<double>, grid, dim3(256), 0, ctx->stream, (const double*)X->dptr, n, (const double*)Y->dptr,
                               m, X->d, (double*)o, ld, iso, hk.kp.scale, make_params<FAM_EXPR_ISO, double>(hk));
    } else if (dtype == COVGRAM_F32)
        hipLaunchKernelGGL(matrix_kernel<float>, grid, dim3(256), 0, ctx->stream, (const float*)X->dptr, n, (const float*)Y->dptr, m,
                           X->d, (float*)o, ld, k->family, (float)hk.kp.scale, cast_params<float>(hk.kp));
    else
        hipLaunchKernelGGL(matrix_kernel<double>, grid, dim3(256), 0, ctx->stream, (const double*)X->dptr, n, (const double*)Y->dptr, m,
                           X->d, (double*)o, ld, k->family, hk.kp.scale, cast_params<double>(hk.kp));
    CG_CHECK_HIP(hipGetLastError());
    if (loc == COVGRAM_HOST) {
        CG_CHECK_HIP(hipMemcpy2DAsync(out, (size_t)ldo * ts, o, (size_t)n * ts, (size_t)n * ts, m, hipMemcpyDeviceToHost, ctx->stream));
        CG_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return COVGRAM_OK;
}

// vg = 0: GradientKernel blocks (d × d); vg = 1: ValueGradientKernel blocks ((d+1) × (d+1))
// a: (m bd) x nrhs, y: (n bd) x nrhs, column-major (lda, ldy), bd = d + vg entries per block (src/gramian.jl:241-257: blockmul! takes
// vectors OF MATRICES too, and the block mul! of src/gradient.jl:86-92 broadcasts over their columns)
static int grad_mvm_impl(covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, const covgram_points* Y, const void* a, int64_t lda,
                         void* y, int64_t ldy, int32_t nrhs, double alpha, double beta, int32_t loc, int vg) {
    int rc = check_pair(ctx, X, Y);
    if (rc) return rc;
    CG_REQUIRE(nrhs >= 1, COVGRAM_EINVAL, "nrhs must be >= 1");
    CG_REQUIRE(lda >= Y->n * (int64_t)(Y->d + vg) && ldy >= X->n * (int64_t)(X->d + vg), COVGRAM_EINVAL,
               "lda / ldy smaller than the block vectors (%lld, %lld)", (long long)(Y->n * (int64_t)(Y->d + vg)), (long long)(X->n * (int64_t)(X->d + vg)));
    {
        SumTerm terms[COVGRAM_COMPOSITE_MAX_TERMS];
        int nt = 0;
        double constant = 0.0;
        // (a Sum the one-pass matrix-core kernels take is NOT split: every term shares the pair's distance, as in the reference's
        //  per-pair evaluation src/algebra.jl:27-47 — dense_mfma.hip: sum_fused_applies)
        if (!sum_fused_applies(ctx, k, X, Y, nrhs) && composite_sum_terms(ctx, k, loc, terms, &nt, &constant)) {   // derivatives are linear in the kernel: term by term as well
            HostKernel chk;
            rc = make_host_kernel(k, X->dtype, true, &chk);
            if (rc) return rc;
            for (int t = 0; t < nt; ++t) {
                rc = grad_mvm_impl(ctx, terms[t].ptr(), X, Y, a, lda, y, ldy, nrhs, alpha, t == 0 ? beta : 1.0, loc, vg);
                if (rc) return rc;
            }
            // a constant has zero derivatives: only the value-value entry of the value-gradient blocks sees it
            if (!vg) return COVGRAM_OK;
            return constant_term_mvm(ctx, X->dtype, a, Y->n, lda, X->d + 1, y, X->n, ldy, X->d + 1, nrhs, alpha * constant);
        }
    }
    const int64_t n = X->n, m = Y->n;
    const int d = X->d;
    const int bd = d + vg;                                  // entries per block of a and y
    CG_REQUIRE((a != nullptr || m == 0) && (y != nullptr || n == 0), COVGRAM_EINVAL, "a or y is NULL");
    const int dtype = X->dtype;
    const size_t ts = dtype_size(dtype);
    HostKernel hk;
    rc = make_host_kernel(k, dtype, true, &hk);
    if (rc) return rc;
    // lane-owned rows up to d = 64 (fp32) / 48 (fp64) (grad_mvm.hpp); wider rows take the two-kernel panel path (grad_wide.hpp)
    // (Round 1 sent composites with d >= 8 to the panel path too: the lane-per-row kernel, interpreting their jets once per pair
    // at 3 waves per SIMD, spilled — C4-shaped EQ*RQ 16.1 ms against 8.9.  With the registers the interpreter needs — two waves per
    // SIMD for fp64, grad_temp_regs — it is the faster one at every d it reaches: tools/compgrad_ab.py, panel / lane-per-row:
    // fp64 EQ*RQ d = 32 12.2 / 10.5 ms, d = 48 17.9 / 13.1, MaternP(2)*EQ d = 8 10.1 / 4.1; fp32 d = 32 3.9 / 1.7, d = 8 3.5 / 1.2.)
    bool heavy_factor = false;                                       // a composite with a Matern(nu) factor: the lane-per-row interpreter is built without it
    if (hk.tu_family >= COVGRAM_NFAMILY) {
        int nf = 0;
        for (int t = 0; t < hk.nterms; ++t) nf += hk.nfac[t];
        for (int f = 0; f < nf; ++f) heavy_factor = heavy_factor || hk.ffam[f] == COVGRAM_MATERN;
    }
    const bool wide = d > (dtype == COVGRAM_F64 ? 48 : 64) || ctx->grad_keep_r == 2 || heavy_factor;
    const int D = wide ? ((d + 31) / 32) * 32 : pad_dim(d);
    grad_launch_fn launch = grad_launcher(hk.tu_family);
    CG_DEVICE(ctx);
    if (n == 0) return COVGRAM_OK;

    const void* a_all = a;
    void* y_all = y;
    int64_t lda_d = lda, ldy_d = ldy;
    if (loc == COVGRAM_HOST) {
        void *sa, *sy;
        rc = ws_reserve(ctx, 2, (size_t)std::max<int64_t>(m, 1) * bd * nrhs * ts, &sa); if (rc) return rc;
        rc = ws_reserve(ctx, 3, (size_t)n * bd * nrhs * ts, &sy); if (rc) return rc;
        if (m > 0) CG_CHECK_HIP(hipMemcpy2DAsync(sa, (size_t)m * bd * ts, a, (size_t)lda * ts, (size_t)m * bd * ts, nrhs, hipMemcpyHostToDevice, ctx->stream));
        if (beta != 0.0) CG_CHECK_HIP(hipMemcpy2DAsync(sy, (size_t)n * bd * ts, y, (size_t)ldy * ts, (size_t)n * bd * ts, nrhs, hipMemcpyHostToDevice, ctx->stream));
        a_all = sa; y_all = sy; lda_d = m * (int64_t)bd; ldy_d = n * (int64_t)bd;
    }
    const bool iso = (k->trait == COVGRAM_ISOTROPIC);
    const void* Cn = iso ? Y->center : nullptr;
    const double alpha0 = alpha * hk.kp.scale;              // value row of the value-gradient blocks
    // isotropic: b = -2 gamma^2 (psi' a + 2 psi'' r'(r'.a));  dot product: b = k1 a + k2 y (x.a)
    const double alpha_eff = alpha * hk.kp.scale * (iso ? -2.0 * hk.kp.gamma2 : 1.0);
    const int64_t rowblocks = (n + GRAD_THREADS - 1) / GRAD_THREADS;
    const int64_t npad = rowblocks * GRAD_THREADS;
    // two right-hand sides per pass where the lane-per-row kernel is compiled for it (grad_mvm.hpp: r, s, phi', phi'' once per pair)
    const bool two_ok = !wide && m > 0 && hk.k.power == 1 && grad_two_rhs_ok(ts, D, hk.tu_family);
    // expanded form (grad_mvm.hpp): fp64 isotropic simple profiles whose pre-scaled clouds lie within the radius gate
    // fp32 (round 5): the same form inside GRAD_EXPAND_GATE_F32 — there the absolute error of s is a few fp32 roundings of R^2, what the fp32
    // matrix-core dense kernels carry inside their gate of the same size (dense_mfma.hip), measured at the gate in tests/test_gpu_grad_expand32.py
    const double expand_gate = dtype == COVGRAM_F64 ? GRAD_EXPAND_GATE : GRAD_EXPAND_GATE_F32;
    const bool expd_ok = !wide && m > 0 && iso && hk.tu_family < COVGRAM_NFAMILY && ctx->grad_keep_r != 1 && !(dtype == COVGRAM_F32 && hk.k.power != 1) &&
                          (ctx->grad_expand == 1 ||
                           // measured (tools/c4_expand_ab.py, profiles/r02_c4_expand_ab.txt): pays from d = 8 (C4 0.86x, d = 8 0.95x,
                           // d = 3 +4 %); MaternP(p >= 1) 0.89x since its exp(-r) is the library's own (it was +3 % with the
                           // 34-instruction one).  Never by default for the profiles that are singular at s = 0 (exponential,
                           // gamma-exponential, MaternP(0)): their diagonal blocks are NaN in the reference (inf * 0), which the
                           // exact zero of a direct difference reproduces and the rounded zero of the expanded form would not
                           // fp32 (tools/grad32_expand_ab.py, profiles/r05_grad32_expand_ab.txt, n = 16384): EQ d = 32 994 -> 828 us, d = 48 (n = 8192) 523 -> 391,
                           // d = 16 552 -> 457; RQ 0.85-0.94x; MaternP only from d = 32 (d = 8, 16: 1.12-1.19x — its jet's square root and the
                           // expanded form's clamp sit in front of a shorter dimension loop)
                           (ctx->grad_expand < 0 && D >= 8 && !(dtype == COVGRAM_F32 && hk.tu_family == COVGRAM_MATERNP && D < 32) &&
                            hk.tu_family != COVGRAM_MATERN && hk.tu_family != COVGRAM_EXP &&
                            hk.tu_family != COVGRAM_GAMMAEXP && !(hk.tu_family == COVGRAM_MATERNP && hk.k.p == 0) &&
                            hk.kp.gamma2 * gate_radius2(X, Y) <= expand_gate));
    // round 4: the expanded form with the column records in VGPRs (grad_bcast.hpp; option "grad_bcast": -1 auto, 0 never, 1 / 4 = with
    // that many waves per workgroup).  One right-hand side per pass: where it applies, matrix right-hand sides run column by column on it
    // (C4 shape: 2 x 1.36 ms against 3.30 ms for the scalar-stream kernel's two-column pass)
    int bcast_sel = 0;
    if (expd_ok && dtype == COVGRAM_F64 && hk.k.power == 1 && grad_bcast_ok(D) && hk.tu_family != COVGRAM_MATERN && ctx->grad_bcast != 0)
        bcast_sel = ctx->grad_bcast == 1 ? 1 : (ctx->grad_bcast == 4 ? 4 : GRAD_BCAST_AUTO(D));
    for (int c0 = 0; c0 < nrhs;) {
    const int nr = (two_ok && !bcast_sel && c0 + 1 < nrhs) ? 2 : 1;
    const void* a_dev = (const char*)a_all + (size_t)c0 * lda_d * ts;
    void* y_dev = (char*)y_all + (size_t)c0 * ldy_d * ts;
    const dim3 rgrid((unsigned)((n + 255) / 256), (unsigned)bd, (unsigned)nr);

    if (m == 0) {
        if (dtype == COVGRAM_F32)
            hipLaunchKernelGGL(grad_reduce_kernel<float>, rgrid, dim3(256), 0, ctx->stream, (const float*)nullptr,
                               npad, D, 0, (float*)y_dev, n, d, 0.0f, (float)beta, vg, 0.0f);
        else
            hipLaunchKernelGGL(grad_reduce_kernel<double>, rgrid, dim3(256), 0, ctx->stream, (const double*)nullptr,
                               npad, D, 0, (double*)y_dev, n, d, 0.0, beta, vg, 0.0);
    } else if (wide) {
        const int PKN = (dtype == COVGRAM_F32) ? 2 : 1;
        const int64_t BC = (int64_t)GJG * PKN;
        const int64_t mpad = ((m + BC - 1) / BC) * BC;
        const int64_t npad64 = ((n + 63) / 64) * 64;
        int64_t panel = (((int64_t)256 << 20) / (npad64 * 2 * (int64_t)ts)) / BC * BC;   // coefficient slab <= 256 MB
        panel = std::max<int64_t>(BC, std::min<int64_t>(panel, mpad));
        grad_wide_launch_fn wlaunch = grad_wide_launcher(hk.tu_family);
        // (row blocks x dimension chunks) waves may not fill the chip: split each panel's column blocks over z slices that
        // accumulate raw sums into their own output slabs, reduced in fixed order after the last panel
        const int64_t waves = (npad64 / 64) * (D / 32);
        int zs = (int)std::min<int64_t>(16, std::max<int64_t>(1, ((int64_t)ctx->num_cus * 16 + waves - 1) / waves));
        zs = (int)std::min<int64_t>(zs, std::max<int64_t>(1, panel / BC));
        void* zslab = nullptr;
        const int64_t total = n * (int64_t)bd;
        if (zs > 1) { rc = ws_reserve(ctx, 4, (size_t)zs * total * ts, &zslab); if (rc) return rc; }
        for (int64_t col0 = 0; col0 < mpad; col0 += panel) {
            const int64_t pc = std::min<int64_t>(panel, mpad - col0);
            void *P, *C;
            // stream + (value-gradient) the columns' value weights; coefficient slabs C1, C2 + the value row's block partials C0
            rc = ws_reserve(ctx, 0, (size_t)pc * (D * 2 + vg) * ts, &P); if (rc) return rc;
            rc = ws_reserve(ctx, 1, ((size_t)pc * 2 + (vg ? (size_t)(pc / BC) : 0)) * npad64 * ts, &C); if (rc) return rc;
            void* A0P = vg ? (void*)((char*)P + (size_t)pc * D * 2 * ts) : nullptr;
            void* C0 = vg ? (void*)((char*)C + (size_t)pc * 2 * npad64 * ts) : nullptr;
            const int64_t pe = pc * (int64_t)D;
            if (dtype == COVGRAM_F32)
                hipLaunchKernelGGL(grad_wide_pack_kernel<float>, dim3((unsigned)((pe + 255) / 256)), dim3(256), 0, ctx->stream,
                                   (const float*)Y->dptr, m, d, D, (const float*)a_dev, col0, pc, (float*)P, PKN, (float)hk.kp.gamma, vg, (float*)A0P, (const float*)Cn);
            else
                hipLaunchKernelGGL(grad_wide_pack_kernel<double>, dim3((unsigned)((pe + 255) / 256)), dim3(256), 0, ctx->stream,
                                   (const double*)Y->dptr, m, d, D, (const double*)a_dev, col0, pc, (double*)P, PKN, hk.kp.gamma, vg, (double*)A0P, (const double*)Cn);
            GradWideArgs wa;
            wa.Cn = Cn;
            wa.vg = vg; wa.A0P = A0P; wa.C0 = C0; wa.alpha0 = zs > 1 ? 1.0 : alpha0;
            wa.vg_c = (iso ? -1.0 : 1.0) / hk.kp.gamma; wa.vg_b = iso ? -2.0 * hk.kp.gamma : hk.kp.gamma;
            wa.X = X->dptr; wa.n = n; wa.d = d; wa.dpad = D; wa.P = P; wa.C1 = C; wa.C2 = (char*)C + (size_t)pc * npad64 * ts;
            wa.npad = npad64; wa.nblocks = pc / BC; wa.y = y_dev; wa.alpha = alpha_eff; wa.beta = beta; wa.accumulate = col0 > 0 ? 1 : 0;
            if (zs > 1) { wa.zs = zs; wa.zstride = total; wa.y = zslab; wa.alpha = 1.0; wa.beta = 0.0; }
            wa.hk = &hk; wa.stream = ctx->stream;
            auto* tm = timer_next(ctx);
            if (tm) (void)hipEventRecord(tm->first, ctx->stream);
            rc = wlaunch(wa, dtype); if (rc) return rc;
            if (tm) (void)hipEventRecord(tm->second, ctx->stream);
        }
        if (zs > 1) {
            const dim3 zg((unsigned)((n + 255) / 256), (unsigned)bd);
            if (dtype == COVGRAM_F32)
                hipLaunchKernelGGL(grad_wide_reduce_kernel<float>, zg, dim3(256), 0, ctx->stream, (const float*)zslab, zs, total, n, (float*)y_dev,
                                   (float)alpha_eff, (float)beta, vg, bd, (float)alpha0);
            else
                hipLaunchKernelGGL(grad_wide_reduce_kernel<double>, zg, dim3(256), 0, ctx->stream, (const double*)zslab, zs, total, n, (double*)y_dev,
                                   alpha_eff, beta, vg, bd, alpha0);
        }
    } else {
        void* P;
        // + 1 prefetch-only record; the value weights A0[0..m] of the value-gradient variant follow the stream
        // expanded form (grad_mvm.hpp): fp64 isotropic simple profiles whose pre-scaled clouds lie within the radius gate
        const bool expd = expd_ok;
        rc = ws_reserve(ctx, 0, (size_t)(m + 1) * ((1 + nr) * D + nr * vg + (expd ? 1 + nr : 0)) * ts, &P); if (rc) return rc;
        void* A0 = vg ? (void*)((char*)P + (size_t)(m + 1) * (1 + nr) * D * ts) : nullptr;
        void* Ex = expd ? (void*)((char*)P + (size_t)(m + 1) * ((1 + nr) * D + nr * vg) * ts) : nullptr;
        const int64_t pe = (m + 1) * (int64_t)D;
        if (dtype == COVGRAM_F32)
            hipLaunchKernelGGL(grad_pack_kernel<float>, dim3((unsigned)((pe + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const float*)Y->dptr, m, d, (const float*)a_dev, (float*)P, D, (float)hk.kp.gamma, vg, (float*)A0, (const float*)Cn, nr, lda_d);
        else
            hipLaunchKernelGGL(grad_pack_kernel<double>, dim3((unsigned)((pe + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const double*)Y->dptr, m, d, (const double*)a_dev, (double*)P, D, hk.kp.gamma, vg, (double*)A0, (const double*)Cn, nr, lda_d);
#define CG_PEL(TT, DLV) hipLaunchKernelGGL((grad_pack_extra_lanes_kernel<TT, DLV>), dim3((unsigned)(((m + 1) * DLV + 255) / 256)), dim3(256), 0, ctx->stream, (const TT*)Y->dptr, m, \
                                           (const TT*)a_dev, (TT)hk.kp.gamma, vg, (const TT*)Cn, (TT*)Ex, nr, lda_d)
#define CG_PE(TT) do { if (d == 8) CG_PEL(TT, 8); else if (d == 16) CG_PEL(TT, 16); else if (d == 32) CG_PEL(TT, 32); else if (d == 64) CG_PEL(TT, 64); \
                       else hipLaunchKernelGGL(grad_pack_extra_kernel<TT>, dim3((unsigned)((m + 256) / 256)), dim3(256), 0, ctx->stream, (const TT*)Y->dptr, m, d, \
                                               (const TT*)a_dev, (TT)hk.kp.gamma, vg, (const TT*)Cn, (TT*)Ex, nr, lda_d); } while (0)
        if (expd && dtype == COVGRAM_F64) CG_PE(double);
        else if (expd) CG_PE(float);
#undef CG_PE
#undef CG_PEL
        int64_t jchunk; int jsplit;
        // partial slabs cost jsplit * n * d * sizeof(T) bytes, but several rounds of workgroups balance the tail
        // (C4: 2.47 ms at CUs*8, 2.08 at CUs*32, 2.01 at CUs*64, 2.05 at CUs*96, 2.15 at CUs*128 — interleaved A/B, tools/c4_ab.py)
        // round 4: the expanded form with the column records in VGPRs (grad_bcast.hpp; option "grad_bcast": -1 auto, 0 never, 1 / 4 = with
        // that many waves per workgroup)
        const int bcast = bcast_sel;
        ctx->last_grad_bcast = bcast;
        const int bwaves = std::max(1, std::min(4, 512 / (4 * D + 4 * ((2 * D + 15) / 16) + 44)));   // grad_bcast_waves<D>()
        const int gthreads = bcast ? 64 * bcast : grad_block_threads((int)ts, D, hk.tu_family);          // 64 or 256 threads per workgroup (grad_mvm.hpp)
        // Column split: about 64 waves per CU over the launch (round 1's sweep), then (round 2, tools/c4_jsplit_sweep.py)
        //  * capped so that the partial slabs (split x n x d results, written here and read back by the reduction) stay inside the
        //    256 MB Infinity Cache — C4 at 64 splits writes 268 MB, d = 48 403 MB: 3.86 ms against 3.61 at 48 splits —
        //  * and, when that cap binds, snapped DOWN to a whole number of rounds of resident workgroups if one lies within reach (the
        //    waves of the EQ kernel all take the same time, so a ragged last round idles most of the chip: C4 at 64 row workgroups x
        //    {36, 42, 48, 54, 60} splits = {3.0, 3.5, 4.0, 4.5, 5.0} rounds: 1.744, 1.853, 1.748, 1.809, 1.770 ms).
        //  Old and new library alternating on one box: C4 1.79 -> 1.77 ms, value-gradient -2 %, d = 48 3.85 -> 3.65; the heavier
        //  profiles at the C4 shape lose 1.5 % (MaternP(2) 2.33 -> 2.37, RQ 3.60 -> 3.66): their slab is the same 268 MB but a
        //  smaller share of a longer kernel.
        const int64_t growwgs = (n + gthreads - 1) / gthreads;
        const int64_t gslots = (int64_t)ctx->num_cus * 4 * (bcast ? bwaves : grad_waves_per_simd((int)ts, D, hk.tu_family)) / (gthreads / 64);
        int64_t gsplit = std::max<int64_t>(1, ((int64_t)ctx->num_cus * 64 * 64 / gthreads + growwgs - 1) / growwgs);
        gsplit = std::min(gsplit, std::max<int64_t>(1, m / 64));   // >= 64 columns per workgroup: below that its prologue and slab rows dominate
                                                                    // (tools/c4_jsplit_sweep.py small: n = 4096, d = 8: 16-column chunks 0.109 ms, 64-column 0.059)
        const int64_t gcap = std::max<int64_t>(1, (int64_t)(256.0e6 / ((double)npad * (D + vg) * nr * ts)));
        if (ctx->jsplit <= 0 && ctx->target_wgs <= 0 && gsplit > gcap) {
            gsplit = gcap;
            for (int64_t js = gsplit; js >= std::max<int64_t>(1, gsplit / 2); --js) {
                const double rounds = (double)(js * growwgs) / (double)gslots;
                const double ragged = std::ceil(rounds) - rounds;
                if (rounds >= 1.0 && ragged <= 0.04) { gsplit = js; break; }
            }
        }
        choose_split(ctx, growwgs, m, 8, &jchunk, &jsplit, gsplit * growwgs);
        GradArgs ga;
        ga.C = Cn;
        ga.X = X->dptr; ga.n = n; ga.d = d; ga.P = P; ga.m = m; ga.npad = npad; ga.Dpad = D; ga.jchunk = jchunk; ga.jsplit = jsplit; ga.keep_r = (int)ctx->grad_keep_r;
        ga.alpha = alpha_eff; ga.beta = beta; ga.hk = &hk; ga.stream = ctx->stream;
        ga.vg = vg; ga.A0 = A0; ga.alpha0 = alpha0;
        ga.expd = expd ? 1 : 0; ga.Ex = Ex; ga.bcast = bcast;
        ga.nr = nr; ga.ldy = ldy_d;
        ctx->last_grad_expand = ga.expd;
        ga.vg_c = (iso ? -1.0 : 1.0) / hk.kp.gamma;
        ga.vg_b = iso ? -2.0 * hk.kp.gamma : hk.kp.gamma;
        if (jsplit == 1) ga.out = y_dev;
        else { rc = ws_reserve(ctx, 1, (size_t)jsplit * nr * (D + vg) * npad * ts, &ga.out); if (rc) return rc; }
        auto* tm = timer_next(ctx);
        if (tm) (void)hipEventRecord(tm->first, ctx->stream);
        rc = launch(ga, dtype); if (rc) return rc;
        if (tm) (void)hipEventRecord(tm->second, ctx->stream);
        if (jsplit > 1) {
            if (dtype == COVGRAM_F32)
                hipLaunchKernelGGL(grad_reduce_kernel<float>, rgrid, dim3(256), 0, ctx->stream, (const float*)ga.out,
                                   npad, D, jsplit, (float*)y_dev, n, d, (float)alpha_eff, (float)beta, vg, (float)alpha0, ldy_d);
            else
                hipLaunchKernelGGL(grad_reduce_kernel<double>, rgrid, dim3(256), 0, ctx->stream, (const double*)ga.out,
                                   npad, D, jsplit, (double*)y_dev, n, d, alpha_eff, beta, vg, alpha0, ldy_d);
        }
    }
    c0 += nr;
    }   // right-hand sides
    CG_CHECK_HIP(hipGetLastError());
    if (loc == COVGRAM_HOST) {
        CG_CHECK_HIP(hipMemcpy2DAsync(y, (size_t)ldy * ts, y_all, (size_t)n * bd * ts, (size_t)n * bd * ts, nrhs, hipMemcpyDeviceToHost, ctx->stream));
        CG_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return COVGRAM_OK;
}

int covgram_mvm_sym_supported(covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, int32_t world, int32_t* supported) {
    CG_REQUIRE(supported != nullptr, COVGRAM_EINVAL, "NULL argument");
    *supported = 0;
    CG_REQUIRE(world >= 1, COVGRAM_EINVAL, "world = %d", world);
    int rc = check_pair(ctx, X, X);
    if (rc) return rc;
    if (k == nullptr || k->family == COVGRAM_COMPOSITE) return COVGRAM_OK;
    HostKernel hk;
    rc = make_host_kernel(k, X->dtype, false, &hk);
    if (rc) return rc;
    if (X->n <= 0) return COVGRAM_OK;
    // fp64 (the reference's default element type), and fp32 where no matrix-core kernel takes the pair (profile, cloud): the direct-
    // difference symmetric kernels over cyclic row blocks, with the SAME shape / slab predicate covgram_mvm applies to the partial call
    // (it depends on world: a rank's column-sum slab is ceil(row blocks / world) x n scalars, capped at 2 GiB)
    if (X->dtype == COVGRAM_F64) { *supported = dense_sym_shape(hk, X, world, 2, nullptr) ? 1 : 0; return COVGRAM_OK; }
    if (mfma_eq_sym_eligible(ctx, hk, X, X, 1) || mfma_gen_sym_eligible(ctx, hk, X, X, 1)) { *supported = 1; return COVGRAM_OK; }
    if (mfma_eq_eligible(ctx, hk, X, X, 1) || mfma_gen_eligible(ctx, hk, X, X)) return COVGRAM_OK;   // the general matrix-core kernel is the better row shard
    *supported = dense_sym_shape(hk, X, world, 2, nullptr) ? 1 : 0;
    return COVGRAM_OK;
}

int covgram_mvm_sym_partial(covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, const void* a, void* y,
                            int32_t rank, int32_t world) {
    CG_REQUIRE(world >= 1 && rank >= 0 && rank < world, COVGRAM_EINVAL, "rank %d outside [0, %d)", rank, world);
    CG_REQUIRE(a != nullptr && y != nullptr, COVGRAM_EINVAL, "a or y is NULL");
    int32_t ok = 0;
    int rc = covgram_mvm_sym_supported(ctx, k, X, world, &ok);
    if (rc) return rc;
    if (!ok) { set_error("no symmetric kernel applies to this kernel / point set / world size"); return COVGRAM_EUNSUPPORTED; }
    HostKernel hk;
    rc = make_host_kernel(k, X->dtype, false, &hk);
    if (rc) return rc;
    const bool mc = X->dtype == COVGRAM_F32 && (mfma_eq_sym_eligible(ctx, hk, X, X, 1) || mfma_gen_sym_eligible(ctx, hk, X, X, 1));
    if (!mc) {
        // rank r takes the row blocks r, r + P, ... of the upper triangle on the direct-difference kernel of the data's precision;
        // the partials of all ranks add up to G a
        ctx->sym_part_rank = rank; ctx->sym_part_world = world;
        rc = covgram_mvm(ctx, k, X, X, a, X->n, y, X->n, 1, 1.0, 0.0, COVGRAM_DEVICE);
        ctx->sym_part_rank = 0; ctx->sym_part_world = 0;
        return rc;
    }
    CG_DEVICE(ctx);
    ctx->last_dense_path = 2; ctx->last_mfma_sym = 1;
    const bool fast = mfma_eq_sym_eligible(ctx, hk, X, X, 1);
    return mvm_eq_mfma_sym(ctx, hk, X, (const float*)a, (float*)y, 1.0, 0.0, rank, world, fast ? nullptr : k);
}

int covgram_grad_mvm(covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, const covgram_points* Y, const void* a, int64_t lda,
                     void* y, int64_t ldy, int32_t nrhs, double alpha, double beta, int32_t loc) {
    return grad_mvm_impl(ctx, k, X, Y, a, lda, y, ldy, nrhs, alpha, beta, loc, 0);
}

int covgram_valgrad_mvm(covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, const covgram_points* Y, const void* a, int64_t lda,
                        void* y, int64_t ldy, int32_t nrhs, double alpha, double beta, int32_t loc) {
    return grad_mvm_impl(ctx, k, X, Y, a, lda, y, ldy, nrhs, alpha, beta, loc, 1);
}

// debugging / test hook: the double-precision parameter block the device kernels receive.
// out[0..8] = gamma, gamma2, scale, param, c0, mp_c, mp_bound, mp_d1, mp_d2; then h0, h1, h2, ty (9 each).
int covgram_debug_kernel_params(const covgram_kernel* k, int32_t dtype, int32_t for_gradient, double* out45) {
    HostKernel hk;
    int rc = make_host_kernel(k, dtype, for_gradient != 0, &hk);
    if (rc) return rc;
    CG_REQUIRE(out45 != nullptr, COVGRAM_EINVAL, "out is NULL");
    const KParams<double>& p = hk.kp;
    double* o = out45;
    *o++ = p.gamma; *o++ = p.gamma2; *o++ = p.scale; *o++ = p.param; *o++ = p.c0; *o++ = p.mp_c; *o++ = p.mp_bound; *o++ = p.mp_d1; *o++ = p.mp_d2;
    for (int i = 0; i <= MAXP; ++i) *o++ = p.h0[i];
    for (int i = 0; i <= MAXP; ++i) *o++ = p.h1[i];
    for (int i = 0; i <= MAXP; ++i) *o++ = p.h2[i];
    for (int i = 0; i <= MAXP; ++i) *o++ = p.ty[i];
    return COVGRAM_OK;
}

}  // extern "C"
