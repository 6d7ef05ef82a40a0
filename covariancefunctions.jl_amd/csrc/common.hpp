// common.hpp — host-side plumbing shared by every translation unit of libcovgram.so.
// gfx950 only; no CPU compute path exists in this library (include/covgram.h).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <utility>
#include <vector>

#include "../../include/covgram.h"

namespace covgram {

// fused multiply-add in the operands' own precision (__builtin_fma on floats is the DOUBLE fma behind two conversions)
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }


void set_error(const char* fmt, ...);

#define CG_CHECK_HIP(expr)                                                                          \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            ::covgram::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                                 __LINE__);                                                         \
            return COVGRAM_EHIP;                                                                    \
        }                                                                                           \
    } while (0)

#define CG_REQUIRE(cond, code, ...)              \
    do {                                         \
        if (!(cond)) {                           \
            ::covgram::set_error(__VA_ARGS__);   \
            return (code);                       \
        }                                        \
    } while (0)

// Every entry point runs on its ctx's device and leaves the CALLER's current device as it found it (a process that holds
// tensors on several GPUs must not have its current device changed behind its back, also not from a destroy call that a
// garbage collector runs at an arbitrary time).
struct DeviceGuard {
    int prev = -1;
    bool good = true;
    explicit DeviceGuard(int dev) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        if (cur != dev) { good = hipSetDevice(dev) == hipSuccess; prev = cur; }
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
    bool ok() const { return good; }
};
#define CG_DEVICE(ctx)                                                                             \
    ::covgram::DeviceGuard _cg_dev((ctx)->device);                                                 \
    CG_REQUIRE(_cg_dev.ok(), COVGRAM_EHIP, "hipSetDevice(%d) failed", (ctx)->device)

// ---------------------------------------------------------------------------------------------
// Device-side kernel parameters (passed by value in the kernarg segment -> scalar loads).
// The host fills the double version from covgram_kernel; kernels receive the T version.
// ---------------------------------------------------------------------------------------------
constexpr int MAXP = COVGRAM_MATERNP_MAX_P;

template <typename T>
struct KParams {
    T gamma;        // coordinate pre-scale applied to x and y before differences / dots
    T gamma2;       // gamma^2 (chain-rule factor for the gradient kernel)
    T scale;        // Constant multiplier
    T param;        // RQ alpha | gammaExp gamma/2 | IMQ c^2 (pre-scaled)
    T c0;           // EQ: exponent factor for exp2 (unfolded path); RQ: 1/(2 alpha)
    T mp_c;         // MaternP: 2p+1
    T mp_bound;     // MaternP: eps(T)^(1/p) Taylor guard (src/stationary.jl:135-136)
    T mp_d1, mp_d2; // MaternP: first / second derivative at zero
    T h0[MAXP + 1]; // MaternP: normalised polynomial of H_p      (value)
    T h1[MAXP + 1]; //          H_{p-1}                           (phi')
    T h2[MAXP + 1]; //          H_{p-2}                           (phi'')
    T ty[MAXP + 1]; // MaternP: Taylor coefficients d_i / i!  (ty[0] = 1)
    int32_t p;      // MaternP order
    int32_t power;  // Power exponent
};

// Composite kernels (covgram_kernel_composite: a sum of products of simple profiles that share one input trait,
// src/algebra.jl:5-63, src/properties.jl:47-63) run through the same kernels under two extra template "families"
// whose parameter block carries one KParams per factor; the profile is interpreted per pair with wave-uniform
// (scalar) control flow.
constexpr int FAM_EXPR_ISO = COVGRAM_NFAMILY;        // composite of isotropic factors:   phi(s), s = |x-y|^2
constexpr int FAM_EXPR_DOT = COVGRAM_NFAMILY + 1;    // composite of dot-product factors: phi(s), s = x.y
constexpr int NUM_TU_FAMILIES = COVGRAM_NFAMILY + 2; // translation units per kernel kind (Makefile FAMS)
constexpr int EXPR_MAXT = COVGRAM_COMPOSITE_MAX_TERMS;
constexpr int EXPR_MAXF = COVGRAM_COMPOSITE_MAX_FACTORS;

template <typename T>
struct ExprParams {
    T gamma;                     // coordinate pre-scale of the rows: always 1 (every factor scales s by its own gamma2)
    int32_t power;               // always 1 (Power is applied per factor)
    int32_t nterms;
    int32_t nfac[EXPR_MAXT];     // profile factors per term, stored consecutively in f[] (0: the term is its coefficient)
    T coef[EXPR_MAXT];           // product of the term's scales and Constant factors (folded by the host)
    int32_t fam[EXPR_MAXF];      // covgram_family of each profile factor
    KParams<T> f[EXPR_MAXF];     // per factor: gamma2 = 1/l^2, power and the profile constants (scale is in coef)
};

// A Sum of two or three single-profile isotropic terms (src/algebra.jl:5-14; the reference evaluates every term on the SAME pair, :27-47) on
// the fp32 matrix-core kernels in ONE pass (round 5): the MFMA yields s' = g^2 |x - y|^2 once, term t sees s' ratio[t] (ratio[0] = 1: g carries
// the first term's whole argument scale, lengthscale and folded constants), and the profiles are evaluated term after term on the tile's
// register pairs (dense_mfma.hpp: mfma_sum_block).  A matrix-core-only pseudo-family: no lane-per-row / gradient instance exists.
constexpr int FAM_SUM_ISO = COVGRAM_NFAMILY + 2;
constexpr int SUM_MAXT = 3;
template <typename T>
struct SumParams {
    T gamma;                     // coordinate pre-scale of rows and columns: the first term's argument is the MFMA result itself
    int32_t power;               // always 1
    int32_t nterms;              // 2 or 3
    int32_t fam[SUM_MAXT];       // covgram_family of each term: EQ, RQ, Cauchy, IMQ or MaternP
    int32_t p[SUM_MAXT];         // MaternP order (1 .. 3)
    T ratio[SUM_MAXT];           // the term's argument is s' ratio[t]
    // the term's constants, its coefficient folded in: EQ {coef}: coef exp2(-arg); MaternP {coef h_0 .. coef h_3} (tables rescaled to
    // sqrt(arg) = r log2 e): q(sqrt arg) exp2(-sqrt arg); RQ {coef, -alpha}: coef exp2(-alpha log2(1 + arg)); Cauchy {coef}: coef / (1 + arg);
    // IMQ {coef, c^2}: coef rsq(arg + c^2)
    T c[SUM_MAXT][4];
};

template <int FAM, typename T> struct ParamsOf { using type = KParams<T>; };
template <typename T> struct ParamsOf<FAM_EXPR_ISO, T> { using type = ExprParams<T>; };
template <typename T> struct ParamsOf<FAM_EXPR_DOT, T> { using type = ExprParams<T>; };
template <typename T> struct ParamsOf<FAM_SUM_ISO, T> { using type = SumParams<T>; };
template <int FAM> constexpr bool fam_is_expr = (FAM == FAM_EXPR_ISO || FAM == FAM_EXPR_DOT);
template <int FAM> constexpr bool fam_is_iso = (FAM != COVGRAM_DOT && FAM != COVGRAM_EXPDOT && FAM != COVGRAM_ASINDOT && FAM != FAM_EXPR_DOT);

struct HostKernel {
    covgram_kernel k;     // simple kernel, or the head of a composite
    KParams<double> kp;   // un-typed master copy (composite: gamma = 1, scale = head scale)
    bool eq_folded;       // dense path folds -log2(e)/2 into gamma for EQ
    int tu_family;        // launcher index: k.family, or FAM_EXPR_ISO / FAM_EXPR_DOT
    int nterms;           // composite only
    int nfac[EXPR_MAXT];
    double coef[EXPR_MAXT];
    int ffam[EXPR_MAXF];
    KParams<double> fkp[EXPR_MAXF];
};

// Validates `k` and fills the parameter block.  `for_gradient` keeps gamma = 1/l (no log2e fold).
int make_host_kernel(const covgram_kernel* k, int dtype, bool for_gradient, HostKernel* out);

template <typename T>
inline KParams<T> cast_params(const KParams<double>& s) {
    KParams<T> d;
    d.gamma = (T)s.gamma; d.gamma2 = (T)s.gamma2; d.scale = (T)s.scale; d.param = (T)s.param;
    d.c0 = (T)s.c0; d.mp_c = (T)s.mp_c; d.mp_bound = (T)s.mp_bound; d.mp_d1 = (T)s.mp_d1; d.mp_d2 = (T)s.mp_d2;
    for (int i = 0; i <= MAXP; ++i) { d.h0[i] = (T)s.h0[i]; d.h1[i] = (T)s.h1[i]; d.h2[i] = (T)s.h2[i]; d.ty[i] = (T)s.ty[i]; }
    d.p = s.p; d.power = s.power;
    return d;
}

// Is the composite `hk` a Sum the one-pass kernels take?  2-3 terms, each ONE profile (times constants) of EQ / RQ / Cauchy / IMQ /
// MaternP(1..3) without a Power wrapper.
inline bool sum_fusable(const HostKernel& hk) {
    if (hk.tu_family != FAM_EXPR_ISO || hk.nterms < 2 || hk.nterms > SUM_MAXT) return false;
    for (int t = 0; t < hk.nterms; ++t) {
        if (hk.nfac[t] != 1 || hk.fkp[t].power != 1) return false;
        const int f = hk.ffam[t];
        const bool ok = f == COVGRAM_EQ || f == COVGRAM_RQ || f == COVGRAM_CAUCHY || f == COVGRAM_IMQ ||
                        (f == COVGRAM_MATERNP && hk.fkp[t].p >= 1 && hk.fkp[t].p <= 3);
        if (!ok) return false;
    }
    return true;
}
// the one-pass parameter block of such a Sum (hk.fkp[t]: the factor's UNFOLDED block, gamma = 1 / l_t, natural tables)
inline SumParams<double> make_sum_params(const HostKernel& hk) {
    SumParams<double> sp;
    memset(&sp, 0, sizeof(sp));
    sp.power = 1; sp.nterms = hk.nterms;
    const double LOG2E = 1.4426950408889634074;
    double a0 = 1.0;
    for (int t = 0; t < hk.nterms && t < SUM_MAXT; ++t) {
        const KParams<double>& q = hk.fkp[t];
        const double coef = hk.coef[t];
        double at = q.gamma2;                                   // the term's argument = at |x - y|^2
        sp.fam[t] = hk.ffam[t]; sp.p[t] = q.p;
        switch (hk.ffam[t]) {
            case COVGRAM_EQ: at *= 0.5 * LOG2E; sp.c[t][0] = coef; break;                      // exp(-s/2) = exp2(-arg)
            case COVGRAM_MATERNP: {                                                              // sqrt(arg) = r log2 e, r = sqrt((2p+1) s)
                at *= q.mp_c * LOG2E * LOG2E;
                double li = 1.0;
                for (int i = 0; i < 4; ++i) { sp.c[t][i] = i <= q.p ? coef * q.h0[i] * li : 0.0; li /= LOG2E; }
                break;
            }
            case COVGRAM_RQ: at *= q.c0; sp.c[t][0] = coef; sp.c[t][1] = -q.param; break;     // u = 1 + s / (2 alpha)
            case COVGRAM_IMQ: sp.c[t][0] = coef; sp.c[t][1] = q.param; break;                  // param holds c^2
            default: sp.c[t][0] = coef; break;                                                   // Cauchy
        }
        if (t == 0) a0 = at;
        sp.ratio[t] = at / a0;
    }
    sp.gamma = sqrt(a0);
    return sp;
}

template <int FAM, typename T>
inline typename ParamsOf<FAM, T>::type make_params(const HostKernel& hk) {
    if constexpr (FAM == FAM_SUM_ISO) {
        const SumParams<double> s = make_sum_params(hk);
        SumParams<T> d;
        memset(&d, 0, sizeof(d));
        d.gamma = (T)s.gamma; d.power = 1; d.nterms = s.nterms;
        for (int t = 0; t < SUM_MAXT; ++t) {
            d.fam[t] = s.fam[t]; d.p[t] = s.p[t]; d.ratio[t] = (T)s.ratio[t];
            for (int i = 0; i < 4; ++i) d.c[t][i] = (T)s.c[t][i];
        }
        return d;
    } else if constexpr (fam_is_expr<FAM>) {
        ExprParams<T> e;
        memset(&e, 0, sizeof(e));
        e.gamma = (T)1; e.power = 1; e.nterms = hk.nterms;
        for (int t = 0; t < EXPR_MAXT; ++t) { e.nfac[t] = hk.nfac[t]; e.coef[t] = (T)hk.coef[t]; }
        for (int f = 0; f < EXPR_MAXF; ++f) { e.fam[f] = hk.ffam[f]; e.f[f] = cast_params<T>(hk.fkp[f]); }
        return e;
    } else {
        return cast_params<T>(hk.kp);
    }
}

// ---------------------------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------------------------
struct Workspace {
    void* ptr = nullptr;
    size_t bytes = 0;
};

}  // namespace covgram

struct covgram_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    covgram::Workspace ws[5];  // 0: packed tile stream, 1: split-J partials, 2/3: host staging (device copies of a / y), 4: wide-gradient slices
    // options
    int64_t dense_variant = 0;   // 0 auto (fp32 EQ on the matrix cores when the norm bound allows; Dot() as X (Y' a)), 1 direct differences / entry by entry, 2 MFMA whenever the shape allows
    int64_t rows_per_lane = 0;   // 0 = auto
    int64_t jsplit = 0;          // 0 = auto
    int64_t target_wgs = 0;      // 0 = auto (CUs * 8)
    int64_t grad_expand = -1;    // fp64 isotropic gradient Gramians in the expanded form (4 fma per dimension and pair): -1 inside the radius gate, 0 never, 1 always
    int64_t grad_keep_r = -1;    // -1 auto
    int64_t grad_bcast = -1;     // fp64 expanded-form gradient MVM with the column records in VGPRs (v_fmac_f64_dpp row_newbcast): -1 auto, 0 never, 1 / 4 = always with that many waves per workgroup
    int64_t last_grad_bcast = 0;
    int64_t lds_pad = 0;         // occupancy experiments: dynamic LDS bytes per dense workgroup
    int64_t mfma_sym_rt = -1;    // symmetric matrix-core EQ kernel: row tiles per wave (-1 / 2: two, 4-wave workgroups, dense_mfma_sym2.hpp; 1: one, 8-wave workgroups)
    int64_t last_mfma_sym_rt = 0;
    int64_t mfma_sym_st = 0;     // symmetric matrix-core kernels with 4-wave workgroups: column tiles per stage (0 = auto: 8 where three workgroups still fit a CU's LDS, 4 = always four)
    int64_t lowrank_wgs = 0;      // low rank: row slabs of the first pass per CU (0 = auto: 2, matrix right-hand sides 4)
    int64_t lowrank_reverse = -1; // low rank U (V' a), one right-hand side: the second pass walks U back to front (-1: when U is V and fits the Infinity Cache, 0 never, 1 always)
    int64_t sum_fused = -1;      // fp32 Sum of 2-3 single-profile isotropic terms: one pass of the matrix-core kernels over the shared distance (-1 / 1 where they apply, 0: one MVM per term)
    int64_t last_sum_fused = 0;
    int64_t composite_termwise = 1;   // Sum of single-profile terms: one MVM per term on its own path (0: the composite interpreter)
    int64_t dense_sym = -1;      // fp64 direct-difference path on gramian(k, x): upper triangle once (-1 auto: n >= 6144 / 16384, 0 never, 1 always)
    int64_t last_dense_sym = 0;
    int64_t dense_bcast = -1;    // fp64 dense MVM of wide points on dense_bcast_kernel (expanded distance, v_fmac_f64_dpp): -1 = from padded d = 16 inside the radius gate, 0 never, 1 whenever compiled (d >= 8)
    int64_t last_dense_bcast = 0;
    // split-J partials summed by the last-arriving workgroup of each row block instead of a reduce launch (pack.hpp: last_arrival).  ONE rule
    // (inkernel_reduce_on below, used by both call sites; include/covgram.h says the same): 1 = wherever the column split is > 1 (lane-per-row
    // and matrix-core EQ kernels), 0 = never, -1 = the matrix-core EQ kernel up to n = 4096 only (where the whole MVM is launch latency)
    int64_t inkernel_reduce = -1;
    int64_t last_inkernel_reduce = 0;
    unsigned* tickets = nullptr;  // one arrival counter per row block, zero between launches (pack.hpp: last_arrival)
    size_t tickets_cap = 0;
    int32_t sym_part_rank = 0, sym_part_world = 0;   // set by covgram_mvm_sym_partial around its fp64 call of covgram_mvm: world > 0 = partial form
    int64_t mfma_sym = -1;       // matrix-core EQ path on gramian(k, x): evaluate the upper triangle once (-1 auto, 0 never, 1 always)
    int64_t mfma_lds = -1;       // matrix-core EQ path: 4 waves share the column tiles through LDS (-1 auto, 0 never, 1 always)
    int64_t mfma_stamp = 0;      // 1: the general matrix-core EQ kernel runs its clock-stamping diagnostic build (info key "last_clock_khz")
    void* stamp_buf = nullptr;   // [workgroup][4]: s_memtime / s_memrealtime before and after the column loop
    size_t stamp_cap = 0, stamp_count = 0;
    int64_t matrix_variant = 0;     // Matrix(G): 0 = rows in registers + 64-column strips (d <= 64), 1 = the generic entry-by-entry kernel
    int64_t mfma_mrhs = -1;         // matrix right-hand sides on the fp32 matrix cores (dense_mfma_mrhs_kernel): -1 from 5 (9: cheap profiles, d <= 3) columns, 0 never, 1 from 2
    int64_t toeplitz_real_spectrum = 1; // handles of symmetric Toeplitz matrices created while this is 1 keep the row kernel's spectrum copy as reals
    int64_t toeplitz_colfft = 16; // column FFT of the Toeplitz fast path: 16 = radix-16 register butterflies (colfft16_kernel), 4 = the radix-4 LDS kernel
    int64_t mfma_fuse_w = -1;      // general matrix-core EQ kernel: the column weights a_j exp2(f_j) formed in the kernel (-1 / 1) or by a pack launch in front of it (0)
    int64_t kron_fill = 1;         // Kronecker mode kernel: workgroups per CU its column tiling aims at (1, 2)
    int64_t mfma_gate_pct = 100;   // both radius gates of the fp32 matrix-core kernels, in percent of MFMA_GATE / MFMA_F16_GATE (1..100): the worst-case ROW-wise error scales with the gate (profiles/r05_gate_scan.txt)
    int64_t mfma_f16 = -1;         // general matrix-core EQ kernel: the fp16 two-way split (half the MFMAs per tile): -1 / 1 = within MFMA_F16_GATE, 0 = never, 2 = within MFMA_GATE (measurements only)
    int64_t last_mfma_f16 = 0;
    int64_t last_mfma_instance = 0;   // template arguments of the last matrix-core EQ kernel launched: K2 1e5 + RT 1e4 + WPB 1e3 + LDS 100 + STAMP 10 + FMT (general), -(K2 10 + FMT) (symmetric), 0 other
    int64_t toeplitz_persist = -1; // fused radix-16 row kernel: persistent workgroups (one per CU; a value > 1 = that many) that prefetch the next row pair into registers: -1 = fp64 only (measured), 0 = one pair per workgroup, 1 = always
    int64_t toeplitz_fused = 1;  // 1: row FFT + spectral step + inverse row FFT as one kernel when M' = 4^L <= 4096; 0: rocFFT batches
    void* comm = nullptr;        // ncclComm_t of covgram_comm_create (comm.hip): the MVM's one collective runs on `stream`
    int32_t comm_rank = 0, comm_world = 0;
    int num_cus = 256;
    void* blas = nullptr;        // rocblas_handle of the compute-bound Kronecker mode products (kron.hip), created on first use
    int live_handles = 0;
    int64_t last_dense_path = 0; // 1 lane-per-row, 2 matrix cores, 3 wide rows, 4 factored dot product X (Y' a)
    int64_t last_kron_path = 0;  // kernels of the last Kronecker MVM, bits: 1 fused last-two-modes pass, 2 single-mode kernel, 4 last-mode kernel, 8 rocBLAS GEMM, 16 two small trailing factors multiplied out
    int64_t last_jsplit = 0;     // column split of the last lane-per-row dense launch (tools)
    int32_t* sym_map = nullptr;  // symmetric kernel: device list of its (local panel, chunk) workgroups, keyed by sym_key
    size_t sym_map_cap = 0, sym_map_len = 0;
    int64_t sym_key[4] = {-1, -1, -1, -1};
    int64_t last_mfma_sym = 0;   // the last dense MVM ran the symmetric (upper-triangle) matrix-core kernel
    int64_t last_grad_expand = 0; // the last lane-per-row gradient MVM ran the expanded form
    int64_t last_mfma_lds = 0;   // the last matrix-core EQ MVM shared its column tiles through LDS
    // optional HIP-event bracketing of the dominant kernel of each MVM (bench.py's live roofline measurement)
    int64_t time_kernels = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timers;
    size_t timers_used = 0;
};

struct covgram_points {
    covgram_ctx* ctx = nullptr;
    void* dptr = nullptr;  // device pointer, point-major n×d
    int64_t n = 0;
    int32_t d = 0;
    int32_t dtype = 0;
    bool owns = false;
    double max_norm2 = 0;  // max_i |x_i|^2 (upper bound; slices inherit the parent's), computed once at creation
    // Common centre of isotropic kernels: a sample mean of the ROOT point set (slices inherit it), read by the kernels from
    // device memory (center) and by the host gates from the copy taken at creation (center_host).  Isotropic kernels
    // evaluate ((x - c) - (y - c)) gamma with c = the COLUMN side's centre, so the pre-scaled coordinates round relative to
    // the cloud's extent and results do not depend on where the cloud sits (translation invariance of r = |x - y|).
    const void* center = nullptr;
    void* center_buf = nullptr;       // the device buffer behind `center` (d scalars), freed by the handle that owns it
    bool owns_center = false;
    std::vector<double> center_host;
    double max_cnorm2 = 0; // max_i |x_i - c|^2 (upper bound), same reduction as max_norm2
    // matrix-core EQ path: the B fragments of this point set as the COLUMN side depend only on (points, gamma), not on the
    // weights, so they are packed once and reused by every later MVM (the points are resident and immutable while the handle
    // lives); only the 4-byte-per-column weights are rebuilt per MVM.  FRAG_SLOTS slots keyed on (gamma, K2): a hyper-parameter
    // loop or the terms of EQ(l1) + EQ(l2) alternate lengthscales on one handle, and a changed gamma re-packs the least recently
    // used slot IN PLACE, ordered on the ctx stream behind the MVMs that read it — no hipFree, no stream synchronisation and no
    // allocation on the MVM path once the slots exist (a slot is allocated the first time it is needed and lives as long as the handle)
    static constexpr int FRAG_SLOTS = 4;
    struct FragSlot { void* ptr = nullptr; size_t bytes = 0; float g = 0; int k2 = 0; int fmt = 0; uint64_t used = 0; bool pinned = false; };   // pinned: handed out during a stream capture — a graph has its address baked in, so it is never re-packed or freed while the handle lives
    mutable FragSlot frag[FRAG_SLOTS];
    mutable uint64_t frag_clock = 0;
};

namespace covgram {

inline bool inkernel_reduce_on(const covgram_ctx* ctx, bool mfma_eq_kernel, int64_t n, int64_t jsplit) {
    if (jsplit <= 1 || ctx->inkernel_reduce == 0) return false;
    return ctx->inkernel_reduce == 1 || (mfma_eq_kernel && n <= 4096);
}
int ws_reserve(covgram_ctx* ctx, int slot, size_t bytes, void** out);
// >= count zeroed arrival counters (grown stream-ordered; they return to zero by themselves after every launch that uses them)
int tickets_reserve(covgram_ctx* ctx, size_t count, unsigned** out);
// returns an event pair to record around the dominant kernel, or nullptr when timing is off / the pool is full
std::pair<hipEvent_t, hipEvent_t>* timer_next(covgram_ctx* ctx);
inline size_t dtype_size(int dtype) { return dtype == COVGRAM_F64 ? 8 : 4; }

// dispatch tables implemented in the per-family translation units -------------------------------
struct DenseArgs {
    const void* X; int64_t n; int32_t d;   // rows (raw user layout, stride d)
    const void* P; int64_t m;              // packed [m][D + NRHS] stream
    void* out;                             // partials [jsplit][NRHS][npad] or final y when jsplit == 1
    int64_t npad; int64_t ldy;
    int32_t nrhs; int32_t Dpad; int32_t NRpad;
    int64_t jchunk; int32_t jsplit; int32_t rows_per_lane; int32_t variant;
    int32_t lds_pad = 0;                   // dynamic LDS bytes requested only to cap waves per CU (occupancy experiments)
    const void* C = nullptr;               // common centre (d scalars on the device) subtracted from both sides by isotropic kernels
    int32_t bcast = 0; const void* Ex = nullptr;   // fp64 wide points: dense_bcast_kernel (expanded distance, records in VGPRs); P = [mpad][D] points, Ex = [mpad][2] (|y'|^2, a)
    unsigned* tickets = nullptr; void* yfinal = nullptr;   // jsplit > 1: the last workgroup of a row block sums the slab into yfinal (pack.hpp: last_arrival)
    int32_t sym = 0; void* colslab = nullptr;   // fp64 gramian(k, x): dense_sym_kernel (upper triangle once) + its [row blocks of this launch][npad] column-sum slab
    int32_t sym_first = 0, sym_stride = 1;      // ... over the 64-row blocks first, first + stride, ... (covgram_mvm_sym_partial: rank, world)
    double alpha, beta;
    const HostKernel* hk;
    hipStream_t stream;
};
typedef int (*dense_launch_fn)(const DenseArgs&, int dtype);
dense_launch_fn dense_launcher(int family);
dense_launch_fn dense_wide_launcher(int family);

struct GradArgs {
    const void* X; int64_t n; int32_t d;
    const void* P; int64_t m;              // packed [m][2*D] stream: y_j (scaled) then a_j
    void* out; int64_t npad;               // partials [jsplit][npad][D] or final
    int32_t Dpad; int64_t jchunk; int32_t jsplit;
    int32_t keep_r;                        // -1 auto, 0 recompute r in sweep 2, 1 keep r in VGPRs
    const void* C = nullptr;               // common centre of isotropic kernels (dense_mvm.hpp)
    int32_t vg = 0;                        // 1: ValueGradientKernel blocks of d+1 (out slab rows D+1)
    const void* A0 = nullptr;              // vg: value weights of the columns, m+1 entries
    double alpha0 = 0, vg_c = 0, vg_b = 0; // vg: scale of the value row, c2 and b0 coupling coefficients
    int32_t nr = 1;                        // right-hand sides of this launch (1 or 2: grad_two_rhs_ok): records of (1 + nr) D scalars
    int64_t ldy = 0;                       // elements between the outputs of two right-hand sides
    int32_t expd = 0;                      // 1: expanded form (fp64 isotropic): Ex holds (|y'_j|^2, y'_j . a_j^(0..nr-1)) per column, m + 1 entries
    const void* Ex = nullptr;
    int32_t bcast = 0;                     // expanded form with the records in VGPRs (grad_bcast.hpp): waves per workgroup (1 / 4), 0 = the scalar-stream kernel
    double alpha, beta;
    const HostKernel* hk;
    hipStream_t stream;
};
typedef int (*grad_launch_fn)(const GradArgs&, int dtype);
grad_launch_fn grad_launcher(int family);

// fp32 EQ on the matrix cores (dense_mfma.hip)
// P = max(max|x~|, max|y~|)^2 (both radii about the column side's centre) up to which the expanded exponent is used.  Its
// absolute error is a few fp32 roundings of O(P): measured contribution to the MVM's 2-norm relative error ~4e-9 P (C2:
// P = 40, +0.7e-7; P = 125: 5e-7, tests), against the 1e-5 fp32 tolerance; the never-attained all-roundings-aligned bound is
// ~1.7e-7 P per entry.  The bound is on EACH side, not on the product: the partial sums of |x~|^2 + |y~|^2 - 2 x~.y~ reach
// the larger norm, and a far X cluster over a compact Y would otherwise pass.
constexpr int MFMA_LDS_MIN_TILES = 64;
// gramian(k, x) takes the symmetric (upper triangle once) kernels from these sizes on; below, the general kernel is as fast — the
// symmetric launch has a floor of ~33 us (EQ) / ~90 us (MaternP) from its panel structure.  tools/sym_threshold_sweep.py, all
// entries / symmetric in us: EQ d = 3: n = 16384 38.8 / 40.0, 20000 55.9 / 51.5; EQ d = 8: 16384 48.9 / 47.0, 20000 70.7 / 60.0;
// MaternP(2): 12000 89.3 / 91.3, 16384 148 / 93.5; RQ: 12000 60.2 / 64.4, 16384 95.5 / 68.6.
constexpr int64_t MFMA_SYM_MIN_N_EQ = 18000;        // EQ, d <= 4
constexpr int64_t MFMA_SYM_MIN_N_EQ_WIDE = 15000;   // EQ, d > 4, and the cheap generic profiles (Cauchy, IMQ, Dot^p, ...)
constexpr int64_t MFMA_SYM_MIN_N_HEAVY = 12500;     // MaternP, RQ, composites
constexpr double MFMA_GATE = 126.0;
// the fp16 two-way split of the general EQ kernel (dense_mfma.hip, "Which split"): admitted up to this g^2 R^2.  Its worst case — both clouds
// on the gate's sphere and aligned (tests: "wide aligned") — measured 1.09e-5 row-wise at g^2 R^2 = 96 against the bf16 split's 7.2e-6, growing
// linearly with the bound: 8.2e-6 at 72, what the bf16 split shows at 110 of its 126 (tools/f16_split_ab.py: typical clouds differ by < 10 %)
constexpr double MFMA_F16_GATE = 72.0;
inline double mfma_gate_of(const covgram_ctx* ctx) { return MFMA_GATE * 0.01 * (double)ctx->mfma_gate_pct; }
inline double mfma_f16_gate_of(const covgram_ctx* ctx) { return MFMA_F16_GATE * 0.01 * (double)ctx->mfma_gate_pct; }
constexpr double GRAD_EXPAND_GATE_F32 = 128.0;   // the same form in fp32 (round 5): abs. error of s a few fp32 roundings of R^2 — the size of the fp32 matrix-core gate
constexpr double GRAD_EXPAND_GATE = 1000.0;   // gamma^2 R^2 up to which the fp64 gradient kernel expands |x - y|^2 (abs. error ~1e-16 R^2; grad_mvm.hpp)
double gate_radius2(const covgram_points* X, const covgram_points* Y);
int points_max_norm2(covgram_points* p);
bool mfma_eq_eligible(const covgram_ctx* ctx, const HostKernel& hk, const covgram_points* X, const covgram_points* Y, int nrhs);
int mvm_eq_mfma(covgram_ctx* ctx, const HostKernel& hk, const covgram_points* X, const covgram_points* Y, const float* a, float* y,
                double alpha, double beta);
// the other smooth fp32 profiles, dot-product kernels and several right-hand sides (dense_mfma.hpp)
bool mfma_eq_sym_eligible(const covgram_ctx* ctx, const HostKernel& hk, const covgram_points* X, const covgram_points* Y, int nrhs);
int mvm_eq_mfma_sym(covgram_ctx* ctx, const HostKernel& hk, const covgram_points* X, const float* a, float* y, double alpha, double beta,
                    int pfirst = 0, int pstride = 1, const covgram_kernel* kgen = nullptr);
bool mfma_gen_sym_eligible(const covgram_ctx* ctx, const HostKernel& hk, const covgram_points* X, const covgram_points* Y, int nrhs);
bool mfma_gen_eligible(const covgram_ctx* ctx, const HostKernel& hk, const covgram_points* X, const covgram_points* Y, int nrhs = 1);
bool sum_fused_applies(const covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, const covgram_points* Y, int nrhs);
int mvm_mfma_gen(covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, const covgram_points* Y, const float* a, int64_t lda,
                 float* y, int64_t ldy, int32_t nrhs, double alpha, double beta);

// Gramian(Dot(), x, y) = X Y' as two streaming passes over the point sets (lowrank.hip)
int mvm_dot_factored(covgram_ctx* ctx, const HostKernel& hk, const covgram_points* X, const covgram_points* Y, const void* a, int64_t lda, void* y,
                     int64_t ldy, int32_t nrhs, double alpha, double beta);

void ctx_blas_destroy(covgram_ctx* ctx);
int comm_destroy(covgram_ctx* ctx);      // comm.hip

int pad_dim(int d);          // next compiled D >= d, or -1
extern const int kDims[];    // compiled D list
extern const int kNumDims;

}  // namespace covgram
