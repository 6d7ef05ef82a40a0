// grad_mvm.hpp — O(n m d) MVM with the Gramian of a GradientKernel (reference: blockmul!
// src/gramian.jl:241-253, block mul! src/gradient.jl:86-92 (IsotropicInput) and :109-115
// (DotProductInput), phi', phi'' src/gradient.jl:584-600 as closed forms).
//
//   isotropic:    b_i += alpha * -2 (k1 a_j + 2 k2 r (r.a_j)),   r = x_i - y_j, (k1,k2) = (phi', phi'')(|r|^2)
//   dot product:  b_i += alpha *    (k1 a_j +   k2 y_j (x_i.a_j)),              (k1,k2) = (phi', phi'')(x_i.y_j)
//
// MI355X mapping: one lane owns one block row (x_i and the d-vector accumulator b_i live in VGPRs);
// the column stream P[j] = (gamma*y_j[0..D), a_j[0..D)) is wave-uniform and arrives through the
// scalar data cache as SGPR operands, so the 5d flops per block contain no cross-lane traffic, no
// LDS and no VGPR copies.  With gamma = 1/l the chain rule gives
//   b = -2 gamma^2 (psi' a + 2 psi'' r' (r'.a)),  r' = gamma r,  psi(s') = phi(s'/gamma^2),
// so the host folds -2 gamma^2 * scale into alpha.  Grid = (row blocks) × (J splits) with a
// deterministic second-pass reduction, as in dense_mvm.hpp.
#pragma once
#include "profiles.hpp"

namespace covgram {

constexpr int GRAD_THREADS = 256;

template <typename T, int FAM, int D, bool KEEP_R>
__global__ __launch_bounds__(GRAD_THREADS) void grad_mvm_kernel(const T* __restrict__ X, int64_t n, int32_t d,
                                                                const T* __restrict__ P, const T* __restrict__ P2,
                                                                int64_t m, T* __restrict__ out, int64_t npad,
                                                                int64_t jchunk, T alpha, T beta,
                                                                int32_t final_store, const KParams<T> kp) {
    constexpr bool ISO = (FAM != COVGRAM_DOT && FAM != COVGRAM_EXPDOT);
    const int tid = threadIdx.x;
    int64_t row = (int64_t)blockIdx.x * GRAD_THREADS + tid;
    const bool live = row < n;
    if (!live) row = n - 1;
    const int64_t j0 = (int64_t)blockIdx.y * jchunk;
    const int64_t j1 = (j0 + jchunk < m) ? (j0 + jchunk) : m;

    T x[D], b[D];
    {
        const T* xr = X + row * (int64_t)d;
        if (d == D) {   // common case: no padding, straight vector loads
#pragma unroll
            for (int l = 0; l < D; ++l) x[l] = xr[l] * kp.gamma;
        } else {
#pragma unroll
            for (int l = 0; l < D; ++l) x[l] = (l < d) ? xr[l] * kp.gamma : (T)0;
        }
#pragma unroll
        for (int l = 0; l < D; ++l) b[l] = (T)0;
    }

    // Dimensions in chunks of one 64-byte scalar load per operand; scheduling barriers bound the SGPR
    // live ranges (see dense_mvm.hpp), and the second pass re-reads its operands through P2 — the SAME
    // address passed as a second kernel argument — so the compiler cannot merge the two reads and keep
    // all 2*D scalars of pass 1 alive (SGPR spills) instead of re-fetching them from the scalar cache.
    constexpr int DC = 64 / (int)sizeof(T);
    const int cnt = (int)(j1 - j0);
    const T* __restrict__ p = P + j0 * (2 * D);
    const T* __restrict__ q = P2 + j0 * (2 * D);
    for (int jj = 0; jj < cnt; ++jj, p += 2 * D, q += 2 * D) {
        T s = (T)0, t = (T)0;
        T r[(ISO && KEEP_R) ? D : 1];
#pragma unroll
        for (int c0 = 0; c0 < D; c0 += DC) {
#pragma unroll
            for (int l = c0; l < ((c0 + DC < D) ? c0 + DC : D); ++l) {
                if constexpr (ISO) {
                    const T rl = x[l] - p[l];
                    if constexpr (KEEP_R) r[l] = rl;
                    s = cg_fma(rl, rl, s);
                    t = cg_fma(rl, p[D + l], t);
                } else {
                    s = cg_fma(x[l], p[l], s);
                    t = cg_fma(x[l], p[D + l], t);
                }
            }
            if constexpr (D > DC) {
                // pin both reductions at the chunk boundary: without this hipcc splits the s- and t-chains into separate
                // sweeps and keeps every r_l of the column alive in between (+2D VGPRs, occupancy 1)
                asm("" : "+v"(s), "+v"(t));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        T k1, k2;
        phi_derivs<FAM, T>(s, kp, k1, k2);
        const T c2 = ISO ? (T)2 * k2 * t : k2 * t;
#pragma unroll
        for (int c0 = 0; c0 < D; c0 += DC) {
#pragma unroll
            for (int l = c0; l < ((c0 + DC < D) ? c0 + DC : D); ++l) {
                T v;
                if constexpr (ISO) {
                    if constexpr (KEEP_R) v = r[l];
                    else v = x[l] - q[l];
                } else {
                    v = q[l];
                }
                b[l] = cg_fma(c2, v, cg_fma(k1, q[D + l], b[l]));
            }
            if constexpr (D > DC) __builtin_amdgcn_sched_barrier(0);
        }
    }

    if (!live) return;
    if (final_store) {
        T* yp = out + row * (int64_t)d;
#pragma unroll
        for (int l = 0; l < D; ++l) {
            if (l < d) {
                T v = alpha * b[l];
                if (beta != (T)0) v = cg_fma(beta, yp[l], v);
                yp[l] = v;
            }
        }
    } else {
        // partial slab [jsplit][D][npad]: lane-contiguous rows -> coalesced stores
        T* op = out + (int64_t)blockIdx.y * D * npad + row;
#pragma unroll
        for (int l = 0; l < D; ++l) op[(int64_t)l * npad] = b[l];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void grad_reduce_kernel(const T* __restrict__ partial, int64_t npad, int32_t D, int32_t jsplit,
                                                          T* __restrict__ y, int64_t n, int32_t d, T alpha, T beta) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int l = 0; l < d; ++l) {
        T s = (T)0;
        for (int sp = 0; sp < jsplit; ++sp) s += partial[((int64_t)sp * D + l) * npad + i];
        T* yp = y + i * (int64_t)d + l;
        T v = alpha * s;
        if (beta != (T)0) v = cg_fma(beta, *yp, v);
        *yp = v;
    }
}

// P[j][0..D) = gamma * Y[j][0..d), P[j][D..2D) = a[j*d + 0..d)   (zero padded)
template <typename T>
__global__ __launch_bounds__(256) void grad_pack_kernel(const T* __restrict__ Y, int64_t m, int32_t d, const T* __restrict__ A,
                                                        T* __restrict__ P, int32_t D, T gamma) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * (int64_t)D) return;
    const int64_t j = e / D;
    const int l = (int)(e - j * D);
    T* p = P + j * (int64_t)(2 * D);
    p[l] = (l < d) ? Y[j * (int64_t)d + l] * gamma : (T)0;
    p[D + l] = (l < d) ? A[j * (int64_t)d + l] : (T)0;
}

template <typename T, int FAM, int D>
static int launch_grad_one(const GradArgs& a) {
    const KParams<T> kp = cast_params<T>(a.hk->kp);
    dim3 grid((unsigned)((a.n + GRAD_THREADS - 1) / GRAD_THREADS), (unsigned)a.jsplit);
    const int final_store = (a.jsplit == 1) ? 1 : 0;
    // keep r = x - y in registers (4 flops per dim and block) while 3 d-vectors of state leave >= 2 waves per SIMD,
    // else recompute it in the second sweep (5 flops per dim, 2 d-vectors of state)
    constexpr int W3 = 3 * D * (int)(sizeof(T) / 4);
    constexpr bool CAN_KEEP = (W3 <= 200);
    bool keep = (W3 <= 144);
    if (a.keep_r == 0) keep = false;
    if (a.keep_r == 1) keep = CAN_KEEP;
    if constexpr (CAN_KEEP) {
        if (keep) {
            hipLaunchKernelGGL((grad_mvm_kernel<T, FAM, D, true>), grid, dim3(GRAD_THREADS), 0, a.stream, (const T*)a.X, a.n, a.d,
                               (const T*)a.P, (const T*)a.P, a.m, (T*)a.out, a.npad, a.jchunk, (T)a.alpha, (T)a.beta, final_store, kp);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { set_error("grad_mvm launch failed: %s", hipGetErrorString(e)); return COVGRAM_EHIP; }
            return COVGRAM_OK;
        }
    }
    hipLaunchKernelGGL((grad_mvm_kernel<T, FAM, D, false>), grid, dim3(GRAD_THREADS), 0, a.stream, (const T*)a.X, a.n,
                       a.d, (const T*)a.P, (const T*)a.P, a.m, (T*)a.out, a.npad, a.jchunk, (T)a.alpha, (T)a.beta, final_store, kp);
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
