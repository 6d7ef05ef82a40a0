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

union Frag {
    uint4 u;
    bf16x8 v;
};


constexpr unsigned BF16_ONE = 0x3F80u;

template <int FAM, int K2, int RT, int NR>
__global__ __launch_bounds__(64) void dense_mfma_gen_kernel(const float* __restrict__ X, int64_t n, int32_t d,
                                                            const uint4* __restrict__ PB, const float* __restrict__ W, int64_t ntile,
                                                            float* __restrict__ out, int64_t npad, int64_t ldy, int32_t nrhs,
                                                            int64_t tchunk, float alpha, float beta, int32_t final_store,
                                                            const float* __restrict__ Cn, const KParams<float> kp) {
    constexpr bool ISO = fam_is_iso<FAM>;
    const int l = threadIdx.x, t = l & 31, h = l >> 5;
    const int64_t i0 = (int64_t)blockIdx.x * (32 * RT);
    const float g = kp.gamma;
    Frag a[RT][K2];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        int64_t row = i0 + 32 * r + t;
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
            a[r][mm].u = f;
        }
    }

    const int64_t T0 = (int64_t)blockIdx.y * tchunk;
    const int64_t T1 = (T0 + tchunk < ntile) ? (T0 + tchunk) : ntile;
    float acc[RT][NR][16];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < NR; ++c)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[r][c][v] = 0.0f;

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
            for (int mm = 0; mm < K2; ++mm) D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r][mm].v, f[mm].v, D, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float s = D[v];
                if constexpr (FAM == COVGRAM_MATERNP) s = fmaxf(s, 0.0f);          // sqrt of a rounding-negative s
                float kv = Phi<FAM, float, false>::eval(s, kp);
                if (kp.power != 1) kv = ipow(kv, kp.power);
#pragma unroll
                for (int c = 0; c < NR; ++c) acc[r][c][v] = __builtin_fmaf(w[c], kv, acc[r][c][v]);
            }
        }
    };
    Frag f0[K2], f1[K2];
    float w0[NR], w1[NR];
    load_tile(0, f0, w0);
    for (int ti = 0; ti < nt; ti += 2) {
        load_tile(ti + 1, f1, w1);
        process(f0, w0);
        load_tile(ti + 2, f0, w0);
        if (ti + 1 < nt) process(f1, w1);
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
                float s = acc[r][c][v];
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
};

// returns the resident blocks per CU of the instance when `query` is set (no launch), COVGRAM_OK / error otherwise
template <int FAM, int K2, int RT, int NR>
static int mfma_gen_one(const MfmaArgs& a, bool query) {
    if (query) {
        static int cached = 0;
        if (!cached) {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, dense_mfma_gen_kernel<FAM, K2, RT, NR>, 64, 0) != hipSuccess || nb <= 0) nb = 16;
            cached = nb;
        }
        return cached;
    }
    hipLaunchKernelGGL((dense_mfma_gen_kernel<FAM, K2, RT, NR>), a.grid, dim3(64), 0, a.stream, a.X, a.n, a.d, a.PB, a.W, a.ntile, a.out, a.npad,
                       a.ldy, a.nrhs, a.tchunk, a.alpha, a.beta, a.final_store, a.Cn, cast_params<float>(a.hk->kp));
    return COVGRAM_OK;
}

template <int FAM, int K2>
static int mfma_gen_K(const MfmaArgs& a, bool query) {
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
