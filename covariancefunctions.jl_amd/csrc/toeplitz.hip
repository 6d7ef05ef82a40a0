// toeplitz.hip — O(N log N) MVM with SymmetricToeplitz / Toeplitz / Circulant Gramians.
//
// Reference: gramian(k, x::StepRangeLen, y::StepRangeLen) builds SymmetricToeplitz(k.(x[1], x)),
// Toeplitz(k.(x, y[1]), k.(x[1], y)) or Circulant(k.(x[1], x)) (src/gramian.jl:167-189); the MVM itself
// is ToeplitzMatrices 0.7.1 (+ FFTW 1.5.0), a third-party dependency whose source is not part of the
// reference tree.  Its published algorithm is restated here: embed T (n×m) in an N×N circulant
// C with first column  c = [vc_0 .. vc_{n-1}, 0 .., vr_{m-1} .. vr_1], then  T a = (C [a; 0])_{0..n-1}
// = irfft(rfft(c) ⊙ rfft([a; 0])).  Any N >= n+m-1 gives the identical product, so N is the next
// power of two (real-to-complex rocFFT, half the traffic of the complex transform the reference uses),
// and — unlike the reference, which re-plans and re-transforms c on every mul! — the plan and the
// spectrum of c (pre-divided by N) are cached in the handle.
//
// HBM-bound: per MVM one pad pass, R2C, one pointwise pass, C2R, one epilogue pass.
#include <rocfft/rocfft.h>

#include <algorithm>
#include <vector>

#include "common.hpp"

namespace covgram {

static int g_rocfft_users = 0;

#define CG_CHECK_FFT(expr)                                                                     \
    do {                                                                                       \
        rocfft_status _s = (expr);                                                             \
        if (_s != rocfft_status_success) {                                                     \
            ::covgram::set_error("%s failed: rocfft_status %d (%s:%d)", #expr, (int)_s, __FILE__, __LINE__); \
            return COVGRAM_EHIP;                                                               \
        }                                                                                      \
    } while (0)

// c[0..n) = vc, c[n..N-m+1) = 0, c[N-(m-1)+k] = vr[m-1-k]  (k = 0..m-2)
template <typename T>
__global__ __launch_bounds__(256) void embed_kernel(const T* __restrict__ vc, const T* __restrict__ vr, int64_t n, int64_t m,
                                                    int64_t N, T* __restrict__ c) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    T v = (T)0;
    if (i < n) v = vc[i];
    else if (i >= N - (m - 1)) v = vr[N - i];
    c[i] = v;
}

// buf[0..m) = a, buf[m..N) = 0   (16-byte vectorised where aligned)
template <typename T>
__global__ __launch_bounds__(256) void pad_kernel(const T* __restrict__ a, int64_t m, int64_t N, T* __restrict__ buf) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    buf[i] = (i < m) ? a[i] : (T)0;
}

// spec <- spec * s   (complex, interleaved); also used to pre-scale the cached spectrum by 1/N
template <typename T>
__global__ __launch_bounds__(256) void cmul_kernel(T* __restrict__ z, const T* __restrict__ s, int64_t nc, T scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    const T zr = z[2 * i], zi = z[2 * i + 1];
    if (s) {
        const T sr = s[2 * i], si = s[2 * i + 1];
        z[2 * i] = (zr * sr - zi * si) * scale;
        z[2 * i + 1] = (zr * si + zi * sr) * scale;
    } else {
        z[2 * i] = zr * scale;
        z[2 * i + 1] = zi * scale;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void toep_epilogue_kernel(const T* __restrict__ t, T* __restrict__ y, int64_t n, T alpha, T beta) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T v = alpha * t[i];
    if (beta != (T)0) v = fma_t(beta, y[i], v);
    y[i] = v;
}


// ================================================================================================
// Fast path for power-of-two embeddings (N >= 16384): a four-step FFT that never transposes.
//
// rocFFT runs the 2^23-point real transform of config C5 as SIX HBM passes per direction (three transposes, two
// length-2048 kernels, one real<->complex pass; profiles/r01_toeplitz_c5_kernel_stats.csv) because a strided
// (column) batch costs it 125-135 us against 29 us for the same batch laid out contiguously
// (profiles/r01_fft_probe.txt).  Here the packed complex signal z_j = x_2j + i x_2j+1 (M = N/2 points) is viewed as
// a 1024 x M' matrix:
//     colfft_kernel   length-1024 FFTs down the columns (stride M'): tiles of TW adjacent columns, i.e. coalesced
//                     128-byte row segments, whole tile in LDS (radix-4 DIT, five in-place stages), the inter-step
//                     twiddle W_M^(n' k1) applied on the way out; it reads the real input directly (no pad pass) and
//                     skips the zero half;
//     rocFFT          1024 contiguous length-M' transforms (ONE kernel);
// which leaves the spectrum in the permuted order  Z[k1 + 1024 k'] at [k1][k'].  The convolution does not care:
//     spectral_kernel real-FFT untangling, multiplication by the cached (equally permuted, 1/M-scaled) spectrum and
//                     re-tangling for the inverse, fused, one thread per conjugate pair (k, M-k);
// and the inverse is the transposed algorithm (the DFT matrix is symmetric): contiguous inverse rocFFT, then the
// inverse colfft_kernel, which ends in natural order and writes  y <- alpha t + beta y  directly.
// Five passes over HBM instead of fifteen.
// Round 2: the defaults are colfft16_kernel (the same column FFT as 16 x 4 x 16 with register butterflies, below) and, for
// M' = 4^L <= 4096, ONE fused kernel for the three middle passes (rowfft_fused_kernel; M' = 4096: rowfft16_fused_kernel) —
// three passes over zbuf per MVM; the column length is 512 or 2048 instead of 1024 when that is what makes M' a power of four
// <= 4096 (covgram_toeplitz::n1).  colfft_kernel / the rocFFT batches + spectral_kernel remain as the A/B arms
// (options toeplitz_colfft = 4, toeplitz_fused = 0) and serve the M' the fused kernels do not cover.
// ================================================================================================
template <typename T> struct V2T;
template <> struct V2T<float> { using type = float2; };
template <> struct V2T<double> { using type = double2; };

template <typename V> __device__ __forceinline__ V cmul(V a, V b) { return V{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
template <typename V> __device__ __forceinline__ V cconj(V a) { return V{a.x, -a.y}; }
template <typename V> __device__ __forceinline__ V cadd(V a, V b) { return V{a.x + b.x, a.y + b.y}; }
template <typename V> __device__ __forceinline__ V csub(V a, V b) { return V{a.x - b.x, a.y - b.y}; }
// a * (+i) and a * (-i)
template <typename V> __device__ __forceinline__ V cmuli(V a) { return V{-a.y, a.x}; }
template <typename V> __device__ __forceinline__ V cmulmi(V a) { return V{a.y, -a.x}; }

__device__ __forceinline__ int rev4_10(int x) {   // reverse the five base-4 digits of a 10-bit index
    return ((x & 0x3) << 8) | ((x & 0xC) << 4) | (x & 0x30) | ((x >> 4) & 0xC) | ((x >> 8) & 0x3);
}

constexpr int COLFFT_N1 = 1024;
constexpr int TWID_LB = 11;   // two-level twiddle: W_M^idx = thi[idx >> 11] * tlo[idx & 2047]

// Forward (INV = false): src (real, length src_len, implicitly zero beyond) -> zbuf[k1][n'] = twiddled column FFT.
// Inverse (INV = true):  zbuf[k1][n'] -> y[0..n) (real) with alpha/beta.
constexpr int COLFFT_THREADS = 1024;   // 4 waves per SIMD: one wave per SIMD reaches only a fraction of the LDS rates

template <typename T, int TW, bool INV>
__global__ __launch_bounds__(COLFFT_THREADS, 8) void colfft_kernel(const T* __restrict__ src, int64_t src_len,
                                                     typename V2T<T>::type* __restrict__ zbuf, int64_t Mp,
                                                     const typename V2T<T>::type* __restrict__ tw1024,
                                                     const typename V2T<T>::type* __restrict__ tlo,
                                                     const typename V2T<T>::type* __restrict__ thi, T* __restrict__ y, int64_t n,
                                                     T alpha, T beta) {
    using V = typename V2T<T>::type;
    __shared__ V buf[COLFFT_N1 * TW];
    __shared__ V stw[COLFFT_N1];
    const int tid = threadIdx.x;
    // XCD-aware tile map (speed only): within each group of 16 workgroups, b and b+8 take the two halves of one line
    const unsigned bid = blockIdx.x;
    const unsigned tile = (gridDim.x % 16 == 0) ? (2 * (8 * (bid / 16) + (bid % 8)) + ((bid / 8) % 2)) : bid;
    const int64_t col0 = (int64_t)tile * TW;
    constexpr int NT = COLFFT_THREADS;
    for (int t = tid; t < COLFFT_N1; t += NT) stw[t] = INV ? cconj(tw1024[t]) : tw1024[t];
    // ---- load fused with the first radix-4 stage (its twiddles are all 1) --------------------------------------------------
    // DIT stage 0 combines the LDS slots 4g .. 4g+3, i.e. (digit reversal) the rows rr, rr+256, rr+512, rr+768 with
    // rr = rev4(g): each thread fetches those four rows of one column straight from HBM and writes the butterfly outputs.
    for (int e = tid; e < 256 * TW; e += NT) {
        const int c = e % TW, rr = e / TW;
        const int64_t np = col0 + c;
        V v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = rr + 256 * q;
            if constexpr (!INV) {
                const int64_t j = (int64_t)row * Mp + np;             // packed complex index: z_j = x[2j] + i x[2j+1]
                v[q].x = (2 * j < src_len) ? src[2 * j] : (T)0;
                v[q].y = (2 * j + 1 < src_len) ? src[2 * j + 1] : (T)0;
            } else {
                const int64_t idx = np * (int64_t)row;                // n' * k1 < M
                const V w = cmul(thi[idx >> TWID_LB], tlo[idx & ((1 << TWID_LB) - 1)]);
                v[q] = cmul(zbuf[(int64_t)row * Mp + np], cconj(w));
            }
        }
        const V apc = cadd(v[0], v[2]), amc = csub(v[0], v[2]), bpd = cadd(v[1], v[3]), bmd = csub(v[1], v[3]);
        const V ib = INV ? cmuli(bmd) : cmulmi(bmd);
        const int g = ((rr & 0x3) << 6) | ((rr & 0xC) << 2) | ((rr >> 2) & 0xC) | ((rr >> 6) & 0x3);   // rev4 of the 8-bit index
        buf[(4 * g + 0) * TW + c] = cadd(apc, bpd);
        buf[(4 * g + 1) * TW + c] = cadd(amc, ib);
        buf[(4 * g + 2) * TW + c] = csub(apc, bpd);
        buf[(4 * g + 3) * TW + c] = csub(amc, ib);
    }
    __syncthreads();
    // ---- three in-place radix-4 DIT stages in LDS (L = 4, 16, 64) ------------------------------------------------------------
    const int c = tid % TW, bb = tid / TW;
#pragma unroll 1
    for (int s = 1; s < 4; ++s) {
        const int L = 1 << (2 * s);
#pragma unroll
        for (int r = 0; r < 256 * TW / NT; ++r) {
            const int bf = bb + (NT / TW) * r;                         // butterfly 0..255 of this column
            const int j = bf & (L - 1), g = bf >> (2 * s);
            const int i0 = (g * 4 * L + j) * TW + c;
            const int tq = j << (8 - 2 * s);                           // j * (256 / L)
            const V a = buf[i0];
            const V b = cmul(stw[tq], buf[i0 + L * TW]);
            const V cc = cmul(stw[2 * tq], buf[i0 + 2 * L * TW]);
            const V d = cmul(stw[3 * tq], buf[i0 + 3 * L * TW]);
            const V apc = cadd(a, cc), amc = csub(a, cc), bpd = cadd(b, d), bmd = csub(b, d);
            // forward: y1 = (a - c) - i (b - d), y3 = (a - c) + i (b - d); inverse: signs of i swapped
            const V ib = INV ? cmuli(bmd) : cmulmi(bmd);
            buf[i0] = cadd(apc, bpd);
            buf[i0 + L * TW] = cadd(amc, ib);
            buf[i0 + 2 * L * TW] = csub(apc, bpd);
            buf[i0 + 3 * L * TW] = csub(amc, ib);
        }
        __syncthreads();
    }
    // ---- last stage (L = 256) fused with the store: outputs are the rows j, j+256, j+512, j+768 in natural order ---------------
    for (int e = tid; e < 256 * TW; e += NT) {
        const int cq = e % TW, j = e / TW;
        const int64_t np = col0 + cq;
        const V a = buf[j * TW + cq];
        const V b = cmul(stw[j], buf[(j + 256) * TW + cq]);
        const V cc = cmul(stw[2 * j], buf[(j + 512) * TW + cq]);
        const V d = cmul(stw[3 * j], buf[(j + 768) * TW + cq]);
        const V apc = cadd(a, cc), amc = csub(a, cc), bpd = cadd(b, d), bmd = csub(b, d);
        const V ib = INV ? cmuli(bmd) : cmulmi(bmd);
        V o[4] = {cadd(apc, bpd), cadd(amc, ib), csub(apc, bpd), csub(amc, ib)};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = j + 256 * q;
            if constexpr (!INV) {
                const int64_t idx = np * (int64_t)row;
                const V w = cmul(thi[idx >> TWID_LB], tlo[idx & ((1 << TWID_LB) - 1)]);
                zbuf[(int64_t)row * Mp + np] = cmul(o[q], w);
            } else {
                const int64_t jj = (int64_t)row * Mp + np;            // natural order: y[2jj], y[2jj+1]
                if (2 * jj < n) {
                    T ov = alpha * o[q].x;
                    if (beta != (T)0) ov = fma_t(beta, y[2 * jj], ov);
                    y[2 * jj] = ov;
                }
                if (2 * jj + 1 < n) {
                    T ov = alpha * o[q].y;
                    if (beta != (T)0) ov = fma_t(beta, y[2 * jj + 1], ov);
                    y[2 * jj + 1] = ov;
                }
            }
        }
    }
}

// position of the conjugate partner M-k of k = k1 + n1 k' in the [k1][k'] layout (n1 = column length: 512, 1024 or 2048)
__device__ __forceinline__ int64_t partner_pos(int k1, int64_t kp, int64_t Mp, int n1) {
    const int k1p = (n1 - k1) & (n1 - 1);
    const int64_t kpp = (k1 != 0) ? (Mp - 1 - kp) : ((Mp - kp) & (Mp - 1));
    return (int64_t)k1p * Mp + kpp;
}

// S[pos] = X[k] * scale with X the half spectrum of the real signal whose packed transform is D; S[M] = X[M] * scale (Nyquist)
template <typename T>
__global__ __launch_bounds__(256) void half_spectrum_kernel(const typename V2T<T>::type* __restrict__ D, int64_t Mp,
                                                            const typename V2T<T>::type* __restrict__ tA,
                                                            const typename V2T<T>::type* __restrict__ tB,
                                                            typename V2T<T>::type* __restrict__ S, T scale, int n1) {
    using V = typename V2T<T>::type;
    const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pos >= n1 * Mp) return;
    const int k1 = (int)(pos / Mp);
    const int64_t kp = pos - (int64_t)k1 * Mp;
    const V Z = D[pos], Zp = D[partner_pos(k1, kp, Mp, n1)];
    const V w = cmul(tA[k1], tB[kp]);                                   // exp(-i pi k / M)
    const V e = cadd(Z, cconj(Zp)), o = csub(Z, cconj(Zp));
    const V wo = cmul(w, o);
    V X{(T)0.5 * (e.x + wo.y), (T)0.5 * (e.y - wo.x)};                  // 0.5 e - 0.5 i (w o)
    if (pos == 0) { X = V{Z.x + Z.y, (T)0}; S[(int64_t)n1 * Mp] = V{(Z.x - Z.y) * scale, (T)0}; }
    S[pos] = V{X.x * scale, X.y * scale};
}

// D <- the packed spectrum of the product signal, in place, one thread per conjugate pair
template <typename T>
__global__ __launch_bounds__(256) void spectral_kernel(typename V2T<T>::type* __restrict__ D, int64_t Mp,
                                                       const typename V2T<T>::type* __restrict__ S,
                                                       const typename V2T<T>::type* __restrict__ tA,
                                                       const typename V2T<T>::type* __restrict__ tB, int n1) {
    using V = typename V2T<T>::type;
    const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;        // rows k1 = 0 .. n1 / 2
    if (pos >= (int64_t)(n1 / 2 + 1) * Mp) return;
    const int k1 = (int)(pos / Mp);
    const int64_t kp = pos - (int64_t)k1 * Mp;
    const int64_t ppos = partner_pos(k1, kp, Mp, n1);
    if ((k1 == 0 || k1 == n1 / 2) && ppos < pos) return;        // self-paired rows: the lower position owns the pair
    const V Z = D[pos];
    if (pos == 0) {                                                     // k = 0 and the Nyquist bin share Z[0]
        const T X0 = Z.x + Z.y, XM = Z.x - Z.y;
        const T Y0 = X0 * S[0].x, YM = XM * S[(int64_t)n1 * Mp].x;
        D[0] = V{(T)0.5 * (Y0 + YM), (T)0.5 * (Y0 - YM)};
        return;
    }
    const V Zp = D[ppos];
    const V w = cmul(tA[k1], tB[kp]);                                   // w_k = exp(-i pi k / M);  w_{M-k} = -conj(w_k)
    const V wc = cconj(w);
    // untangle: X[k] = 0.5 (Z + conj Zp) - 0.5 i w (Z - conj Zp);  X[M-k] = 0.5 (Zp + conj Z) + 0.5 i conj(w) (Zp - conj Z)
    const V e = cadd(Z, cconj(Zp)), o = csub(Z, cconj(Zp));
    const V wo = cmul(w, o);
    const V X{(T)0.5 * (e.x + wo.y), (T)0.5 * (e.y - wo.x)};
    const V ep = cconj(e), op = V{-o.x, o.y};                           // Zp + conj Z = conj(e);  Zp - conj Z = -conj(o)
    const V wop = cmul(wc, op);
    const V Xp{(T)0.5 * (ep.x - wop.y), (T)0.5 * (ep.y + wop.x)};       // 0.5 ep + 0.5 i (conj(w) op)
    const V Y = cmul(X, S[pos]), Yp = cmul(Xp, S[ppos]);
    // tangle: Z'[k] = 0.5 (Y + conj Yp) + 0.5 i conj(w) (Y - conj Yp);  Z'[M-k] = 0.5 (Yp + conj Y) - 0.5 i w (Yp - conj Y)
    const V f = cadd(Y, cconj(Yp)), h = csub(Y, cconj(Yp));
    const V wh = cmul(wc, h);
    D[pos] = V{(T)0.5 * (f.x - wh.y), (T)0.5 * (f.y + wh.x)};
    if (ppos != pos) {
        const V fp = cconj(f), hp = V{-h.x, h.y};
        const V whp = cmul(w, hp);
        D[ppos] = V{(T)0.5 * (fp.x + whp.y), (T)0.5 * (fp.y - whp.x)};
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The three middle passes as ONE kernel (M' = 4^L <= 4096): forward row FFT, the spectral step, inverse row FFT.
// A workgroup holds TWO rows of zbuf in LDS — k1 and its conjugate partner 1024 - k1 (rows 0 and 512 are self-paired and share
// a workgroup) — so every conjugate pair the spectral step touches is resident:  load -> L radix-4 DIF stages (natural in,
// digit-reversed out) -> untangle * spectrum * re-tangle on the digit-reversed positions (rev(M'-1-k') = M'-1-rev(k')) ->
// L radix-4 DIT stages with conjugated twiddles (digit-reversed in, natural out) -> store.  No digit-reversal pass, one HBM
// read and one write of zbuf instead of three of each (rocFFT batch, spectral_kernel, rocFFT batch).  Twiddles: a quarter
// table W_M'^r, r < M'/4, in LDS; the other quadrants are multiplications by -i, -1, i.
// ------------------------------------------------------------------------------------------------------------------------
template <int L>
__device__ __forceinline__ int rev4(int x) {
    int r = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) { r = (r << 2) | (x & 3); x >>= 2; }
    return r;
}

// W^idx from the quarter table (idx < 3 Q): forward W = exp(-2 pi i / M'), W^Q = -i
template <typename V, bool INV>
__device__ __forceinline__ V quarter_tw(const V* __restrict__ stw, int idx, int Q) {
    const int quad = idx / Q, r = idx & (Q - 1);
    V w = stw[r];
    if (quad == 1) w = V{w.y, -w.x};
    else if (quad == 2) w = V{-w.x, -w.y};
    if (INV) w.y = -w.y;
    return w;
}

template <typename T, int L>
__global__ __launch_bounds__((1 << (2 * L)) / 4, (sizeof(T) == 4 ? 8 : 4)) void rowfft_fused_kernel(typename V2T<T>::type* __restrict__ zbuf,
                                                                          const typename V2T<T>::type* __restrict__ S,
                                                                          const typename V2T<T>::type* __restrict__ tA,
                                                                          const typename V2T<T>::type* __restrict__ tB,
                                                                          const typename V2T<T>::type* __restrict__ twr, int n1) {
    using V = typename V2T<T>::type;
    constexpr int Mp = 1 << (2 * L), Q = Mp / 4, NT = Q;
    // LDS rows are padded by one element per 16 (slot(i) = i + i/16): with 16-byte elements a quarter-wave then always
    // touches 16 distinct 16-byte bank groups — for the unit-stride stages (4 consecutive elements per lane) as well as for
    // the long-stride ones; the span-4 stage additionally walks the groups, not the offsets, along the lanes.
    constexpr int MpP = Mp + Mp / 16;
    auto P = [](int i) { return i + (i >> 4); };
    __shared__ V rowA[MpP];
    __shared__ V rowB[MpP];
    __shared__ V stw[Q];
    const int tid = threadIdx.x;
    const int wg = blockIdx.x;
    const int kA = (wg == 0) ? 0 : wg, kB = (wg == 0) ? n1 / 2 : n1 - wg;
    V* __restrict__ gA = zbuf + (int64_t)kA * Mp;
    V* __restrict__ gB = zbuf + (int64_t)kB * Mp;
#pragma unroll
    for (int i = 0; i < 4; ++i) { rowA[P(tid + i * NT)] = gA[tid + i * NT]; rowB[P(tid + i * NT)] = gB[tid + i * NT]; }
    stw[tid] = twr[tid];
    __syncthreads();

    // ---- forward: radix-4 decimation in frequency ------------------------------------------------------------------------
#pragma unroll 1
    for (int s = 0; s < L; ++s) {
        const int lg = 2 * (L - 1 - s);                  // log2 of the span Ls = M' / 4^(s+1)
        const int Ls = 1 << lg;
        const int j = (Ls == 4) ? (tid >> (2 * L - 4)) : (tid & (Ls - 1));
        const int g = (Ls == 4) ? (tid & ((1 << (2 * L - 4)) - 1)) : (tid >> lg);
        const int i0 = g * 4 * Ls + j;
        const int e = j << (2 * s);                      // twiddle exponent j * 4^s (in units of W_M')
        const V w1 = quarter_tw<V, false>(stw, e, Q), w2 = quarter_tw<V, false>(stw, 2 * e, Q), w3 = quarter_tw<V, false>(stw, 3 * e, Q);
        const int p0 = P(i0), p1 = P(i0 + Ls), p2 = P(i0 + 2 * Ls), p3 = P(i0 + 3 * Ls);
#pragma unroll
        for (int rw = 0; rw < 2; ++rw) {
            V* x = rw ? rowB : rowA;
            const V a = x[p0], b = x[p1], c = x[p2], d = x[p3];
            const V t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = csub(b, d);
            x[p0] = cadd(t0, t2);
            x[p1] = cmul(cadd(t1, cmulmi(t3)), w1);                 // (a - c) - i (b - d)
            x[p2] = cmul(csub(t0, t2), w2);
            x[p3] = cmul(cadd(t1, cmuli(t3)), w3);                  // (a - c) + i (b - d)
        }
        __syncthreads();
    }

    // ---- spectral step on conjugate pairs (same algebra as spectral_kernel; frequency k' lives at slot rev(k')) ----------
    auto pair = [&](V& zs, V& zps, bool self, int k1, int kp, int64_t pos, int64_t ppos) {
        const V Z = zs;
        if (pos == 0) {                                  // k = 0 and the Nyquist bin share Z[0]
            const T X0 = Z.x + Z.y, XM = Z.x - Z.y;
            const T Y0 = X0 * S[0].x, YM = XM * S[(int64_t)n1 * Mp].x;
            zs = V{(T)0.5 * (Y0 + YM), (T)0.5 * (Y0 - YM)};
            return;
        }
        const V Zp = zps;
        const V w = cmul(tA[k1], tB[kp]);
        const V wc = cconj(w);
        const V e = cadd(Z, cconj(Zp)), o = csub(Z, cconj(Zp));
        const V wo = cmul(w, o);
        const V X{(T)0.5 * (e.x + wo.y), (T)0.5 * (e.y - wo.x)};
        const V ep = cconj(e), op = V{-o.x, o.y};
        const V wop = cmul(wc, op);
        const V Xp{(T)0.5 * (ep.x - wop.y), (T)0.5 * (ep.y + wop.x)};
        const V Y = cmul(X, S[pos]), Yp = cmul(Xp, S[ppos]);
        const V f = cadd(Y, cconj(Yp)), h = csub(Y, cconj(Yp));
        const V wh = cmul(wc, h);
        zs = V{(T)0.5 * (f.x - wh.y), (T)0.5 * (f.y + wh.x)};
        if (!self) {
            const V fp = cconj(f), hp = V{-h.x, h.y};
            const V whp = cmul(w, hp);
            zps = V{(T)0.5 * (fp.x + whp.y), (T)0.5 * (fp.y - whp.x)};
        }
    };
    if (wg != 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kp = tid + i * NT;
            const int p = rev4<L>(kp);
            pair(rowA[P(p)], rowB[P(Mp - 1 - p)], false, kA, kp, (int64_t)kA * Mp + kp, (int64_t)kB * Mp + (Mp - 1 - kp));
        }
    } else {
        // row 0: (0, k') pairs with (0, (M' - k') mod M'); row 512: (512, k') with (512, M' - 1 - k'); the lower position owns the pair
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kp = tid + i * NT;
            const int kq = (Mp - kp) & (Mp - 1);
            if (kp <= kq) pair(rowA[P(rev4<L>(kp))], rowA[P(rev4<L>(kq))], kp == kq, 0, kp, (int64_t)kp, (int64_t)kq);
            const int kr = Mp - 1 - kp;
            if (kp < kr) {
                const int p = rev4<L>(kp);
                pair(rowB[P(p)], rowB[P(Mp - 1 - p)], false, n1 / 2, kp, (int64_t)(n1 / 2) * Mp + kp, (int64_t)(n1 / 2) * Mp + kr);
            }
        }
    }
    __syncthreads();

    // ---- inverse: radix-4 decimation in time, conjugated twiddles --------------------------------------------------------
#pragma unroll 1
    for (int s = 0; s < L; ++s) {
        const int lg = 2 * s;
        const int Ls = 1 << lg;
        const int j = (Ls == 4) ? (tid >> (2 * L - 4)) : (tid & (Ls - 1));
        const int g = (Ls == 4) ? (tid & ((1 << (2 * L - 4)) - 1)) : (tid >> lg);
        const int i0 = g * 4 * Ls + j;
        const int e = j << (2 * (L - 1 - s));            // j * M' / (4 Ls)
        const V w1 = quarter_tw<V, true>(stw, e, Q), w2 = quarter_tw<V, true>(stw, 2 * e, Q), w3 = quarter_tw<V, true>(stw, 3 * e, Q);
        const int p0 = P(i0), p1 = P(i0 + Ls), p2 = P(i0 + 2 * Ls), p3 = P(i0 + 3 * Ls);
#pragma unroll
        for (int rw = 0; rw < 2; ++rw) {
            V* x = rw ? rowB : rowA;
            const V a = x[p0], b = cmul(w1, x[p1]), c = cmul(w2, x[p2]), d = cmul(w3, x[p3]);
            const V t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = csub(b, d);
            x[p0] = cadd(t0, t2);
            x[p1] = cadd(t1, cmuli(t3));                            // (a - c) + i (b - d)
            x[p2] = csub(t0, t2);
            x[p3] = cadd(t1, cmulmi(t3));                           // (a - c) - i (b - d)
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { gA[tid + i * NT] = rowA[P(tid + i * NT)]; gB[tid + i * NT] = rowB[P(tid + i * NT)]; }
}

// ------------------------------------------------------------------------------------------------------------------------
// The same fused kernel for M' = 4096 with RADIX-16 stages (round 2): a thread carries 16 elements of one row through a
// 16-point DFT in registers (two radix-4 layers); the FIRST forward stage takes its inputs straight from global memory and the
// LAST inverse stage writes its outputs there (span 256: coalesced along the butterfly index), so a row makes 2 + 2 round trips
// through LDS instead of 6 + 6 plus a load and a store pass, and a butterfly's 15 twiddles W^(e q) come from ONE table read and
// repeated multiplication (<= 15 roundings; the table reads were scattered 16-byte LDS accesses).  512 threads: thread (row, b)
// owns butterfly b of its row.  Measured on C5 (fp64): 128 -> 103 us per MVM, fp32 85 -> 75 us (tools/c5_probe.py); radix-16 stages
// alone, with the load / store passes and the table twiddles kept, measured 129 us — the passes, not the stage count, were the cost.  Forward = decimation in frequency (natural in,
// base-16 digit-reversed out), inverse = decimation in time with conjugated twiddles; frequency k' lives at slot rev16(k')
// between the two, and rev16(M' - 1 - k') = M' - 1 - rev16(k') as for any digit reversal.
// ------------------------------------------------------------------------------------------------------------------------
// W^idx from the quarter table, any idx < 4 Q (forward W = exp(-2 pi i / M'): W^Q = -i, W^2Q = -1, W^3Q = i)
template <typename V, bool INV>
__device__ __forceinline__ V full_tw(const V* __restrict__ stw, int idx, int Q) {
    const int quad = idx / Q, r = idx & (Q - 1);
    V w = stw[r];
    if (quad == 1) w = V{w.y, -w.x};
    else if (quad == 2) w = V{-w.x, -w.y};
    else if (quad == 3) w = V{-w.y, w.x};
    if (INV) w.y = -w.y;
    return w;
}

// in-place 16-point DFT, output X[q] at index q (forward: e^{-2 pi i p q / 16}; INV: conjugate), as two radix-4 layers:
// T[p1][q2] = sum_p2 w4^(p2 q2) x[p1 + 4 p2];  X[q2 + 4 q1] = sum_p1 w4^(p1 q1) (w16^(p1 q2) T[p1][q2])
template <typename V, bool INV>
__device__ __forceinline__ void dft16(V (&x)[16]) {
    using T = decltype(x[0].x);
    constexpr double C1 = 0.92387953251128675613, S1 = 0.38268343236508977173, C2 = 0.70710678118654752440;   // cos / sin of pi/8, pi/4
    V t[16];
#pragma unroll
    for (int p1 = 0; p1 < 4; ++p1) {
        const V a = x[p1], b = x[p1 + 4], c = x[p1 + 8], d = x[p1 + 12];
        const V apc = cadd(a, c), amc = csub(a, c), bpd = cadd(b, d), bmd = csub(b, d);
        const V ib = INV ? cmuli(bmd) : cmulmi(bmd);
        t[p1 * 4 + 0] = cadd(apc, bpd);
        t[p1 * 4 + 1] = cadd(amc, ib);
        t[p1 * 4 + 2] = csub(apc, bpd);
        t[p1 * 4 + 3] = csub(amc, ib);
    }
    // twiddles w16^(p1 q2), p1 q2 in {1, 2, 3, 4, 6, 9}; forward w16^k = (cos(k pi/8), -sin(k pi/8))
    auto tw = [](V v, double c, double sn) { const T cr = (T)c, si = (T)(INV ? sn : -sn); return V{v.x * cr - v.y * si, v.x * si + v.y * cr}; };
    t[1 * 4 + 1] = tw(t[1 * 4 + 1], C1, S1);          // k = 1
    t[1 * 4 + 2] = tw(t[1 * 4 + 2], C2, C2);          // k = 2
    t[1 * 4 + 3] = tw(t[1 * 4 + 3], S1, C1);          // k = 3
    t[2 * 4 + 1] = tw(t[2 * 4 + 1], C2, C2);          // k = 2
    t[2 * 4 + 2] = INV ? cmuli(t[2 * 4 + 2]) : cmulmi(t[2 * 4 + 2]);   // k = 4: -i (forward)
    t[2 * 4 + 3] = tw(t[2 * 4 + 3], -C2, C2);         // k = 6
    t[3 * 4 + 1] = tw(t[3 * 4 + 1], S1, C1);          // k = 3
    t[3 * 4 + 2] = tw(t[3 * 4 + 2], -C2, C2);         // k = 6
    t[3 * 4 + 3] = tw(t[3 * 4 + 3], -C1, -S1);        // k = 9
#pragma unroll
    for (int q2 = 0; q2 < 4; ++q2) {
        const V a = t[q2], b = t[4 + q2], c = t[8 + q2], d = t[12 + q2];
        const V apc = cadd(a, c), amc = csub(a, c), bpd = cadd(b, d), bmd = csub(b, d);
        const V ib = INV ? cmuli(bmd) : cmulmi(bmd);
        x[q2 + 0] = cadd(apc, bpd);
        x[q2 + 4] = cadd(amc, ib);
        x[q2 + 8] = csub(apc, bpd);
        x[q2 + 12] = csub(amc, ib);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// colfft16_kernel: the column FFT (length 1024, stride M') as 16 x 4 x 16 with the radix-16 butterflies in registers.
//
// Same job and same tile shape as colfft_kernel (TW adjacent columns = 64-byte row segments, inter-step twiddle W_M^(n' k1),
// real input read directly, y <- alpha t + beta y on the way out) with two LDS exchanges instead of four passes:
//   forward (decimation in frequency):  rows b + 64 q from HBM -> dft16 over q, x W_1024^(b q') -> LDS
//                                       radix 4 over the span 16 inside each 64-block, x W_64^(j m')      (in place)
//                                       dft16 over the span 1 -> rows q' + 16 m' + 64 q'' to HBM, x W_M^(n' row)
//   inverse (decimation in time):       the transposed graph, conjugated: rows k0 + 64 q'' from HBM -> ... -> rows b + 64 q.
// The frequency digits come out of a DIF in reversed order; the last stage holds them in registers and simply stores each
// value to its true row (rows are 64-byte segments 16 M' bytes apart either way), so no reordering pass exists.
// Position p of a column lives at LDS row p + (p >> 4): the span-1 stage (lanes 16 positions apart) then falls on all four
// 64-byte quarters of the 256-byte bank row inside every lane group of ds_read_b128 (MI355X_MICROARCH.md, LDS) and the
// other two stages stay contiguous.  Twiddles: one table read per butterfly, powers by squaring / one multiply (depth <= 6).
// 64 TW threads, 68 KB of LDS: two workgroups per CU, each thread 16 loads in flight.
// ------------------------------------------------------------------------------------------------------------------------
template <typename V> __device__ __forceinline__ void pow_chain16(V w1, V (&w)[16]) {   // w[q] = w1^q, q = 1..15 (w[0] unused)
    w[1] = w1;
#pragma unroll
    for (int q = 2; q < 16; ++q) w[q] = (q & 1) ? cmul(w[q - 1], w1) : cmul(w[q / 2], w[q / 2]);
}

template <typename V, bool INV> __device__ __forceinline__ void dft4(V (&v)[4]) {
    const V apc = cadd(v[0], v[2]), amc = csub(v[0], v[2]), bpd = cadd(v[1], v[3]), bmd = csub(v[1], v[3]);
    const V ib = INV ? cmuli(bmd) : cmulmi(bmd);
    v[0] = cadd(apc, bpd); v[1] = cadd(amc, ib); v[2] = csub(apc, bpd); v[3] = csub(amc, ib);
}

template <typename V, bool INV> __device__ __forceinline__ void dft2(V (&v)[2]) { const V a = v[0], b = v[1]; v[0] = cadd(a, b); v[1] = csub(a, b); }
template <typename V, bool INV> __device__ __forceinline__ void dft8(V (&v)[8]) {
    // two radix-4 transforms of the even / odd entries, combined with W_8^k (forward W_8 = (1 - i) / sqrt 2; INV: conjugate)
    using T = decltype(v[0].x);
    V e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
    dft4<V, INV>(e); dft4<V, INV>(o);
    constexpr double C = 0.70710678118654752440;
    const T c = (T)C, sn = (T)(INV ? C : -C);
    o[1] = V{o[1].x * c - o[1].y * sn, o[1].x * sn + o[1].y * c};                   // W_8^1
    o[2] = INV ? cmuli(o[2]) : cmulmi(o[2]);                                        // W_8^2 = -i (forward)
    o[3] = V{-o[3].x * c - o[3].y * sn, o[3].x * sn - o[3].y * c};                  // W_8^3 = (-1 - i) / sqrt 2 (forward)
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = cadd(e[k], o[k]); v[k + 4] = csub(e[k], o[k]); }
}
template <typename V, bool INV, int R> __device__ __forceinline__ void dftR(V (&v)[R]) {
    if constexpr (R == 2) dft2<V, INV>(v);
    else if constexpr (R == 4) dft4<V, INV>(v);
    else dft8<V, INV>(v);
}

// R = 2, 4, 8: column length N1 = 256 R = 512, 1024, 2048 as 16 x R x 16 (the middle stage is a radix-R butterfly over the span 16
// inside each block of 16 R positions; block q' sits at LDS rows 17 R q' ...).  16 R TW threads; 35 / 70 / 139 KB of LDS.
template <typename T, int TW, bool INV, int R>
__global__ __launch_bounds__(16 * R * TW, (R == 8 ? 1 : 2)) void colfft16_kernel(const T* __restrict__ src, int64_t src_len,
                                                               typename V2T<T>::type* __restrict__ zbuf, int64_t Mp,
                                                               const typename V2T<T>::type* __restrict__ twN1,
                                                               const typename V2T<T>::type* __restrict__ tlo,
                                                               const typename V2T<T>::type* __restrict__ thi, T* __restrict__ y, int64_t n,
                                                               T alpha, T beta) {
    using V = typename V2T<T>::type;
    constexpr int N1 = 256 * R, BLK = 16 * R, PBS = 17 * R;   // positions per stage-1 block, LDS rows per block
    constexpr int ROWS = N1 + N1 / 16;
    constexpr int NB = 16 / R;                                // radix-R butterflies per thread
    __shared__ V buf[ROWS * TW];
    const int tid = threadIdx.x;
    const unsigned bid = blockIdx.x;
    const unsigned tile = (gridDim.x % 16 == 0) ? (2 * (8 * (bid / 16) + (bid % 8)) + ((bid / 8) % 2)) : bid;   // as colfft_kernel
    const int c = tid % TW, r = tid / TW;                  // r = 0..16 R - 1: the butterfly of this thread in the radix-16 stages
    const int64_t np = (int64_t)tile * TW + c;             // column n'
    auto twM = [&](int64_t idx) { return cmul(thi[idx >> TWID_LB], tlo[idx & ((1 << TWID_LB) - 1)]); };   // W_M^idx
    const int k0 = (r / R) + 16 * (r % R);                 // span-1 stage: butterfly r = R q' + m' owns the rows q' + 16 m' + 16 R q''
    const int j = r & 15, gw = r >> 4;                     // radix-R stage: butterflies (g = NB gw + i, j)
    V x[16], w[16];
    if constexpr (!INV) {
        // ---- rows b + 16 R q of column n' (b = r): z_j = x[2j] + i x[2j+1], zero beyond src_len ------------------------------
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int64_t e = 2 * ((int64_t)(r + BLK * q) * Mp + np);
            x[q].x = (e < src_len) ? src[e] : (T)0;
            x[q].y = (e + 1 < src_len) ? src[e + 1] : (T)0;
        }
        dft16<V, false>(x);
        pow_chain16(twN1[r], w);
        const int p0 = (r + (r >> 4)) * TW + c;
        buf[p0] = x[0];
#pragma unroll
        for (int q = 1; q < 16; ++q) buf[p0 + PBS * q * TW] = cmul(x[q], w[q]);
        __syncthreads();
        {
            V wm[R];                                        // W_(16 R)^(j m') = W_N1^(16 j m')
            wm[1] = twN1[16 * j];
#pragma unroll
            for (int m = 2; m < R; ++m) wm[m] = (m & 1) ? cmul(wm[m - 1], wm[1]) : cmul(wm[m / 2], wm[m / 2]);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int p = (PBS * (NB * gw + i) + j) * TW + c;
                V v[R];
#pragma unroll
                for (int m = 0; m < R; ++m) v[m] = buf[p + 17 * m * TW];
                dftR<V, false, R>(v);
                buf[p] = v[0];
#pragma unroll
                for (int m = 1; m < R; ++m) buf[p + 17 * m * TW] = cmul(v[m], wm[m]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = buf[(17 * r + q) * TW + c];
        dft16<V, false>(x);
        // inter-step twiddle W_M^(n' (k0 + 16 R q)) = W_M^(n' k0) (W_M^(16 R n'))^q
        pow_chain16(twM(BLK * np), w);
        const V wk = twM(np * k0);
        V* __restrict__ zo = zbuf + (int64_t)k0 * Mp + np;
        zo[0] = cmul(x[0], wk);
#pragma unroll
        for (int q = 1; q < 16; ++q) zo[(int64_t)BLK * q * Mp] = cmul(x[q], cmul(wk, w[q]));
    } else {
        const V* __restrict__ zi = zbuf + (int64_t)k0 * Mp + np;
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = zi[(int64_t)BLK * q * Mp];
        pow_chain16(cconj(twM(BLK * np)), w);
        const V wk = cconj(twM(np * k0));
        x[0] = cmul(x[0], wk);
#pragma unroll
        for (int q = 1; q < 16; ++q) x[q] = cmul(x[q], cmul(wk, w[q]));
        dft16<V, true>(x);
#pragma unroll
        for (int q = 0; q < 16; ++q) buf[(17 * r + q) * TW + c] = x[q];
        __syncthreads();
        {
            V wm[R];
            wm[1] = cconj(twN1[16 * j]);
#pragma unroll
            for (int m = 2; m < R; ++m) wm[m] = (m & 1) ? cmul(wm[m - 1], wm[1]) : cmul(wm[m / 2], wm[m / 2]);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int p = (PBS * (NB * gw + i) + j) * TW + c;
                V v[R];
                v[0] = buf[p];
#pragma unroll
                for (int m = 1; m < R; ++m) v[m] = cmul(buf[p + 17 * m * TW], wm[m]);
                dftR<V, true, R>(v);
#pragma unroll
                for (int m = 0; m < R; ++m) buf[p + 17 * m * TW] = v[m];
            }
        }
        __syncthreads();
        pow_chain16(cconj(twN1[r]), w);
        const int p0 = (r + (r >> 4)) * TW + c;
        x[0] = buf[p0];
#pragma unroll
        for (int q = 1; q < 16; ++q) x[q] = cmul(buf[p0 + PBS * q * TW], w[q]);
        dft16<V, true>(x);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int64_t e = 2 * ((int64_t)(r + BLK * q) * Mp + np);       // natural order: y[e], y[e + 1]
            if (e < n) {
                T ov = alpha * x[q].x;
                if (beta != (T)0) ov = fma_t(beta, y[e], ov);
                y[e] = ov;
            }
            if (e + 1 < n) {
                T ov = alpha * x[q].y;
                if (beta != (T)0) ov = fma_t(beta, y[e + 1], ov);
                y[e + 1] = ov;
            }
        }
    }
}

__device__ __forceinline__ int rev16_3(int x) { return ((x & 0xF) << 8) | (x & 0xF0) | ((x >> 8) & 0xF); }

// S16[k1][rev16(k')] = S[k1][k'] for M' = 4096 (rowfft16_fused_kernel reads the spectrum in the slot order of its LDS rows)
// REALS (SymmetricToeplitz: the embedding is even, its spectrum real): the copy keeps the real parts only, M scalars
template <typename T, bool REALS>
__global__ void permute16_kernel(const typename V2T<T>::type* __restrict__ S, void* __restrict__ S16, int64_t tot) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tot) return;
    const int p = (int)(i & 4095);
    const auto v = S[(i & ~(int64_t)4095) | (((p & 0xF) << 8) | (p & 0xF0) | ((p >> 8) & 0xF))];
    if constexpr (REALS) ((T*)S16)[i] = v.x;
    else ((typename V2T<T>::type*)S16)[i] = v;
}


// REALS: the digit-reversed spectrum copy of a symmetric matrix holds real parts only (C5 fp64 89.2 -> 86.7 us).
// (Measured and not kept: fetching the thread's 16 spectrum values into registers at kernel start (fp64, 206 VGPRs): 96.1 vs 89.7 us.)
// (Measured and not kept, round 3 — VERDICT r2 item 8: ONE row in LDS (68 KB) and the other row's spectrum in registers, 256-thread workgroups,
//  two per CU in different phases, twiddles from the global table: C5 fp64 86.9 us against 86.1 for this kernel, fp32 62.8 / 60.5 — the row pass is
//  not short of overlap between a CU's load, compute and store phases; profiles/r03_c5_one_row_lds_ab.txt.)
// PERSIST (round 4): a workgroup walks the row pairs wg, wg + gridDim.x, ... and fetches the NEXT pair's first-stage inputs into a second
// register set before it starts on the current pair.  With one pair per workgroup (and one or two workgroups per CU, all started together)
// every CU loads, computes and stores in lockstep with every other: HBM idles through the twelve LDS stages and the CUs idle through the
// transfers — the row pass moved 201 MB at 4.1 TB/s.  Now the next pair's 128 KB are in flight under the current pair's stages, and the
// current pair's stores (issued from registers by the last inverse stage) drain under the next pair's.
template <typename T, bool REALS, bool PERSIST = false>
__global__ __launch_bounds__(512, (sizeof(T) == 4 && !PERSIST ? 4 : 2)) void rowfft16_fused_kernel(typename V2T<T>::type* __restrict__ zbuf,
                                                                          const typename V2T<T>::type* __restrict__ S,
                                                                          const typename V2T<T>::type* __restrict__ tA,
                                                                          const typename V2T<T>::type* __restrict__ tB,
                                                                          const typename V2T<T>::type* __restrict__ twr,
                                                                          const void* __restrict__ S16v,
                                                                          const typename V2T<T>::type* __restrict__ tB16, int n1) {
    using V = typename V2T<T>::type;
    constexpr int Mp = 4096, Q = Mp / 4, NT = 512;
    constexpr int MpP = Mp + Mp / 16;
    auto P = [](int i) { return i + (i >> 4); };
    __shared__ V rowA[MpP];
    __shared__ V rowB[MpP];
    __shared__ V stw[Q];
    const int tid0 = threadIdx.x;
    const int npairs = n1 / 2;
    // row of this thread for pair wg: A = row wg (0 for wg = 0), B = its conjugate partner n1 - wg (n1 / 2 for wg = 0)
    auto row_of = [&](int wg) { return (tid0 >> 8) ? ((wg == 0) ? n1 / 2 : n1 - wg) : wg; };
#pragma unroll
    for (int i = 0; i < Q / NT; ++i) stw[tid0 + i * NT] = twr[tid0 + i * NT];
    [[maybe_unused]] V xn[16];
    if constexpr (PERSIST) {
        const V* __restrict__ g0 = zbuf + (int64_t)row_of(blockIdx.x) * Mp;
#pragma unroll
        for (int q = 0; q < 16; ++q) xn[q] = g0[(tid0 & 255) + 256 * q];
    }
    for (int wg = blockIdx.x; wg < npairs; wg += gridDim.x) {
    // the thread index is made opaque per pair: everything below depends on it alone, and the compiler otherwise hoists the address
    // arithmetic of all six stages out of the pair loop (256 VGPRs and scratch where the one-pair kernel needs 142)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    V* __restrict__ xr = (tid >> 8) ? rowB : rowA;          // this thread's row
    const int b = tid & 255;
    const int kA = (wg == 0) ? 0 : wg, kB = (wg == 0) ? n1 / 2 : n1 - wg;
    V* __restrict__ gr = zbuf + (int64_t)row_of(wg) * Mp;
    // the first forward stage (span 256) takes its 16 inputs b + 256 q straight from global memory — coalesced along b — and
    // the last inverse stage writes its outputs the same way: no separate load / store pass through LDS
    V x0[16];
    if constexpr (PERSIST) {
#pragma unroll
        for (int q = 0; q < 16; ++q) x0[q] = xn[q];
        const int nwg = wg + (int)gridDim.x;
        if (nwg < npairs) {
            const V* __restrict__ gn = zbuf + (int64_t)row_of(nwg) * Mp;
#pragma unroll
            for (int q = 0; q < 16; ++q) xn[q] = gn[b + 256 * q];
        }
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) x0[q] = gr[b + 256 * q];
    }
    __syncthreads();                                        // the table (first pair) / the previous pair's last stage has left the rows
    // twiddles of a butterfly: w1 = W^e from the table, W^(e q) by repeated multiplication (15 roundings at most)
    {
        dft16<V, false>(x0);
        const V w1 = full_tw<V, false>(stw, b, Q);
        V w = w1;
        xr[P(b)] = x0[0];
#pragma unroll
        for (int q = 1; q < 16; ++q) { xr[P(b + 256 * q)] = cmul(x0[q], w); w = cmul(w, w1); }
        __syncthreads();
    }

    // ---- forward: radix-16 decimation in frequency, remaining spans 16, 1 ------------------------------------------------
#pragma unroll 1
    for (int s = 1; s < 3; ++s) {
        const int lg = 8 - 4 * s, Ls = 1 << lg;
        const int j = b & (Ls - 1), g = b >> lg;
        const int i0 = g * 16 * Ls + j;
        V x[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = xr[P(i0 + q * Ls)];
        dft16<V, false>(x);
        xr[P(i0)] = x[0];
        if (s == 2) {
#pragma unroll
            for (int q = 1; q < 16; ++q) xr[P(i0 + q)] = x[q];
        } else {
            const V w1 = full_tw<V, false>(stw, j << 4, Q);
            V w = w1;
#pragma unroll
            for (int q = 1; q < 16; ++q) { xr[P(i0 + q * Ls)] = cmul(x[q], w); w = cmul(w, w1); }
        }
        __syncthreads();
    }

    // ---- spectral step on conjugate pairs (same algebra as spectral_kernel; frequency k' lives at slot rev16(k')) --------
    // S16 / tB16 = the rows of S and the table tB with their entries at the digit-reversed index (built with the handle): thread
    // t then owns the LDS slots t + 512 i — consecutive lanes on consecutive slots, no bank conflicts — and still reads the
    // spectrum contiguously.  (With natural-order S the lanes sat 256 slots apart: 8-way conflicts, more LDS cycles than all six stages.)
    auto pair = [&](V& zs, V& zps, bool self, int k1, int kp, int64_t pos, int64_t ppos, const V* sp = nullptr, const V* spp = nullptr, const V* wp = nullptr) {
        const V Z = zs;
        if (pos == 0) {
            const T X0 = Z.x + Z.y, XM = Z.x - Z.y;
            const T Y0 = X0 * S[0].x, YM = XM * S[(int64_t)n1 * Mp].x;
            zs = V{(T)0.5 * (Y0 + YM), (T)0.5 * (Y0 - YM)};
            return;
        }
        const V Zp = zps;
        const V w = cmul(tA[k1], wp ? *wp : tB[kp]);
        const V wc = cconj(w);
        const V e = cadd(Z, cconj(Zp)), o = csub(Z, cconj(Zp));
        const V wo = cmul(w, o);
        const V X{(T)0.5 * (e.x + wo.y), (T)0.5 * (e.y - wo.x)};
        const V ep = cconj(e), op = V{-o.x, o.y};
        const V wop = cmul(wc, op);
        const V Xp{(T)0.5 * (ep.x - wop.y), (T)0.5 * (ep.y + wop.x)};
        V Y, Yp;
        if (REALS && sp) { Y = V{X.x * sp->x, X.y * sp->x}; Yp = V{Xp.x * spp->x, Xp.y * spp->x}; }      // real spectrum: two products each
        else { Y = cmul(X, sp ? *sp : S[pos]); Yp = cmul(Xp, spp ? *spp : S[ppos]); }
        const V f = cadd(Y, cconj(Yp)), h = csub(Y, cconj(Yp));
        const V wh = cmul(wc, h);
        zs = V{(T)0.5 * (f.x - wh.y), (T)0.5 * (f.y + wh.x)};
        if (!self) {
            const V fp = cconj(f), hp = V{-h.x, h.y};
            const V whp = cmul(w, hp);
            zps = V{(T)0.5 * (fp.x + whp.y), (T)0.5 * (fp.y - whp.x)};
        }
    };
    if (wg != 0) {
#pragma unroll
        for (int i = 0; i < Mp / NT; ++i) {
            const int p = tid + i * NT;                    // slot p holds frequency k' = rev16(p); its partner M' - 1 - k' sits at slot M' - 1 - p
            const V wt = tB16[p];
            V s1, s2;
            if constexpr (REALS) {
                s1 = V{((const T*)S16v)[(int64_t)kA * Mp + p], (T)0}; s2 = V{((const T*)S16v)[(int64_t)kB * Mp + (Mp - 1 - p)], (T)0};
            } else {
                s1 = ((const V*)S16v)[(int64_t)kA * Mp + p]; s2 = ((const V*)S16v)[(int64_t)kB * Mp + (Mp - 1 - p)];
            }
            pair(rowA[P(p)], rowB[P(Mp - 1 - p)], false, kA, 0, 1, 1, &s1, &s2, &wt);
        }
    } else {
        for (int i = 0; i < Mp / NT; ++i) {
            const int kp = tid + i * NT;
            const int kq = (Mp - kp) & (Mp - 1);
            if (kp <= kq) pair(rowA[P(rev16_3(kp))], rowA[P(rev16_3(kq))], kp == kq, 0, kp, (int64_t)kp, (int64_t)kq);
            const int kr = Mp - 1 - kp;
            if (kp < kr) {
                const int p = rev16_3(kp);
                pair(rowB[P(p)], rowB[P(Mp - 1 - p)], false, n1 / 2, kp, (int64_t)(n1 / 2) * Mp + kp, (int64_t)(n1 / 2) * Mp + kr);
            }
        }
    }
    __syncthreads();

    // ---- inverse: radix-16 decimation in time, conjugated twiddles, spans 1, 16, 256 -------------------------------------
#pragma unroll 1
    for (int s = 0; s < 3; ++s) {
        const int lg = 4 * s, Ls = 1 << lg;
        const int j = b & (Ls - 1), g = b >> lg;
        const int i0 = g * 16 * Ls + j;
        V x[16];
        x[0] = xr[P(i0)];
        if (s == 0) {
#pragma unroll
            for (int q = 1; q < 16; ++q) x[q] = xr[P(i0 + q)];
        } else {
            const V w1 = full_tw<V, true>(stw, j << (4 * (2 - s)), Q);     // conj W^(j M' / (16 Ls))
            V w = w1;
#pragma unroll
            for (int q = 1; q < 16; ++q) { x[q] = cmul(w, xr[P(i0 + q * Ls)]); w = cmul(w, w1); }
        }
        dft16<V, true>(x);
        if (s == 2) {
#pragma unroll
            for (int q = 0; q < 16; ++q) gr[b + 256 * q] = x[q];            // natural order, coalesced along b
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) xr[P(i0 + q * Ls)] = x[q];
            __syncthreads();
        }
    }
    }   // row pairs of this workgroup
}

}  // namespace covgram

using namespace covgram;

struct covgram_toeplitz {
    covgram_ctx* ctx = nullptr;
    int64_t n = 0, m = 0, N = 0;
    int32_t dtype = 0;
    bool circulant = false;
    rocfft_plan fwd = nullptr, inv = nullptr;
    rocfft_execution_info info = nullptr;
    void* work = nullptr; size_t work_bytes = 0;
    void* spec = nullptr;   // (N/2+1) complex, cached, pre-divided by N
    void* rbuf = nullptr;   // N reals
    void* cbuf = nullptr;   // (N/2+1) complex
    void* stage_a = nullptr; void* stage_y = nullptr;  // device staging for loc == HOST
    // fast path (four-step FFT without transposes): M = N/2 = 1024 * Mp
    bool fast = false;
    int64_t Mp = 0;
    rocfft_plan bfwd = nullptr, binv = nullptr;   // 1024 contiguous length-Mp complex transforms, in place
    void* zbuf = nullptr;    // M complex
    void* sperm = nullptr;   // M + 1 complex: permuted half spectrum of the embedding / M, Nyquist bin last
    int n1 = 1024;             // column length of the four-step split: M = n1 x Mp (512, 1024 or 2048 — whichever makes Mp a power of four <= 4096)
    bool sperm16_real = false; // sperm16 holds real parts only (symmetric matrix)
    void* sperm16 = nullptr; // M' = 4096: the same rows with their entries at the base-16 digit-reversed index (rowfft16_fused_kernel)
    void* tables = nullptr;  // tw1024 | tlo(2048) | thi(M/2048) | tA(1024) | tB(Mp)   (complex)
};


namespace covgram {

// Columns per tile.  64-byte row segments (half a cache line) so that a tile + the stage twiddles take 80 KiB of LDS and TWO
// workgroups share a CU: one tile's HBM phases overlap the other's LDS butterfly stages.  The two tiles that split each
// 128-byte line are mapped to workgroups b and b+8, which the dispatcher places on the same XCD (same L2) back to back.
template <typename T> constexpr int colfft_tw() { return sizeof(T) == 8 ? 4 : 8; }

// complex tables for the fast path, computed in long double on the host
template <typename T>
static void fill_tables(std::vector<T>& h, int64_t M, int64_t Mp, int n1) {
    const long double PI = 3.14159265358979323846264338327950288L;
    const int64_t nhi = M >> TWID_LB;
    h.resize(2 * (size_t)(n1 + (1 << TWID_LB) + nhi + n1 + Mp + Mp / 4 + Mp));
    size_t o = 0;
    auto put = [&](long double ang) { h[o++] = (T)cosl(ang); h[o++] = (T)sinl(ang); };
    for (int t = 0; t < n1; ++t) put(-2 * PI * t / n1);                               // tw[t]     = W_n1^t (n1 = column length)
    for (int l = 0; l < (1 << TWID_LB); ++l) put(-2 * PI * l / (long double)M);        // tlo[l]    = W_M^l
    for (int64_t q = 0; q < nhi; ++q) put(-2 * PI * q / (long double)nhi);             // thi[q]    = W_M^(2048 q)
    for (int k1 = 0; k1 < n1; ++k1) put(-PI * k1 / (long double)M);                    // tA[k1]    = exp(-i pi k1 / M)
    for (int64_t kp = 0; kp < Mp; ++kp) put(-PI * kp / (long double)Mp);               // tB[k']    = exp(-i pi 1024 k' / M)
    for (int64_t r = 0; r < Mp / 4; ++r) put(-2 * PI * r / (long double)Mp);           // twr[r]    = W_M'^r (quarter table)
    for (int64_t p = 0; p < Mp; ++p) {                                                 // tB16[p]   = tB[rev16(p)] (M' = 4096 only)
        const int64_t kp = (Mp == 4096) ? (((p & 0xF) << 8) | (p & 0xF0) | ((p >> 8) & 0xF)) : p;
        put(-PI * kp / (long double)Mp);
    }
}

struct FastTables { const void *tw, *tlo, *thi, *tA, *tB, *twr, *tB16; };
static FastTables table_ptrs(const covgram_toeplitz* Tz) {
    const size_t cs = 2 * dtype_size(Tz->dtype);
    const char* b = (const char*)Tz->tables;
    const int64_t M = Tz->N / 2, nhi = M >> TWID_LB;
    FastTables t;
    t.tw = b; b += cs * Tz->n1;
    t.tlo = b; b += cs * (1 << TWID_LB);
    t.thi = b; b += cs * nhi;
    t.tA = b; b += cs * Tz->n1;
    t.tB = b; b += cs * Tz->Mp;
    t.twr = b; b += cs * (Tz->Mp / 4);
    t.tB16 = b;
    return t;
}

// the column FFT of the fast path, either direction (option toeplitz_colfft: 16 = colfft16_kernel, default; 4 = colfft_kernel)
template <typename T, bool INV>
static void launch_colfft(covgram_toeplitz* Tz, const FastTables& t, const T* src, int64_t len, T* y, int64_t n, T alpha, T beta) {
    using V = typename V2T<T>::type;
    constexpr int TW = colfft_tw<T>();
    hipStream_t st = Tz->ctx->stream;
    const dim3 grid((unsigned)(Tz->Mp / TW));
#define CG_COL16(RR) hipLaunchKernelGGL((colfft16_kernel<T, TW, INV, RR>), grid, dim3(16 * RR * TW), 0, st, src, len, (V*)Tz->zbuf, Tz->Mp, \
                                        (const V*)t.tw, (const V*)t.tlo, (const V*)t.thi, y, n, alpha, beta)
    if (Tz->n1 == 512) CG_COL16(2);
    else if (Tz->n1 == 2048) CG_COL16(8);
    else if (Tz->ctx->toeplitz_colfft != 4) CG_COL16(4);
    else
        hipLaunchKernelGGL((colfft_kernel<T, TW, INV>), grid, dim3(COLFFT_THREADS), 0, st, src, len, (V*)Tz->zbuf, Tz->Mp,
                           (const V*)t.tw, (const V*)t.tlo, (const V*)t.thi, y, n, alpha, beta);
#undef CG_COL16
}

// zbuf <- permuted packed spectrum of the real signal src[0..len) (zero beyond), length N
template <typename T>
static int fast_forward(covgram_toeplitz* Tz, const T* src, int64_t len) {
    const FastTables t = table_ptrs(Tz);
    launch_colfft<T, false>(Tz, t, src, len, (T*)nullptr, (int64_t)0, (T)0, (T)0);
    void* io[1] = {Tz->zbuf};
    CG_CHECK_FFT(rocfft_execute(Tz->bfwd, io, nullptr, Tz->info));
    return COVGRAM_OK;
}

template <typename T>
static int fast_mvm(covgram_toeplitz* Tz, const T* a, T* y, double alpha, double beta) {
    using V = typename V2T<T>::type;
    constexpr int TW = colfft_tw<T>();
    const FastTables t = table_ptrs(Tz);
    hipStream_t st = Tz->ctx->stream;
    int L = 0;
    for (int l = 3; l <= 6; ++l) if (Tz->Mp == ((int64_t)1 << (2 * l))) L = l;
    if (L && Tz->ctx->toeplitz_fused) {
        // column FFT -> [row FFT, spectral step, inverse row FFT] in one kernel -> inverse column FFT: three passes over zbuf
        launch_colfft<T, false>(Tz, t, a, Tz->m, (T*)nullptr, (int64_t)0, (T)0, (T)0);
        const dim3 fg(Tz->n1 / 2);
#define CG_FUSED(LL) hipLaunchKernelGGL((rowfft_fused_kernel<T, LL>), fg, dim3((1 << (2 * LL)) / 4), 0, st, (V*)Tz->zbuf, (const V*)Tz->sperm, \
                                        (const V*)t.tA, (const V*)t.tB, (const V*)t.twr, Tz->n1)
        if (L == 6 && Tz->ctx->toeplitz_fused != 2)   // M' = 4096: radix-16 stages (option toeplitz_fused = 2 keeps the radix-4 kernel: A/B)
        {
            // round 4: persistent workgroups (one per CU) with the next row pair prefetched into registers (option toeplitz_persist: -1 = fp64
            // only, 0 = never, 1 = always, k > 1 = with k workgroups).  tools/c5_persist_ab.py / c5_stagger_ab.py, profiles/r04_c5_persist_ab.txt:
            // fp64 87.6 -> 85.3 us, results bit-identical; fp32 61.2 -> 65.3 (at 177 VGPRs the kernel loses its second workgroup per CU).  The
            // pass is NOT short of load / compute / store overlap between workgroups: a late start of half the grid only adds its delay, and
            // the fp32 kernel takes as long as the fp64 one on half the bytes — a workgroup's own chain (one round trip in, six butterfly stages
            // of ~2900 DP instructions per thread at two waves per SIMD, twelve barriers, one round trip out) is what a pair costs.
            const bool persist = Tz->ctx->toeplitz_persist > 0 || (Tz->ctx->toeplitz_persist < 0 && sizeof(T) == 8);
            const dim3 pg((unsigned)std::min<int64_t>(Tz->n1 / 2, Tz->ctx->toeplitz_persist > 1 ? Tz->ctx->toeplitz_persist : Tz->ctx->num_cus));
#define CG_ROW16(REALV, PV, GRID) hipLaunchKernelGGL((rowfft16_fused_kernel<T, REALV, PV>), GRID, dim3(512), 0, st, (V*)Tz->zbuf, (const V*)Tz->sperm, (const V*)t.tA, \
                                                     (const V*)t.tB, (const V*)t.twr, (const void*)Tz->sperm16, (const V*)t.tB16, Tz->n1)
            if (Tz->sperm16_real) { if (persist) CG_ROW16(true, true, pg); else CG_ROW16(true, false, fg); }
            else { if (persist) CG_ROW16(false, true, pg); else CG_ROW16(false, false, fg); }
#undef CG_ROW16
        }
        else
        switch (L) { case 3: CG_FUSED(3); break; case 4: CG_FUSED(4); break; case 5: CG_FUSED(5); break; default: CG_FUSED(6); break; }
#undef CG_FUSED
        launch_colfft<T, true>(Tz, t, (const T*)nullptr, (int64_t)0, y, Tz->n, (T)alpha, (T)beta);
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess) { set_error("toeplitz fused path launch failed: %s", hipGetErrorString(e2)); return COVGRAM_EHIP; }
        return COVGRAM_OK;
    }
    int rc = fast_forward<T>(Tz, a, Tz->m);
    if (rc) return rc;
    const int64_t pairs = (int64_t)(Tz->n1 / 2 + 1) * Tz->Mp;
    hipLaunchKernelGGL(spectral_kernel<T>, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, (V*)Tz->zbuf, Tz->Mp, (const V*)Tz->sperm,
                       (const V*)t.tA, (const V*)t.tB, Tz->n1);
    void* io[1] = {Tz->zbuf};
    CG_CHECK_FFT(rocfft_execute(Tz->binv, io, nullptr, Tz->info));
    launch_colfft<T, true>(Tz, t, (const T*)nullptr, (int64_t)0, y, Tz->n, (T)alpha, (T)beta);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("toeplitz fast path launch failed: %s", hipGetErrorString(e)); return COVGRAM_EHIP; }
    return COVGRAM_OK;
}

}  // namespace covgram

static int64_t next_pow2(int64_t v) { int64_t p = 1; while (p < v) p <<= 1; return p; }

extern "C" {

int covgram_toeplitz_destroy(covgram_toeplitz* T) {
    if (!T) return COVGRAM_OK;
    ::covgram::DeviceGuard _cg_dev(T->ctx->device);
    (void)hipStreamSynchronize(T->ctx->stream);
    if (T->fwd) rocfft_plan_destroy(T->fwd);
    if (T->inv) rocfft_plan_destroy(T->inv);
    if (T->bfwd) rocfft_plan_destroy(T->bfwd);
    if (T->binv) rocfft_plan_destroy(T->binv);
    if (T->info) rocfft_execution_info_destroy(T->info);
    void* bufs[] = {T->work, T->spec, T->rbuf, T->cbuf, T->stage_a, T->stage_y, T->zbuf, T->sperm, T->sperm16, T->tables};
    for (void* b : bufs) if (b) (void)hipFree(b);
    T->ctx->live_handles--;
    if (--g_rocfft_users == 0) rocfft_cleanup();
    delete T;
    return COVGRAM_OK;
}

int covgram_toeplitz_create(covgram_ctx* ctx, covgram_toeplitz** out, const void* vc, const void* vr, int64_t n, int64_t m,
                            int32_t dtype, int32_t loc, int32_t circulant) {
    CG_REQUIRE(ctx && out && vc, COVGRAM_EINVAL, "NULL argument");
    CG_REQUIRE(n >= 1, COVGRAM_EINVAL, "toeplitz: n must be >= 1");
    CG_REQUIRE(dtype == COVGRAM_F32 || dtype == COVGRAM_F64, COVGRAM_EINVAL, "unknown dtype %d", dtype);
    if (!vr) m = n;
    CG_REQUIRE(m >= 1, COVGRAM_EINVAL, "toeplitz: m must be >= 1");
    CG_REQUIRE(!(circulant && vr), COVGRAM_EINVAL, "circulant takes a first column only");
    CG_DEVICE(ctx);
    const size_t ts = dtype_size(dtype);
    if (g_rocfft_users++ == 0) rocfft_setup();
    covgram_toeplitz* T = new covgram_toeplitz();
    T->ctx = ctx; T->n = n; T->m = m; T->dtype = dtype; T->circulant = circulant != 0;
    T->N = circulant ? n : next_pow2(std::max<int64_t>(n + m - 1, 2));
    ctx->live_handles++;
    const int64_t N = T->N, NC = N / 2 + 1;
    int rc = COVGRAM_OK;
    auto fail = [&](int code) { covgram_toeplitz_destroy(T); return code; };
#define TRY_HIP(e) do { hipError_t _e = (e); if (_e != hipSuccess) { set_error("%s failed: %s", #e, hipGetErrorString(_e)); return fail(COVGRAM_EHIP); } } while (0)
#define TRY_FFT(e) do { rocfft_status _s = (e); if (_s != rocfft_status_success) { set_error("%s failed: rocfft_status %d", #e, (int)_s); return fail(COVGRAM_EHIP); } } while (0)
    const int64_t Mh = N / 2;
    // column length of the four-step split: 1024, or 512 / 2048 when that makes the row length a power of four <= 4096 (the sizes
    // the fused row kernels exist for: N = 2^20, 2^22 take 512 x 1024, 512 x 4096; N = 2^24 takes 2048 x 4096)
    const int64_t twmin = (dtype == COVGRAM_F64 ? colfft_tw<double>() : colfft_tw<float>());
    auto fused_len = [](int64_t mp) { return mp == 64 || mp == 256 || mp == 1024 || mp == 4096; };
    T->n1 = 1024;
    for (int cand : {1024, 512, 2048})
        if (Mh % cand == 0 && fused_len(Mh / cand)) { T->n1 = cand; break; }
    T->fast = !T->circulant && (Mh % T->n1 == 0) && (Mh / T->n1 >= twmin);
    T->Mp = T->fast ? Mh / T->n1 : 0;
    const rocfft_precision prec = (dtype == COVGRAM_F64) ? rocfft_precision_double : rocfft_precision_single;
    TRY_HIP(hipMalloc(&T->rbuf, (size_t)N * ts));
    size_t w1 = 0, w2 = 0;
    if (T->fast) {
        TRY_HIP(hipMalloc(&T->zbuf, (size_t)Mh * 2 * ts));
        TRY_HIP(hipMalloc(&T->sperm, (size_t)(Mh + 1) * 2 * ts));
        size_t blen[1] = {(size_t)T->Mp};
        TRY_FFT(rocfft_plan_create(&T->bfwd, rocfft_placement_inplace, rocfft_transform_type_complex_forward, prec, 1, blen, (size_t)T->n1, nullptr));
        TRY_FFT(rocfft_plan_create(&T->binv, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, prec, 1, blen, (size_t)T->n1, nullptr));
        TRY_FFT(rocfft_plan_get_work_buffer_size(T->bfwd, &w1));
        TRY_FFT(rocfft_plan_get_work_buffer_size(T->binv, &w2));
        if (dtype == COVGRAM_F64) {
            std::vector<double> h; fill_tables<double>(h, Mh, T->Mp, T->n1);
            TRY_HIP(hipMalloc(&T->tables, h.size() * sizeof(double)));
            TRY_HIP(hipMemcpy(T->tables, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
        } else {
            std::vector<float> h; fill_tables<float>(h, Mh, T->Mp, T->n1);
            TRY_HIP(hipMalloc(&T->tables, h.size() * sizeof(float)));
            TRY_HIP(hipMemcpy(T->tables, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    } else {
        TRY_HIP(hipMalloc(&T->spec, (size_t)NC * 2 * ts));
        TRY_HIP(hipMalloc(&T->cbuf, (size_t)NC * 2 * ts));
        size_t len[1] = {(size_t)N};
        TRY_FFT(rocfft_plan_create(&T->fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward, prec, 1, len, 1, nullptr));
        TRY_FFT(rocfft_plan_create(&T->inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, prec, 1, len, 1, nullptr));
        TRY_FFT(rocfft_plan_get_work_buffer_size(T->fwd, &w1));
        TRY_FFT(rocfft_plan_get_work_buffer_size(T->inv, &w2));
    }
    T->work_bytes = std::max(w1, w2);
    TRY_FFT(rocfft_execution_info_create(&T->info));
    if (T->work_bytes) {
        TRY_HIP(hipMalloc(&T->work, T->work_bytes));
        TRY_FFT(rocfft_execution_info_set_work_buffer(T->info, T->work, T->work_bytes));
    }
    TRY_FFT(rocfft_execution_info_set_stream(T->info, ctx->stream));

    // first column / row to the device
    const void* dvc = vc; const void* dvr = vr;
    void *tmp_c = nullptr, *tmp_r = nullptr;
    if (loc == COVGRAM_HOST) {
        TRY_HIP(hipMalloc(&tmp_c, (size_t)n * ts));
        TRY_HIP(hipMemcpyAsync(tmp_c, vc, (size_t)n * ts, hipMemcpyHostToDevice, ctx->stream));
        dvc = tmp_c;
        if (vr) {
            TRY_HIP(hipMalloc(&tmp_r, (size_t)m * ts));
            TRY_HIP(hipMemcpyAsync(tmp_r, vr, (size_t)m * ts, hipMemcpyHostToDevice, ctx->stream));
            dvr = tmp_r;
        }
    }
    if (!dvr) dvr = dvc;   // symmetric: vr = vc
    const unsigned gN = (unsigned)((N + 255) / 256), gC = (unsigned)((NC + 255) / 256);
    const int64_t me = T->circulant ? 1 : m;   // circulant: plain copy of vc (no wrapped row part)
    if (dtype == COVGRAM_F32) hipLaunchKernelGGL(embed_kernel<float>, dim3(gN), dim3(256), 0, ctx->stream, (const float*)dvc, (const float*)dvr, n, me, N, (float*)T->rbuf);
    else hipLaunchKernelGGL(embed_kernel<double>, dim3(gN), dim3(256), 0, ctx->stream, (const double*)dvc, (const double*)dvr, n, me, N, (double*)T->rbuf);
    if (T->fast) {
        // permuted half spectrum of the embedding, pre-divided by M (the unnormalised inverse transforms return M * signal)
        const int64_t tot = (int64_t)T->n1 * T->Mp;
        const FastTables ft = table_ptrs(T);
        if (dtype == COVGRAM_F32) {
            if (fast_forward<float>(T, (const float*)T->rbuf, N)) return fail(COVGRAM_EHIP);
            hipLaunchKernelGGL(half_spectrum_kernel<float>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, (const float2*)T->zbuf, T->Mp,
                               (const float2*)ft.tA, (const float2*)ft.tB, (float2*)T->sperm, 1.0f / (float)Mh, T->n1);
        } else {
            if (fast_forward<double>(T, (const double*)T->rbuf, N)) return fail(COVGRAM_EHIP);
            hipLaunchKernelGGL(half_spectrum_kernel<double>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, (const double2*)T->zbuf, T->Mp,
                               (const double2*)ft.tA, (const double2*)ft.tB, (double2*)T->sperm, 1.0 / (double)Mh, T->n1);
        }
        if (T->Mp == 4096) {
            T->sperm16_real = (vr == nullptr) && ctx->toeplitz_real_spectrum != 0;       // c_j = c_(N-j): the spectrum has no imaginary part
            TRY_HIP(hipMalloc(&T->sperm16, (size_t)tot * (T->sperm16_real ? 1 : 2) * ts));
            const dim3 pg((unsigned)((tot + 255) / 256));
            if (dtype == COVGRAM_F32) {
                if (T->sperm16_real) hipLaunchKernelGGL((permute16_kernel<float, true>), pg, dim3(256), 0, ctx->stream, (const float2*)T->sperm, T->sperm16, tot);
                else hipLaunchKernelGGL((permute16_kernel<float, false>), pg, dim3(256), 0, ctx->stream, (const float2*)T->sperm, T->sperm16, tot);
            } else {
                if (T->sperm16_real) hipLaunchKernelGGL((permute16_kernel<double, true>), pg, dim3(256), 0, ctx->stream, (const double2*)T->sperm, T->sperm16, tot);
                else hipLaunchKernelGGL((permute16_kernel<double, false>), pg, dim3(256), 0, ctx->stream, (const double2*)T->sperm, T->sperm16, tot);
            }
        }
        TRY_HIP(hipStreamSynchronize(ctx->stream));
        (void)hipFree(T->rbuf); T->rbuf = nullptr;      // the fast path needs no real scratch buffer per MVM
    } else {
        void* in[1] = {T->rbuf}; void* outb[1] = {T->spec};
        TRY_FFT(rocfft_execute(T->fwd, in, outb, T->info));
        if (dtype == COVGRAM_F32) hipLaunchKernelGGL(cmul_kernel<float>, dim3(gC), dim3(256), 0, ctx->stream, (float*)T->spec, (const float*)nullptr, NC, 1.0f / (float)N);
        else hipLaunchKernelGGL(cmul_kernel<double>, dim3(gC), dim3(256), 0, ctx->stream, (double*)T->spec, (const double*)nullptr, NC, 1.0 / (double)N);
    }
    TRY_HIP(hipStreamSynchronize(ctx->stream));
    if (tmp_c) (void)hipFree(tmp_c);
    if (tmp_r) (void)hipFree(tmp_r);
#undef TRY_HIP
#undef TRY_FFT
    (void)rc;
    *out = T;
    return COVGRAM_OK;
}

int covgram_toeplitz_mvm(covgram_toeplitz* T, const void* a, void* y, double alpha, double beta, int32_t loc) {
    CG_REQUIRE(T && a && y, COVGRAM_EINVAL, "NULL argument");
    covgram_ctx* ctx = T->ctx;
    CG_DEVICE(ctx);
    const size_t ts = dtype_size(T->dtype);
    const int64_t n = T->n, m = T->m, N = T->N, NC = N / 2 + 1;
    const void* a_dev = a; void* y_dev = y;
    if (loc == COVGRAM_HOST) {
        if (!T->stage_a) CG_CHECK_HIP(hipMalloc(&T->stage_a, (size_t)m * ts));
        if (!T->stage_y) CG_CHECK_HIP(hipMalloc(&T->stage_y, (size_t)n * ts));
        CG_CHECK_HIP(hipMemcpyAsync(T->stage_a, a, (size_t)m * ts, hipMemcpyHostToDevice, ctx->stream));
        if (beta != 0.0) CG_CHECK_HIP(hipMemcpyAsync(T->stage_y, y, (size_t)n * ts, hipMemcpyHostToDevice, ctx->stream));
        a_dev = T->stage_a; y_dev = T->stage_y;
    }
    CG_CHECK_FFT(rocfft_execution_info_set_stream(T->info, ctx->stream));
    if (T->fast) {
        int rc = (T->dtype == COVGRAM_F32) ? fast_mvm<float>(T, (const float*)a_dev, (float*)y_dev, alpha, beta)
                                           : fast_mvm<double>(T, (const double*)a_dev, (double*)y_dev, alpha, beta);
        if (rc) return rc;
        if (loc == COVGRAM_HOST) {
            CG_CHECK_HIP(hipMemcpyAsync(y, y_dev, (size_t)n * ts, hipMemcpyDeviceToHost, ctx->stream));
            CG_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        }
        return COVGRAM_OK;
    }
    const unsigned gN = (unsigned)((N + 255) / 256), gC = (unsigned)((NC + 255) / 256), gn = (unsigned)((n + 255) / 256);
    if (T->dtype == COVGRAM_F32) hipLaunchKernelGGL(pad_kernel<float>, dim3(gN), dim3(256), 0, ctx->stream, (const float*)a_dev, m, N, (float*)T->rbuf);
    else hipLaunchKernelGGL(pad_kernel<double>, dim3(gN), dim3(256), 0, ctx->stream, (const double*)a_dev, m, N, (double*)T->rbuf);
    { void* in[1] = {T->rbuf}; void* ob[1] = {T->cbuf}; CG_CHECK_FFT(rocfft_execute(T->fwd, in, ob, T->info)); }
    if (T->dtype == COVGRAM_F32) hipLaunchKernelGGL(cmul_kernel<float>, dim3(gC), dim3(256), 0, ctx->stream, (float*)T->cbuf, (const float*)T->spec, NC, 1.0f);
    else hipLaunchKernelGGL(cmul_kernel<double>, dim3(gC), dim3(256), 0, ctx->stream, (double*)T->cbuf, (const double*)T->spec, NC, 1.0);
    { void* in[1] = {T->cbuf}; void* ob[1] = {T->rbuf}; CG_CHECK_FFT(rocfft_execute(T->inv, in, ob, T->info)); }
    if (T->dtype == COVGRAM_F32) hipLaunchKernelGGL(toep_epilogue_kernel<float>, dim3(gn), dim3(256), 0, ctx->stream, (const float*)T->rbuf, (float*)y_dev, n, (float)alpha, (float)beta);
    else hipLaunchKernelGGL(toep_epilogue_kernel<double>, dim3(gn), dim3(256), 0, ctx->stream, (const double*)T->rbuf, (double*)y_dev, n, alpha, beta);
    CG_CHECK_HIP(hipGetLastError());
    if (loc == COVGRAM_HOST) {
        CG_CHECK_HIP(hipMemcpyAsync(y, y_dev, (size_t)n * ts, hipMemcpyDeviceToHost, ctx->stream));
        CG_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return COVGRAM_OK;
}

}  // extern "C"
