// What does one 32 x 32 tile of MaternP(2) cost on a SIMD, by the form of the profile arithmetic?  (round 5)
// Each wave: iters x { 2 MFMAs (v_mfma_f32_32x32x16_bf16) -> 16 profile arguments per lane -> k = (h0 + h1 r + h2 r^2) exp2(-r), r = sqrt|s| -> one weighted sum
// (SYM: a second one with per-row weights, as the symmetric kernel's column sums) }.  W waves per SIMD run the same program (blocks of 256 W threads,
// one per CU).  Reported: shader cycles per 64 entries (= per register of the tile) per SIMD, median over waves = elapsed / (iters * 16 * W).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize tools/tile_body_probe.hip -o tools/tile_body_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 sp(float x) { return (f32x2){x, x}; }

// FORM 0: entry by entry (round 4); 1: register pairs, NP pairs at a time, stage by stage; 3: EQ (one exp2) for scale
template <int FORM, int NP, bool SYM>
__device__ __forceinline__ void tile(f32x16& D, float h0, float h1, float h2, float w, const float (&u)[16], f32x2 (&acc)[8], f32x2& c01, f32x2& c23) {
    if constexpr (FORM == 0) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const float r = __builtin_amdgcn_sqrtf(fmaxf(D[v], 0.0f));
            D[v] = __builtin_fmaf(__builtin_fmaf(h2, r, h1), r, h0) * __builtin_amdgcn_exp2f(-r);
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            acc[v >> 1][v & 1] = __builtin_fmaf(w, D[v], acc[v >> 1][v & 1]);
            if constexpr (SYM) { if (v & 2) c23[v & 1] = __builtin_fmaf(u[v], D[v], c23[v & 1]); else c01[v & 1] = __builtin_fmaf(u[v], D[v], c01[v & 1]); }
        }
    } else if constexpr (FORM == 3) {
#pragma unroll
        for (int v = 0; v < 16; ++v) D[v] = __builtin_amdgcn_exp2f(D[v]);
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const f32x2 d = {D[2 * v], D[2 * v + 1]};
            acc[v] = pk_fma(sp(w), d, acc[v]);
            if constexpr (SYM) { if (v & 1) c23 = pk_fma((f32x2){u[2 * v], u[2 * v + 1]}, d, c23); else c01 = pk_fma((f32x2){u[2 * v], u[2 * v + 1]}, d, c01); }
        }
    } else {
#pragma unroll
        for (int g = 0; g < 8; g += NP) {
            f32x2 r[NP], e[NP], q[NP];
#pragma unroll
            for (int v = 0; v < NP; ++v) r[v] = (f32x2){__builtin_amdgcn_sqrtf(__builtin_fabsf(D[2 * (g + v)])), __builtin_amdgcn_sqrtf(__builtin_fabsf(D[2 * (g + v) + 1]))};
#pragma unroll
            for (int v = 0; v < NP; ++v) e[v] = (f32x2){__builtin_amdgcn_exp2f(-r[v][0]), __builtin_amdgcn_exp2f(-r[v][1])};
            if constexpr (FORM == 1) {
#pragma unroll
                for (int v = 0; v < NP; ++v) q[v] = pk_fma(sp(h2), r[v], sp(h1));
#pragma unroll
                for (int v = 0; v < NP; ++v) q[v] = pk_fma(q[v], r[v], sp(h0));
#pragma unroll
                for (int v = 0; v < NP; ++v) q[v] = q[v] * e[v];
            } else {      // FORM 2: the chain per pair (what the library's first round-5 build wrote)
#pragma unroll
                for (int v = 0; v < NP; ++v) q[v] = pk_fma(pk_fma(sp(h2), r[v], sp(h1)), r[v], sp(h0)) * e[v];
            }
#pragma unroll
            for (int v = 0; v < NP; ++v) {
                acc[g + v] = pk_fma(sp(w), q[v], acc[g + v]);
                if constexpr (SYM) { if ((g + v) & 1) c23 = pk_fma((f32x2){u[2 * (g + v)], u[2 * (g + v) + 1]}, q[v], c23); else c01 = pk_fma((f32x2){u[2 * (g + v)], u[2 * (g + v) + 1]}, q[v], c01); }
            }
        }
    }
}

template <int FORM, int NP, bool SYM, int W, int NM = 2>
__global__ __launch_bounds__(256 * W) void probe(float* out, long long* stamps, int iters, float seed, const float* hc) {
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    bf8 a0, a1, b0, b1;
    for (int j = 0; j < 8; ++j) { a0[j] = (__bf16)(seed * 0.11f * (j + 1) + 0.01f * l); b0[j] = (__bf16)(seed * 0.07f * (j + 2)); a1[j] = (__bf16)(0.05f * j); b1[j] = (__bf16)(seed * 0.03f * (l & 7)); }
    float u[16];
    for (int v = 0; v < 16; ++v) u[v] = seed * 0.01f * (v + l);
    f32x2 acc[8];
    for (int v = 0; v < 8; ++v) acc[v] = (f32x2){0.f, 0.f};
    f32x2 c01 = {0.f, 0.f}, c23 = {0.f, 0.f};
    const float h0 = hc[0], h1 = hc[1], h2 = hc[2];
    float w = seed;
    float cs = 0.0f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        f32x16 D = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < NM / 2; ++q) {
            D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, D, 0, 0, 0);
            D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, D, 0, 0, 0);
        }
        tile<FORM, NP, SYM>(D, h0, h1, h2, w, u, acc, c01, c23);
        if constexpr (SYM) { cs += (c01[0] + c01[1]) + (c23[0] + c23[1]); c01 = (f32x2){0.f, 0.f}; c23 = (f32x2){0.f, 0.f}; }
        // the next tile's operands differ (nothing hoists): rotate the B fragments by the running sum's low bits
        b0[it & 7] = (__bf16)(w * 1e-3f); w = w * 1.0001f;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = cs;
    for (int v = 0; v < 8; ++v) s += acc[v][0] + acc[v][1];
    out[blockIdx.x * 256 * W + threadIdx.x] = s;
    if (l == 0) stamps[blockIdx.x * 4 * W + wv] = t1 - t0;
}

template <int FORM, int NP, bool SYM, int W, int NM = 2>
static double run1() {
    const int iters = 4000, blocks = 256;
    float* out; (void)hipMalloc(&out, (size_t)blocks * 256 * W * sizeof(float));
    long long* st; (void)hipMalloc(&st, (size_t)blocks * 4 * W * sizeof(long long));
    float hh[3] = {1.0f, 0.693f, 0.16f};
    float* hc; (void)hipMalloc(&hc, sizeof(hh)); (void)hipMemcpy(hc, hh, sizeof(hh), hipMemcpyHostToDevice);
    for (int k = 0; k < 2; ++k) probe<FORM, NP, SYM, W, NM><<<blocks, 256 * W>>>(out, st, iters, 1.0f, hc);
    (void)hipDeviceSynchronize();
    std::vector<long long> h((size_t)blocks * 4 * W);
    (void)hipMemcpy(h.data(), st, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<double> a;
    for (auto x : h) a.push_back((double)x / iters / 16.0 / W);
    std::sort(a.begin(), a.end());
    (void)hipFree(out); (void)hipFree(st); (void)hipFree(hc);
    return a[a.size() / 2];
}
template <int FORM, int NP, bool SYM, int NM = 2>
static void run(const char* name) {
    printf("%-78s W=1 %6.2f | W=2 %6.2f | W=3 %6.2f | W=4 %6.2f   cycles per 64 entries per SIMD\n", name, run1<FORM, NP, SYM, 1, NM>(), run1<FORM, NP, SYM, 2, NM>(), run1<FORM, NP, SYM, 3, NM>(),
           run1<FORM, NP, SYM, 4, NM>());
}

int main() {
    run<3, 8, false>("EQ: 16 exp2 + 8 pk_fma (the general EQ kernel's tile body)");
    run<3, 8, false, 4>("EQ, FOUR MFMAs per tile (a dependent chain; 128 matrix-pipe cycles = 8 per 64 entries)");
    run<3, 8, false, 8>("EQ, EIGHT MFMAs per tile (256 matrix-pipe cycles = 16 per 64 entries)");
    run<3, 8, true, 8>("EQ symmetric, EIGHT MFMAs per tile");
    run<3, 8, false, 16>("EQ, SIXTEEN MFMAs per tile (512 matrix-pipe cycles = 32 per 64 entries)");
    run<3, 8, true>("EQ symmetric: 16 exp2 + 16 pk_fma");
    run<0, 8, false>("MaternP(2) entry by entry: max, sqrt, exp2, 2 fma, mul, fma (round 4)");
    run<0, 8, true>("MaternP(2) entry by entry, symmetric (+ fma)");
    run<2, 4, false>("MaternP(2) pairs, chain per pair, 4 pairs at a time");
    run<1, 2, false>("MaternP(2) pairs, stage by stage, 2 pairs at a time");
    run<1, 4, false>("MaternP(2) pairs, stage by stage, 4 pairs at a time");
    run<1, 8, false>("MaternP(2) pairs, stage by stage, all 8 pairs");
    run<2, 4, true>("MaternP(2) symmetric, pairs, chain per pair, 4 at a time");
    run<1, 4, true>("MaternP(2) symmetric, pairs, stage by stage, 4 at a time");
    run<1, 8, true>("MaternP(2) symmetric, pairs, stage by stage, all 8");
    return 0;
}
