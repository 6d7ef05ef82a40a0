// What does a VALU phase (64 v_exp_f32 + 64 v_fma_f32, the per-tile vector work of dense_mfma_eq_kernel at four row tiles per wave) cost a wave
//   (a) alone on its SIMD, (b) beside a second wave doing the same, (c) beside a partner wave that issues v_mfma_f32_32x32x16_bf16 back to back,
//   (d) beside a partner that issues 16 MFMAs and then idles as long (the ping-pong pairing), and what do the partner's MFMAs cost then?
// One workgroup per CU; waves w and w + 4 share a SIMD.  s_memtime around ITERS repetitions; cycles per repetition, median over waves.
//   hipcc --offload-arch=gfx950 -O3 tools/pp_phase_probe.hip -o tools/pp_phase_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

#define EXP16(D) { _Pragma("unroll") for (int v = 0; v < 16; ++v) asm volatile("v_exp_f32 %0, %0" : "+v"(D[v])); }
#define FMA16(A, D, w) { _Pragma("unroll") for (int v = 0; v < 16; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(A[v]) : "v"(w), "v"(D[v])); }

// role: 0 = VALU phase loop, 1 = MFMA loop (16 per repetition), 2 = idle (exits at once), 3 = 16 MFMAs then 64 exp + 64 fma (the free-running wave)
template <int role>
__device__ __forceinline__ void body(float* out, long long* stamps, int iters, float seed) {
    const int wv = threadIdx.x >> 6;
    f16v D[4], A[4], M[4];
    for (int r = 0; r < 4; ++r) for (int v = 0; v < 16; ++v) { D[r][v] = seed * 1e-3f * (v + r); A[r][v] = 0.f; M[r][v] = 0.f; }
    bf8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed * 0.01f * j); b[j] = (__bf16)(seed * 0.02f * j); }
    const float w = seed * 0.5f;
    long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (role == 0) {
        for (int it = 0; it < iters; ++it) {
            EXP16(D[0]) EXP16(D[1]) EXP16(D[2]) EXP16(D[3])
            FMA16(A[0], D[0], w) FMA16(A[1], D[1], w) FMA16(A[2], D[2], w) FMA16(A[3], D[3], w)
        }
    } else if constexpr (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 16; ++m) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(M[m & 3]) : "v"(a), "v"(b));
        }
    } else if constexpr (role == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 16; ++m) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(M[m & 3]) : "v"(a), "v"(b));
            EXP16(D[0]) EXP16(D[1]) EXP16(D[2]) EXP16(D[3])
            FMA16(A[0], D[0], w) FMA16(A[1], D[1], w) FMA16(A[2], D[2], w) FMA16(A[3], D[3], w)
        }
    } else if constexpr (role == 4) {      // interleaved by hand: per MFMA 4 exp + 4 fma
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(M[m & 3]) : "v"(a), "v"(b));
#pragma unroll
                for (int v = 0; v < 4; ++v) asm volatile("v_exp_f32 %0, %0" : "+v"(D[m >> 2][4 * (m & 3) + v]));
#pragma unroll
                for (int v = 0; v < 4; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(A[(m + 15) & 15 >> 2][4 * (((m + 15) & 15) & 3) + v]) : "v"(w), "v"(D[((m + 15) & 15) >> 2][4 * (((m + 15) & 15) & 3) + v]));
            }
        }
    } else if constexpr (role == 5) {      // the kernel's dependency pattern: MFMA chains of tile k + 1 into one buffer, exp / fma of tile k out of the other, RT = 2
#define STEP(DN0, DN1, DC0, DC1)                                                                                              \
        _Pragma("unroll") for (int m = 0; m < 8; ++m) {                                                                       \
            if (m == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(DN0) : "v"(a), "v"(b));                 \
            else if (m == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(DN1) : "v"(a), "v"(b));            \
            else if (m & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(DN1) : "v"(a), "v"(b));             \
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(DN0) : "v"(a), "v"(b));                        \
            _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                                                   \
                if (m < 4) asm volatile("v_exp_f32 %0, %0" : "+v"(DC0[4 * m + v])); else asm volatile("v_exp_f32 %0, %0" : "+v"(DC1[4 * (m - 4) + v])); }  \
            _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                                                   \
                if (m >= 1 && m < 5) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(A[0][4 * (m - 1) + v]) : "v"(w), "v"(DC0[4 * (m - 1) + v]));           \
                else if (m >= 5) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(A[1][4 * (m - 5) + v]) : "v"(w), "v"(DC1[4 * (m - 5) + v])); }              \
        }                                                                                                                     \
        _Pragma("unroll") for (int v = 0; v < 4; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(A[1][12 + v]) : "v"(w), "v"(DC1[12 + v]));
        for (int it = 0; it < iters; ++it) {
            STEP(D[2], D[3], D[0], D[1])
            STEP(D[0], D[1], D[2], D[3])
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 4; ++r) for (int v = 0; v < 16; ++v) s += D[r][v] + A[r][v] + M[r][v];
    out[blockIdx.x * 768 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 12 + wv] = t1 - t0;
}

template <int R0, int R1, int R2>
__global__ __launch_bounds__(R2 >= 0 ? 768 : 512) void probe(float* out, long long* stamps, int iters, float seed) {
    const int wv = threadIdx.x >> 6;
    if (wv < 4) body<R0>(out, stamps, iters, seed);
    else if (wv < 8) body<R1>(out, stamps, iters, seed);
    else body<(R2 >= 0 ? R2 : 2)>(out, stamps, iters, seed);
}

template <int lo, int hi, int third = -1>
static void run(const char* name) {
    const int iters = 2000, blocks = 256, nw = third >= 0 ? 12 : 8;
    float* out; hipMalloc(&out, blocks * 768 * sizeof(float));
    long long* st; hipMalloc(&st, blocks * 12 * sizeof(long long));
    hipMemset(st, 0, blocks * 12 * sizeof(long long));
    for (int k = 0; k < 2; ++k) probe<lo, hi, third><<<blocks, 64 * nw>>>(out, st, iters, 1.0f);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 12);
    hipMemcpy(h.data(), st, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<double> a, b, c;
    for (int i = 0; i < blocks; ++i) for (int w = 0; w < nw; ++w) (w < 4 ? a : w < 8 ? b : c).push_back((double)h[i * 12 + w] / iters);
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end()); std::sort(c.begin(), c.end());
    printf("%-74s waves 0-3: %7.1f cycles per repetition | waves 4-7: %7.1f", name, a[a.size() / 2], b[b.size() / 2]);
    if (third >= 0) printf(" | waves 8-11: %7.1f", c[c.size() / 2]);
    printf("\n");
    hipFree(out); hipFree(st);
}

int main() {
    run<0, 2>("VALU phase (64 exp + 64 fma) alone on the SIMD");
    run<0, 0>("VALU phase on both waves of the SIMD");
    run<1, 2>("16 MFMAs alone");
    run<1, 1>("16 MFMAs on both waves");
    run<0, 1>("VALU phase (older wave) beside a partner issuing MFMAs back to back");
    run<1, 0>("MFMAs back to back (older wave) beside a partner in VALU phases");
    run<3, 2>("free-running: 16 MFMAs then the VALU phase, one wave per SIMD");
    run<3, 3>("free-running: 16 MFMAs then the VALU phase, two waves per SIMD");
    run<3, 3, 3>("free-running: 16 MFMAs then the VALU phase, three waves per SIMD");
    run<4, 2>("hand-interleaved (MFMA, 4 exp, 4 fma) x 16, one wave per SIMD");
    run<4, 4>("hand-interleaved (MFMA, 4 exp, 4 fma) x 16, two waves per SIMD");
    run<4, 4, 4>("hand-interleaved x 16, three waves per SIMD");
    run<5, 2>("pipelined: MFMA chains of tile k + 1 between exp / fma of tile k, one wave");
    run<5, 5>("pipelined, two waves per SIMD");
    run<5, 5, 5>("pipelined, three waves per SIMD");
    return 0;
}
