// Is v_pk_fma_f32 worth two v_fma_f32 on gfx950?  Per repetition 64 independent v_fma_f32 against 32 v_pk_fma_f32 (the same 64 fmas), alone and behind 64 v_exp_f32
// (the VALU phase of the matrix-core EQ kernels), one and two waves per SIMD, and beside a partner wave issuing v_mfma_f32_32x32x16_bf16 back to back.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pkfma_probe.hip -o tools/pkfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int ROLE>
__device__ __forceinline__ void body(float* out, long long* stamps, int iters, float seed) {
    const int wv = threadIdx.x >> 6;
    f16v D[4], A[4], M[4];
    for (int r = 0; r < 4; ++r) for (int v = 0; v < 16; ++v) { D[r][v] = seed * 1e-3f * (v + r); A[r][v] = 0.f; M[r][v] = 0.f; }
    bf8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed * 0.01f * j); b[j] = (__bf16)(seed * 0.02f * j); }
    f2 w2 = {seed * 0.5f, seed * 0.25f};
    const float w = seed * 0.5f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (ROLE == 1 || ROLE == 3) {         // 64 exp first
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int v = 0; v < 16; ++v) asm volatile("v_exp_f32 %0, %0" : "+v"(D[r][v]));
        }
        if constexpr (ROLE == 0 || ROLE == 1) {         // 64 v_fma_f32
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int v = 0; v < 16; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(A[r][v]) : "v"(w), "v"(D[r][v]));
        }
        if constexpr (ROLE == 2 || ROLE == 3) {         // 32 v_pk_fma_f32 on register pairs
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int v = 0; v < 16; v += 2) {
                    f2 acc = {A[r][v], A[r][v + 1]}, dd = {D[r][v], D[r][v + 1]};
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w2), "v"(dd));
                    A[r][v] = acc[0]; A[r][v + 1] = acc[1];
                }
        }
        if constexpr (ROLE == 5) {
#pragma unroll
            for (int m = 0; m < 16; ++m) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(M[m & 3]) : "v"(a), "v"(b));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 4; ++r) for (int v = 0; v < 16; ++v) s += D[r][v] + A[r][v] + M[r][v];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 8 + wv] = t1 - t0;
}

template <int R0, int R1>
__global__ __launch_bounds__(512) void probe(float* out, long long* stamps, int iters, float seed) {
    if ((threadIdx.x >> 6) < 4) body<R0>(out, stamps, iters, seed); else body<R1>(out, stamps, iters, seed);
}

template <int lo, int hi>
static void run(const char* name) {
    const int iters = 2000, blocks = 256;
    float* out; (void)hipMalloc(&out, blocks * 512 * sizeof(float));
    long long* st; (void)hipMalloc(&st, blocks * 8 * sizeof(long long));
    (void)hipMemset(st, 0, blocks * 8 * sizeof(long long));
    for (int k = 0; k < 2; ++k) probe<lo, hi><<<blocks, 512>>>(out, st, iters, 1.0f);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), st, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<double> a, b;
    for (int i = 0; i < blocks; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? a : b).push_back((double)h[i * 8 + w] / iters);
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%-66s waves 0-3: %7.1f cycles per repetition | waves 4-7: %7.1f\n", name, a[a.size() / 2], b[b.size() / 2]);
    (void)hipFree(out); (void)hipFree(st);
}

int main() {
    run<0, 4>("64 v_fma_f32, one wave per SIMD");
    run<2, 4>("32 v_pk_fma_f32, one wave per SIMD");
    run<0, 0>("64 v_fma_f32, two waves per SIMD");
    run<2, 2>("32 v_pk_fma_f32, two waves per SIMD");
    run<1, 4>("64 v_exp_f32 + 64 v_fma_f32, one wave");
    run<3, 4>("64 v_exp_f32 + 32 v_pk_fma_f32, one wave");
    run<1, 1>("64 v_exp_f32 + 64 v_fma_f32, two waves");
    run<3, 3>("64 v_exp_f32 + 32 v_pk_fma_f32, two waves");
    run<1, 5>("64 exp + 64 fma beside a partner's back-to-back MFMAs");
    run<3, 5>("64 exp + 32 pk_fma beside a partner's back-to-back MFMAs");
    return 0;
}
