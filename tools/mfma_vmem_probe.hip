// Do global loads overlap with f64 / f32 MFMAs of the same SIMD?  One 512-thread workgroup per CU (2 waves per SIMD), each step
// = 32 MFMAs per wave (operands in registers) + 64 KB per CU of streamed global loads that are consumed one step later, either
// into VGPRs (global_load_dwordx4) or straight into LDS (global_load_lds_dwordx4).  Prints the time of: MFMAs alone, loads alone, both.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_vmem_probe.hip -o tools/mfma_vmem_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// MODE bit 1: MFMAs, bit 2: loads; DMA: loads go to LDS directly
template <int MODE, int DMA, typename T>
__global__ __launch_bounds__(512) void probe(const float4* __restrict__ src, size_t span_vec, T* out, int steps) {
    using V4 = typename std::conditional<sizeof(T) == 8, d4, f4>::type;
    __shared__ float4 buf[2][4096];      // 2 x 64 KB
    V4 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = V4{0, 0, 0, 0};
    T a[4], b[4];
    for (int k = 0; k < 4; ++k) { a[k] = (T)(1.0 + threadIdx.x * 1e-3 + k); b[k] = (T)(0.5 + threadIdx.x * 1e-4 - k); }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // each step the workgroup streams 64 KB = 4096 float4: 8 per thread
    const float4* base = src + ((size_t)blockIdx.x * 4096 * 64) % span_vec;
    float4 r[8], q[8];
    float4 sum = {0, 0, 0, 0};
    if (MODE & 2) {
        if (DMA) {
#pragma unroll
            for (int i = 0; i < 8; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(base + (i * 8 + wave) * 64 + lane), (lptr_t)&buf[0][(i * 8 + wave) * 64], 16, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = base[i * 512 + tid];
        }
    }
    for (int st = 0; st < steps; ++st) {
        const float4* nxt = base + (size_t)((st + 1) & 63) * 4096;
        if (MODE & 2) {
            if (DMA) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 8; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(nxt + (i * 8 + wave) * 64 + lane), (lptr_t)&buf[(st + 1) & 1][(i * 8 + wave) * 64], 16, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) q[i] = nxt[i * 512 + tid];
            }
        }
        if (MODE & 1) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if constexpr (sizeof(T) == 8) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[(s + k) & 3], acc[k], 0, 0, 0);
                    else acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[(s + k) & 3], acc[k], 0, 0, 0);
                }
        }
        if (MODE & 2) {
            if (DMA) {
                const float4 v = buf[st & 1][tid];
                sum.x += v.x;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) { sum.x += r[i].x; r[i] = q[i]; }
            }
        }
    }
    T s = (T)sum.x;
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * 512 + tid] = s;
}

template <int MODE, int DMA, typename T>
static float run(const float4* src, size_t span, T* out, int steps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) probe<MODE, DMA, T><<<256, 512>>>(src, span, out, steps);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); probe<MODE, DMA, T><<<256, 512>>>(src, span, out, steps); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

template <typename T>
static void suite(const char* name, const float4* src, size_t span, int steps) {
    T* out; (void)hipMalloc(&out, 256 * 512 * sizeof(T));
    const float m = run<1, 0, T>(src, span, out, steps);
    const float lv = run<2, 0, T>(src, span, out, steps), bv = run<3, 0, T>(src, span, out, steps);
    const float ld = run<2, 1, T>(src, span, out, steps), bd = run<3, 1, T>(src, span, out, steps);
    const double gb = 256.0 * 65536 * steps;
    printf("%s, %d steps of 32 MFMAs per wave + 64 KB per CU (footprint %zu MB): MFMAs alone %.1f us | to VGPRs: loads alone %.1f us (%.2f TB/s), both %.1f us | "
           "LDS-DMA: loads alone %.1f us (%.2f TB/s), both %.1f us\n", name, steps, span * 16 >> 20, m, lv, gb / lv * 1e-6, bv, ld, gb / ld * 1e-6, bd);
    (void)hipFree(out);
}

int main() {
    for (size_t mb : {16, 1024}) {
        const size_t bytes = mb << 20;
        float4* src; (void)hipMalloc(&src, bytes + (1 << 24)); (void)hipMemset(src, 0, bytes + (1 << 24));
        suite<double>("f64 16x16x4", src, bytes / 16, 64);
        suite<float>("f32 16x16x4", src, bytes / 16, 64);
        (void)hipFree(src);
    }
    return 0;
}
