// fp64 MFMA rate and its overlap with fp64 VALU work on gfx950: hipcc --offload-arch=gfx950 -O3 tools/mfma64_probe.hip -o tools/mfma64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: MFMA only, 1: VALU fma only, 2: both interleaved
__global__ __launch_bounds__(256) void probe(double* out, int iters, double seed) {
    d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double a = seed + threadIdx.x * 1e-3, b = seed * 0.5 + threadIdx.x * 1e-4;
    double v[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
    for (int it = 0; it < iters; ++it) {
        if (MODE != 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
        }
        if (MODE != 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = __builtin_fma(v[k], 1.0000001, 1e-9);
        }
    }
    double s = 0;
    for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    for (int k = 0; k < 8; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, int waves_per_simd) {
    const int iters = 20000;
    const int blocks = 256 * waves_per_simd;     // 256 threads = 4 waves per block -> one wave per SIMD per block and CU
    double* out; hipMalloc(&out, (size_t)blocks * 256 * sizeof(double));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(out, 100, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0); probe<MODE><<<blocks, 256>>>(out, iters, 1.0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * 4;
    const double mfma = (MODE != 1) ? waves * iters * 4 : 0, fma = (MODE != 0) ? waves * iters * 32 : 0;
    const double tf = (mfma * 2048 + fma * 128) / (ms * 1e-3) * 1e-12;
    // cycles per instruction per SIMD at 2.4 GHz: time * 2.4e9 / (instructions per SIMD)
    const double per_simd = waves / 1024.0;
    printf("%-28s waves/SIMD=%d: %8.3f ms  %6.1f TFLOP/s", name, waves_per_simd, ms, tf);
    if (mfma) printf("  %.1f cyc/MFMA", ms * 1e-3 * 2.4e9 / (per_simd * iters * 4) * (MODE == 2 ? 1 : 1));
    if (fma && !mfma) printf("  %.2f cyc/v_fma_f64", ms * 1e-3 * 2.4e9 / (per_simd * iters * 32));
    printf("\n");
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("MFMA f64 16x16x4 only", w);
        run<1>("v_fma_f64 only (32 per iter)", w);
        run<2>("MFMA + 8x v_fma_f64 per MFMA", w);
    }
    return 0;
}
