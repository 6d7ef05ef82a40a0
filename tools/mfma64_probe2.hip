// v_mfma_f64_16x16x4_f64 issue rate on gfx950 by the number of independent accumulators, with distinct operand registers and with one
// operand re-read from LDS per MFMA (the Kronecker kernels' inner loop shape), in shader cycles (s_memtime) and wall time:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma64_probe2.hip -o tools/mfma64_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC, int LDSB, typename T>
__global__ __launch_bounds__(256) void probe(T* out, long long* stamps, int iters, T seed) {
    using V4 = typename std::conditional<sizeof(T) == 8, d4, f4>::type;
    __shared__ T sh[32 * 144];
    for (int i = threadIdx.x; i < 32 * 144; i += blockDim.x) sh[i] = seed + i * (T)1e-6;
    __syncthreads();
    V4 acc[NACC];
    for (int k = 0; k < NACC; ++k) acc[k] = V4{0, 0, 0, 0};
    T a[8], b[8];
    for (int k = 0; k < 8; ++k) { a[k] = seed + threadIdx.x * (T)1e-3 + k; b[k] = seed * (T)0.5 + threadIdx.x * (T)1e-4 - k; }
    const int lane = threadIdx.x & 63;
    const T* bp = sh + (lane >> 4) * 144 + (lane & 15);
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int k = 0; k < NACC; ++k) {
                T bv = LDSB ? bp[(s & 7) * 4 * 144 + (k & 7) * 16] : b[(s + k) & 7];
                if constexpr (sizeof(T) == 8) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], bv, acc[k], 0, 0, 0);
                else acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bv, acc[k], 0, 0, 0);
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    T s = 0;
    for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) { long long* st = stamps + 2 * (blockIdx.x * 4 + (threadIdx.x >> 6)); st[0] = t1 - t0; st[1] = r1 - r0; }
}

template <int NACC, int LDSB, typename T>
static void run(int waves_per_simd) {
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd;
    T* out; hipMalloc(&out, (size_t)blocks * 256 * sizeof(T));
    long long* st; hipMalloc(&st, (size_t)blocks * 4 * 2 * sizeof(long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) probe<NACC, LDSB, T><<<blocks, 256>>>(out, st, iters, (T)1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0); probe<NACC, LDSB, T><<<blocks, 256>>>(out, st, iters, (T)1.0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h((size_t)blocks * 8);
    hipMemcpy(h.data(), st, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (size_t i = 0; i < h.size() / 2; ++i) { cyc.push_back((double)h[2 * i]); clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double n_mfma = (double)iters * 8 * NACC;
    const double waves = (double)blocks * 4;
    const double tf = waves * n_mfma * 2048 / (ms * 1e-3) * 1e-12;
    printf("%s NACC=%2d %s waves/SIMD=%d: %7.3f ms %6.1f TFLOP/s | %.1f shader cycles per MFMA per wave (x%d waves = %.1f per SIMD), clock %.0f MHz\n",
           sizeof(T) == 8 ? "f64" : "f32", NACC, LDSB ? "B<-LDS" : "B=reg ", waves_per_simd, ms, tf, cyc[cyc.size() / 2] / n_mfma, waves_per_simd,
           cyc[cyc.size() / 2] / n_mfma / waves_per_simd, clk[clk.size() / 2]);
    hipFree(out); hipFree(st);
}

int main() {
    for (int w : {1, 2}) {
        run<4, 0, double>(w); run<8, 0, double>(w); run<16, 0, double>(w);
        run<8, 1, double>(w); run<16, 1, double>(w);
    }
    for (int w : {1, 2}) {
        run<4, 0, float>(w); run<8, 0, float>(w); run<8, 1, float>(w); run<16, 1, float>(w);
    }
    return 0;
}
