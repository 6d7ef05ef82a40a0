// Do v_mfma_f64_16x16x4_f64 and v_fma_f64 overlap on gfx950?  (DESIGN 3.2's "the pipes do not overlap" rested on round 1's
// probe, whose MFMA timing round 3's probe invalidated; the interleaved half was never repeated.)
// Per loop iteration: NM independent MFMAs (distinct operand registers, 16 accumulators round-robin) with NF independent v_fma_f64
// (8 chains, three-address, operands in registers) after EACH MFMA, in one instruction stream per wave.  Reported: shader cycles
// (s_memtime) per iteration for MFMA only (NF = 0), FMA only (NM = 0) and both; "overlap" = (t_mfma + t_fma - t_both) / min(t_mfma, t_fma):
// 1 = the shorter stream hides completely, 0 = the times add.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma64_interleave_probe.hip -o tools/mfma64_interleave_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NM, int NF>
__global__ __launch_bounds__(256) void probe(double* out, long long* stamps, int iters, double seed) {
    d4 acc[16];
    for (int k = 0; k < 16; ++k) acc[k] = d4{0, 0, 0, 0};
    double a[8], b[8], f[8], g[8];
    for (int k = 0; k < 8; ++k) {
        a[k] = seed + threadIdx.x * 1e-3 + k; b[k] = seed * 0.5 + threadIdx.x * 1e-4 - k;
        f[k] = seed * 1e-3 + k * 1e-6; g[k] = 1.0 + 1e-9 * (threadIdx.x + k);
    }
    const double c = seed * 1e-12;
    const int lane = threadIdx.x & 63;
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < (NM > 0 ? NM : 1); ++m) {
            // inline asm as well: hipcc otherwise batches the MFMAs of an iteration behind the fmas, and the question is about ONE interleaved stream
            if constexpr (NM > 0) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[m & 15]) : "v"(a[m & 7]), "v"(b[(m + 3) & 7]));
#pragma unroll
            for (int q = 0; q < NF; ++q) {
                // three-address fma on chain q & 7: f <- f * g + c   (inline asm: the compiler must not fold or reorder the stream)
                asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(f[q & 7]) : "v"(f[q & 7]), "v"(g[q & 7]), "v"(c));
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int k = 0; k < 16; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    for (int k = 0; k < 8; ++k) s += f[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) { long long* st = stamps + 2 * (blockIdx.x * 4 + (threadIdx.x >> 6)); st[0] = t1 - t0; st[1] = r1 - r0; }
}

template <int NM, int NF>
static double run(int waves_per_simd, double* ms_out) {
    const int iters = 1000;
    const int blocks = 256 * waves_per_simd;
    double* out; hipMalloc(&out, (size_t)blocks * 256 * sizeof(double));
    long long* st; hipMalloc(&st, (size_t)blocks * 4 * 2 * sizeof(long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) probe<NM, NF><<<blocks, 256>>>(out, st, iters, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0); probe<NM, NF><<<blocks, 256>>>(out, st, iters, 1.0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h((size_t)blocks * 8);
    hipMemcpy(h.data(), st, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<double> cyc;
    for (size_t i = 0; i < h.size() / 2; ++i) cyc.push_back((double)h[2 * i]);
    std::sort(cyc.begin(), cyc.end());
    hipFree(out); hipFree(st);
    *ms_out = ms;
    return cyc[cyc.size() / 2] / iters;          // shader cycles per loop iteration, median over waves
}

template <int NF>
static void trio(int w) {
    constexpr int NM = 16;
    double m0, m1, m2;
    const double tm = run<NM, 0>(w, &m0);           // 16 MFMAs
    const double tf = run<0, NF * NM>(w, &m1);      // the same number of fmas, no MFMA
    const double tb = run<NM, NF>(w, &m2);          // NF fmas after each MFMA
    const double ov = (tm + tf - tb) / std::min(tm, tf);
    const double flops_b = 256.0 * w * 4 * 1000 * (NM * 2048.0 + NM * NF * 128.0);
    printf("waves/SIMD=%d  %2d fma per MFMA: MFMA only %7.1f cyc/iter (%5.1f per MFMA) | fma only %7.1f (%4.2f per fma) | interleaved %7.1f | overlap %5.2f | "
           "combined %5.1f TFLOP/s (MFMA only %5.1f, fma only %5.1f)\n", w, NF, tm, tm / NM, tf, tf / (NF * NM), tb, ov, flops_b / (m2 * 1e-3) * 1e-12,
           256.0 * w * 4 * 1000 * NM * 2048.0 / (m0 * 1e-3) * 1e-12, 256.0 * w * 4 * 1000 * NM * NF * 128.0 / (m1 * 1e-3) * 1e-12);
}

int main() {
    for (int w : {1, 2, 3}) {
        trio<4>(w); trio<8>(w); trio<16>(w); trio<32>(w);
    }
    return 0;
}
