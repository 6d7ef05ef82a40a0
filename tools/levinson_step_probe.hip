// Where does a step of the on-chip Levinson chain (csrc/toeplitz_direct.hip: levinson_reg_kernel) spend its time?  Includes the
// library source with -DLV_DIAG: thread 0 stamps the shader clock (s_memtime) at the phase boundaries of one chosen step; the
// whole launch is timed with events (the kernel is renamed on the command line: the library holds one of the same name), and s_memtime / s_memrealtime over a spin gives the clock a lone workgroup runs at.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLV_DIAG -Dlevinson_reg_kernel=levinson_reg_kernel_diag -Icovariancefunctions.jl_amd/csrc -Iinclude tools/levinson_step_probe.hip \
//         -Lcovariancefunctions.jl_amd/lib -lcovgram -Wl,-rpath,'$ORIGIN/../covariancefunctions.jl_amd/lib' -o tools/levinson_step_probe
#include "toeplitz_direct.hip"
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void clock_kernel(unsigned long long* out, int spin) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    double v = threadIdx.x;
    for (int i = 0; i < spin; ++i) v = __builtin_fma(v, 1.0000001, 1e-9);
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (unsigned long long)v; }
}

template <typename T, bool SOLVE>
static void run(int n, const char* name) {
    using namespace covgram;
    std::vector<T> r(n), b(n);
    for (int i = 0; i < n; ++i) { r[i] = (T)std::exp(-2.0 * (i + 1) / n); b[i] = (T)std::sin(0.37 * i); }
    T *rd, *bd, *xd, *yd;
    hipMalloc(&rd, n * sizeof(T)); hipMalloc(&bd, n * sizeof(T)); hipMalloc(&xd, n * sizeof(T)); hipMalloc(&yd, n * sizeof(T));
    hipMemcpy(rd, r.data(), n * sizeof(T), hipMemcpyHostToDevice); hipMemcpy(bd, b.data(), n * sizeof(T), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int steps[3] = {n / 16, n / 2, n - 2};
    for (int si = 0; si < 3; ++si) {
        const int step = steps[si];
        hipMemcpyToSymbol(HIP_SYMBOL(lv_diag_step), &step, sizeof(int));
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            levinson_reg_launch<T, SOLVE>(0, rd, bd, xd, yd, n);    // the library's choice of geometry for this size
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        unsigned long long st[16];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(lv_stamps), sizeof(st));
        std::printf("%s n=%d: launch %.3f ms = %.0f ns/step | step %d, clocks: sums %llu, reduction+barrier %llu, update %llu, slide %llu, barrier %llu, y -> LDS %llu, total %llu\n",
                    name, n, best, 1e6 * best / (n - 1), step, st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5], st[6] - st[0]);
    }
    hipFree(rd); hipFree(bd); hipFree(xd); hipFree(yd);
}

int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(clock_kernel, dim3(1), dim3(64), 0, 0, d, 2000000);
    unsigned long long h[3]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    std::printf("lone wave: %llu shader clocks in %llu ticks of the 100 MHz counter -> %.0f MHz; dependent f64 fma: %.1f clocks\n", h[0], h[1], 100.0 * h[0] / h[1], (double)h[0] / 2000000);
    run<double, true>(1024, "levinson fp64"); run<double, true>(16384, "levinson fp64"); run<double, false>(16384, "durbin fp64");
    run<float, true>(1024, "levinson fp32"); run<float, true>(16384, "levinson fp32");
    return 0;
}
