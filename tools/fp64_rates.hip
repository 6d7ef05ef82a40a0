// Issue cost of the fp64 (and helper) instructions the profile arithmetic uses, one SIMD's view: 8 waves per SIMD, 4 independent
// register chains per wave, cycles per wave-instruction at the clock the device reports.  Build: hipcc --offload-arch=gfx950 -O3
// tools/fp64_rates.hip -o tools/fp64_rates ; run on the GPU box.  Dev tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)
constexpr int ITER = 2000, UNROLL = 8;
template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    double a[4]; int e[4];
    for (int i = 0; i < 4; ++i) { a[i] = 1.0 + 1e-3 * (threadIdx.x + i); e[i] = (threadIdx.x & 1) + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 3]));
                if constexpr (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 3]));
                if constexpr (OP == 2) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[i]) : "v"(e[i]));
                if constexpr (OP == 3) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[i]));
                if constexpr (OP == 4) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(e[i]) : "v"(a[i]));
                if constexpr (OP == 5) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[i]));
                if constexpr (OP == 6) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
                if constexpr (OP == 7) asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 3]));
                if constexpr (OP == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(e[i]) : "v"(e[(i + 1) & 3]));
                if constexpr (OP == 9) asm volatile("v_lshl_add_u32 %0, %0, 20, %1" : "+v"(e[i]) : "v"(e[(i + 1) & 3]));
                if constexpr (OP == 10) asm volatile("v_cmp_class_f64 vcc, %0, %1" : : "v"(a[i]), "v"(e[i]) : "vcc");
                if constexpr (OP == 11) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[i]), "v"(a[(i + 1) & 3]) : "vcc");
                if constexpr (OP == 12) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(a[i]));
                if constexpr (OP == 13) asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(e[i]) : "v"(a[i]));
                if constexpr (OP == 14) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(e[i]));
                if constexpr (OP == 15) asm volatile("v_max_i32 %0, %0, %1" : "+v"(e[i]) : "v"(e[(i + 1) & 3]));
                if constexpr (OP == 16) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 3]));
                if constexpr (OP == 17) asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 3]));
            }
        }
    }
    double s = 0; for (int i = 0; i < 4; ++i) s += a[i] + e[i];
    if (s == 123.456) out[0] = s;
}
template <int OP>
int run(const char* name, double* out, int cus, double khz) {
    const int waves_per_simd = 8;
    dim3 grid(cus * waves_per_simd), block(256);     // 4 waves per workgroup, one per SIMD; 8 workgroups per CU
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, ITER / 4);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, ITER);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_simd = (double)waves_per_simd * ITER * UNROLL * 4;
    printf("%-22s %8.3f ms  %6.2f cycles per wave-instruction per SIMD at %.2f GHz\n", name, ms, ms * 1e-3 * khz * 1e3 / instr_per_simd, khz * 1e-6);
    return 0;
}
int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    double* out; CHECK(hipMalloc(&out, 64));
    const int cus = p.multiProcessorCount; const double khz = p.clockRate;
    run<0>("v_fma_f64", out, cus, khz); run<16>("v_add_f64", out, cus, khz); run<1>("v_mul_f64", out, cus, khz);
    run<2>("v_ldexp_f64", out, cus, khz); run<3>("v_rndne_f64", out, cus, khz); run<4>("v_cvt_i32_f64", out, cus, khz);
    run<14>("v_cvt_f64_i32", out, cus, khz); run<5>("v_rsq_f64", out, cus, khz); run<6>("v_rcp_f64", out, cus, khz);
    run<7>("v_min_f64", out, cus, khz); run<12>("v_frexp_mant_f64", out, cus, khz); run<13>("v_frexp_exp_i32_f64", out, cus, khz);
    run<10>("v_cmp_class_f64", out, cus, khz); run<11>("v_cmp_lt_f64", out, cus, khz); run<8>("v_cndmask_b32", out, cus, khz);
    run<9>("v_lshl_add_u32", out, cus, khz); run<15>("v_max_i32", out, cus, khz); run<17>("v_mov_b64", out, cus, khz);
    return 0;
}
