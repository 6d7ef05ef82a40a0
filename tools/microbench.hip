// microbench.hip — per-instruction VALU issue rates on gfx950, the constants the dense-MVM roofline is priced with.
// Usage: ./microbench   (prints lane-ops/s for v_fma_f32, v_pk_fma_f32, v_exp_f32, v_fma_f64, v_sub+v_fma mix, and the
// dense EQ pair body) at 1, 2, 4, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a[8];
    double da[4];
    float x = seed + threadIdx.x * 1e-7f;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = x + i;
#pragma unroll
    for (int i = 0; i < 4; ++i) da[i] = x + i;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = (v2f){a[i], a[i] + 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (OP == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(x));
            } else if (OP == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
            } else if (OP == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            } else if (OP == 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(da[i]) : "v"(da[(i + 1) & 3]));
            } else if (OP == 4) {   // dense EQ pair body: 3 sub, 1 mul, 2 fma, 1 exp, 1 fma  (8 VALU, 1 of them transcendental)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float d0, d1, d2, s, e;
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d0) : "v"(x), "v"(a[i]));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d1) : "v"(a[(i + 1) & 7]), "v"(x));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d2) : "v"(a[(i + 2) & 7]), "v"(x));
                    asm volatile("v_mul_f32 %0, %1, %1" : "=v"(s) : "v"(d0));
                    asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(s) : "v"(d1));
                    asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(s) : "v"(d2));
                    asm volatile("v_exp_f32 %0, -%1" : "=v"(e) : "v"(s));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(e), "v"(x));
                }
            } else if (OP == 5) {   // 7 plain VALU only (the pair body without the exp)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float d0, d1, d2, s;
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d0) : "v"(x), "v"(a[i]));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d1) : "v"(a[(i + 1) & 7]), "v"(x));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d2) : "v"(a[(i + 2) & 7]), "v"(x));
                    asm volatile("v_mul_f32 %0, %1, %1" : "=v"(s) : "v"(d0));
                    asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(s) : "v"(d1));
                    asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(s) : "v"(d2));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(s), "v"(x));
                }
            } else if (OP == 8) {   // packed pair body: two pairs per lane with v_pk_* (7 packed VALU + 2 v_exp per 2 pairs)
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    v2f d0, d1, d2, s, e;
                    v2f xx = (v2f){x, x};
                    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d0) : "v"(xx), "v"(p[i]));
                    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d1) : "v"(p[(i + 1) & 7]), "v"(xx));
                    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d2) : "v"(p[(i + 2) & 7]), "v"(xx));
                    asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(s) : "v"(d0));
                    asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(s) : "v"(d1));
                    asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(s) : "v"(d2));
                    asm volatile("v_exp_f32 %0, -%1" : "=v"(e.x) : "v"(s.x));
                    asm volatile("v_exp_f32 %0, -%1" : "=v"(e.y) : "v"(s.y));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(e), "v"(xx));
                }
            } else if (OP == 9) {   // scalar pair body, software-pipelined: all distances, then all exps, then all accumulates
                float sv[8], ev[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float d0, d1, d2;
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d0) : "v"(x), "v"(a[i]));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d1) : "v"(a[(i + 1) & 7]), "v"(x));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d2) : "v"(a[(i + 2) & 7]), "v"(x));
                    asm volatile("v_mul_f32 %0, %1, %1" : "=v"(sv[i]) : "v"(d0));
                    asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(sv[i]) : "v"(d1));
                    asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(sv[i]) : "v"(d2));
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, -%1" : "=v"(ev[i]) : "v"(sv[i]));
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(ev[i]), "v"(x));
            } else if (OP == 10) {  // alternate 1 exp : 1 fma (independent registers)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                    asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i + 4]) : "v"(x));
                }
            } else if (OP == 11) {  // 1 exp : 3 fma
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                    asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i + 2]) : "v"(x));
                    asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i + 4]) : "v"(x));
                    asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i + 6]) : "v"(x));
                }
            } else if (OP == 12) {  // 1 exp : 7 fma (all independent)
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[0]));
#pragma unroll
                for (int i = 1; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(x));
            } else if (OP == 13) {  // 1 exp : 7 pk_fma-halves (4 exps + 14 pk_fma per 8-slot group ~ same ratio)
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[0]));
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[1]));
#pragma unroll
                for (int i = 0; i < 7; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[7]));
            } else if (OP == 6) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(da[i]) : "v"(da[(i + 1) & 3]));
            } else if (OP == 7) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (float)da[i];
    if (s == 12345.678f) out[0] = s;
}

template <int OP>
int run(const char* name, double ops_per_inner, float* dout, int cus) {
    const int iters = 4096;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = cus * wps;   // 256 threads = 4 waves = 1 wave per SIMD per block
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, dout, iters, 1.0f);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, dout, iters, 1.0f);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double lane_ops = (double)blocks * 256 * iters * 4 * ops_per_inner;
        const double rate = lane_ops / (best * 1e-3);
        // cycles per wave64-instruction per SIMD at 2.4 GHz:  64 lanes / (rate / (cus*4)) * 2.4e9
        const double cyc = 64.0 / (rate / (cus * 4.0)) * 2.4e9;
        printf("%-28s waves/SIMD=%d  %8.3f ms  %9.3f T lane-ops/s  (%.2f cyc per wave-instr per SIMD @2.4GHz)\n", name, wps, best, rate * 1e-12, cyc);
    }
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    float* dout; CHECK(hipMalloc(&dout, 4096));
    const int cus = prop.multiProcessorCount;
    run<0>("v_fma_f32", 8, dout, cus);
    run<1>("v_pk_fma_f32 (2 fma/lane)", 16, dout, cus);
    run<2>("v_exp_f32", 8, dout, cus);
    run<7>("v_sqrt_f32", 8, dout, cus);
    run<3>("v_fma_f64", 4, dout, cus);
    run<6>("v_add_f64", 4, dout, cus);
    run<5>("pair body w/o exp (7 VALU)", 8 * 7, dout, cus);
    run<4>("EQ pair body (7 VALU+exp)", 8 * 8, dout, cus);
    run<9>("EQ pair body, grouped", 8 * 8, dout, cus);
    run<8>("EQ pair body, packed", 8 * 8, dout, cus);
    run<10>("1 exp : 1 fma", 8, dout, cus);
    run<11>("1 exp : 3 fma", 8, dout, cus);
    run<12>("1 exp : 7 fma", 8, dout, cus);
    run<13>("2 exp : 7 pk_fma (16 lane-ops)", 16, dout, cus);
    return 0;
}
