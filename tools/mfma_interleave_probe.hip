// mfma_interleave_probe.hip — does an MFMA / v_exp / v_fma stream interleaved INSIDE one wave reach the guide-priced issue
// ceiling where the compiler's "all MFMAs, then all exponentials, then all fmas" order does not?  Bare loops of the EQ kernel's
// tile-pair body (two row tiles, K2 MFMAs each, 16 exp + 16 fma per row tile), operands in registers, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/mfma_interleave_probe.hip -o tools/mfma_interleave_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union Frag { uint4 u; bf16x8 v; };

#define EXP8(D, o) "v_exp_f32 %" #D "0, %" #D "0\n\t"
// one half: K2 MFMAs into DN (operand 0, C = 0 for the first), 16 exp on DO (operands 1..16) + 16 fmac into acc (17..32) with w (last)
#define HALF_ASM_4(DN, DO, ACC, A0, A1, A2, A3, F0, F1, F2, F3, W)                                                              \
    asm volatile(                                                                                                               \
        "v_mfma_f32_32x32x16_bf16 %0, %33, %37, 0\n\t"                                                                          \
        "v_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\tv_exp_f32 %4, %4\n\t"                                      \
        "v_fmac_f32 %17, %41, %1\n\tv_fmac_f32 %18, %41, %2\n\tv_fmac_f32 %19, %41, %3\n\tv_fmac_f32 %20, %41, %4\n\t"          \
        "v_mfma_f32_32x32x16_bf16 %0, %34, %38, %0\n\t"                                                                         \
        "v_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\t"                                      \
        "v_fmac_f32 %21, %41, %5\n\tv_fmac_f32 %22, %41, %6\n\tv_fmac_f32 %23, %41, %7\n\tv_fmac_f32 %24, %41, %8\n\t"          \
        "v_mfma_f32_32x32x16_bf16 %0, %35, %39, %0\n\t"                                                                         \
        "v_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_exp_f32 %12, %12\n\t"                                \
        "v_fmac_f32 %25, %41, %9\n\tv_fmac_f32 %26, %41, %10\n\tv_fmac_f32 %27, %41, %11\n\tv_fmac_f32 %28, %41, %12\n\t"       \
        "v_mfma_f32_32x32x16_bf16 %0, %36, %40, %0\n\t"                                                                         \
        "v_exp_f32 %13, %13\n\tv_exp_f32 %14, %14\n\tv_exp_f32 %15, %15\n\tv_exp_f32 %16, %16\n\t"                              \
        "v_fmac_f32 %29, %41, %13\n\tv_fmac_f32 %30, %41, %14\n\tv_fmac_f32 %31, %41, %15\n\tv_fmac_f32 %32, %41, %16\n\t"      \
        : "=&v"(DN), "+v"(DO[0]), "+v"(DO[1]), "+v"(DO[2]), "+v"(DO[3]), "+v"(DO[4]), "+v"(DO[5]), "+v"(DO[6]), "+v"(DO[7]),    \
          "+v"(DO[8]), "+v"(DO[9]), "+v"(DO[10]), "+v"(DO[11]), "+v"(DO[12]), "+v"(DO[13]), "+v"(DO[14]), "+v"(DO[15]),         \
          "+v"(ACC[0]), "+v"(ACC[1]), "+v"(ACC[2]), "+v"(ACC[3]), "+v"(ACC[4]), "+v"(ACC[5]), "+v"(ACC[6]), "+v"(ACC[7]),       \
          "+v"(ACC[8]), "+v"(ACC[9]), "+v"(ACC[10]), "+v"(ACC[11]), "+v"(ACC[12]), "+v"(ACC[13]), "+v"(ACC[14]), "+v"(ACC[15])  \
        : "v"(A0), "v"(A1), "v"(A2), "v"(A3), "v"(F0), "v"(F1), "v"(F2), "v"(F3), "v"(W))

template <int MODE>   // 0: compiler order (the library's process()), 1: interleaved asm stream
__global__ __launch_bounds__(256) void probe(const uint4* __restrict__ in, float* __restrict__ out, int iters) {
    const int l = threadIdx.x & 63;
    Frag a[2][4], f[4];
    for (int r = 0; r < 2; ++r) for (int mm = 0; mm < 4; ++mm) a[r][mm].u = in[(r * 4 + mm) * 64 + l];
    for (int mm = 0; mm < 4; ++mm) f[mm].u = in[(8 + mm) * 64 + l];
    float acc0[16], acc1[16];
    for (int v = 0; v < 16; ++v) { acc0[v] = 0.0f; acc1[v] = 0.0f; }
    const float w = 1.0f + 1e-6f * l;
    if constexpr (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
            f32x16 D[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                D[r] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) D[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r][mm].v, f[mm].v, D[r], 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) { D[0][v] = __builtin_amdgcn_exp2f(D[0][v]); D[1][v] = __builtin_amdgcn_exp2f(D[1][v]); }
#pragma unroll
            for (int v = 0; v < 16; ++v) { acc0[v] = __builtin_fmaf(w, D[0][v], acc0[v]); acc1[v] = __builtin_fmaf(w, D[1][v], acc1[v]); }
            asm volatile("" : "+v"(f[0].u.x));    // keep the loop honest
        }
    } else {
        f32x16 P0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, P1 = P0;
        for (int it = 0; it < iters; ++it) {
            HALF_ASM_4(P0, P1, acc1, a[0][0].v, a[0][1].v, a[0][2].v, a[0][3].v, f[0].v, f[1].v, f[2].v, f[3].v, w);
            HALF_ASM_4(P1, P0, acc0, a[1][0].v, a[1][1].v, a[1][2].v, a[1][3].v, f[0].v, f[1].v, f[2].v, f[3].v, w);
        }
        for (int v = 0; v < 16; ++v) acc1[v] += P1[v];
    }
    float s = 0;
    for (int v = 0; v < 16; ++v) s += acc0[v] + acc1[v];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int iters = 20000;
    uint4* in; float* out;
    hipMalloc(&in, 12 * 64 * sizeof(uint4)); hipMalloc(&out, 1 << 24);
    std::vector<unsigned> h(12 * 64 * 4);
    for (size_t i = 0; i < h.size(); ++i) { unsigned lo = 0x3c00 + (i * 37) % 200, hi = 0x3b80 + (i * 11) % 100; h[i] = lo | (hi << 16); }   // small bf16 values ~0.01
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wg = 1; wg <= 4; ++wg) {            // workgroups of 256 threads per CU = waves per SIMD
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256 * wg), dim3(256), 0, 0, in, out, iters);
                else hipLaunchKernelGGL(probe<1>, dim3(256 * wg), dim3(256), 0, 0, in, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            // per SIMD: wg waves, each iters tile pairs of 2048 pairs
            const double pairs_per_simd = (double)wg * iters * 2048.0;
            const double cyc64 = best * 1e-3 * 2.4e9 / (pairs_per_simd / 64.0);
            printf("waves/SIMD %d  %s: %.3f ms  -> %.2f cycles per 64 pairs per SIMD at 2.4 GHz (priced: 14.0)\n", wg, mode ? "interleaved asm" : "compiler order ", best, cyc64);
        }
    }
    return 0;
}
