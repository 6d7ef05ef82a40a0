// v_fmac_f64 with a DPP row_newbcast source on gfx950: semantics and issue rate against the plain three-address v_fma_f64.
// (C4: can the gradient kernel take its column records from VGPRs — loaded by counted vector loads, 16 doubles per register pair and
// row of 16 lanes — instead of SGPRs, whose scalar loads return out of order and allow one chunk in flight?)
//   hipcc --offload-arch=gfx950 -O3 tools/dpp64_probe.hip -o tools/dpp64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

// (the VOP3 v_fma_f64 has no DPP encoding on gfx950; the VOP2 accumulate form v_fmac_f64 has — DP-ALU DPP, row_newbcast only)
#define FMA_DPP(acc, y, x, K) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(y), "v"(x))
#define FMA_PLAIN(acc, y, x) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc) : "v"(y), "v"(x))

__global__ void semantics(double* out) {
    double y = (double)threadIdx.x, x = 1.0, acc = 0.0;
    FMA_DPP(acc, y, x, 5);
    out[threadIdx.x] = acc;                       // expect 16 * (lane / 16) + 5
}

template <int DPP>
__global__ __launch_bounds__(256) void rate(double* out, long long* stamps, int iters) {
    double acc[16], y[4], x[8];
    for (int k = 0; k < 16; ++k) acc[k] = 0.0;
    for (int k = 0; k < 4; ++k) y[k] = 1.0 + 1e-9 * (threadIdx.x + k);
    for (int k = 0; k < 8; ++k) x[k] = 1e-6 * (k + 1);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if constexpr (DPP) {
                    switch (k) {   // a different broadcast lane per instruction, as the kernel's d-loop would have
#define C(K) case K: FMA_DPP(acc[K], y[r], x[(K + r) & 7], K); break;
                        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15)
#undef C
                    }
                } else {
                    FMA_PLAIN(acc[k], y[r], x[(k + r) & 7]);
                }
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int k = 0; k < 16; ++k) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int DPP>
static void run(int waves) {
    const int iters = 2000, blocks = 256 * waves;
    double* out; long long* st;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 8); (void)hipMalloc(&st, (size_t)blocks * 4 * 8);
    for (int w = 0; w < 2; ++w) rate<DPP><<<blocks, 256>>>(out, st, iters);
    (void)hipDeviceSynchronize();
    std::vector<long long> h((size_t)blocks * 4);
    (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%s waves/SIMD=%d: %.2f shader cycles per v_fmac_f64 per wave (%.2f per SIMD)\n", DPP ? "row_newbcast" : "plain       ", waves,
           (double)h[h.size() / 2] / (iters * 64.0), (double)h[h.size() / 2] / (iters * 64.0) / waves);
    (void)hipFree(out); (void)hipFree(st);
}

int main() {
    double* o; (void)hipMalloc(&o, 64 * 8);
    semantics<<<1, 64>>>(o);
    double h[64]; (void)hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    bool ok = true;
    for (int l = 0; l < 64; ++l) ok = ok && h[l] == 16.0 * (l / 16) + 5.0;
    printf("semantics row_newbcast:5 -> lane l reads lane 16 (l / 16) + 5: %s (lanes 0, 17, 63 got %.0f %.0f %.0f)\n", ok ? "ok" : "MISMATCH", h[0], h[17], h[63]);
    for (int w : {1, 2, 3}) { run<0>(w); run<1>(w); }
    return 0;
}
