#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned lane = threadIdx.x;
    unsigned a = 100 + lane, b = 200 + lane;
    auto r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    auto r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[lane] = r32[0]; out[64 + lane] = r32[1]; out[128 + lane] = r16[0]; out[192 + lane] = r16[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"swap32 first ", "swap32 second", "swap16 first ", "swap16 second"};
    for (int r = 0; r < 4; ++r) { printf("%s:", names[r]); for (int l = 0; l < 64; l += 8) printf(" %u", h[r * 64 + l]); printf("\n"); }
    return 0;
}
