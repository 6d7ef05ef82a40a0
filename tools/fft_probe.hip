// fft_probe.hip — how does rocFFT execute the two halves of a four-step 2^22-point complex FFT (2048 x 2048)?
// (a) 2048 contiguous length-2048 transforms, (b) 2048 strided (stride 2048, dist 1) length-2048 transforms, in place and out of place.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <stdio.h>
#include <vector>
#define CK(e) do { auto _e = (e); if (_e != 0) { printf("error %d at line %d\n", (int)_e, __LINE__); return 1; } } while (0)

int run(const char* name, rocfft_result_placement place, size_t len, size_t stride, size_t dist, size_t batch, rocfft_precision prec, void* in, void* out) {
    rocfft_plan_description desc; CK(rocfft_plan_description_create(&desc));
    size_t strides[1] = {stride};
    CK(rocfft_plan_description_set_data_layout(desc, rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, nullptr, nullptr,
                                               1, strides, dist, 1, strides, dist));
    rocfft_plan plan; size_t lens[1] = {len};
    CK(rocfft_plan_create(&plan, place, rocfft_transform_type_complex_forward, prec, 1, lens, batch, desc));
    size_t wb = 0; CK(rocfft_plan_get_work_buffer_size(plan, &wb));
    rocfft_execution_info info; CK(rocfft_execution_info_create(&info));
    void* work = nullptr; if (wb) { CK(hipMalloc(&work, wb)); CK(rocfft_execution_info_set_work_buffer(info, work, wb)); }
    void* ib[1] = {in}; void* ob[1] = {place == rocfft_placement_inplace ? in : out};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(rocfft_execute(plan, ib, place == rocfft_placement_inplace ? nullptr : ob, info));
    CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int i = 0; i < 10; ++i) {
        CK(hipEventRecord(e0)); CK(rocfft_execute(plan, ib, place == rocfft_placement_inplace ? nullptr : ob, info)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-44s work=%zu B  best %.1f us\n", name, wb, best * 1e3);
    return 0;
}
int main() {
    rocfft_setup();
    const size_t M = (size_t)1 << 22;
    void *a, *b; CK(hipMalloc(&a, M * 16)); CK(hipMalloc(&b, M * 16)); CK(hipMemset(a, 0, M * 16)); CK(hipMemset(b, 0, M * 16));
    for (auto prec : {rocfft_precision_double, rocfft_precision_single}) {
        printf("== %s\n", prec == rocfft_precision_double ? "double" : "single");
        run("contiguous 2048 x len2048 inplace", rocfft_placement_inplace, 2048, 1, 2048, 2048, prec, a, b);
        run("contiguous 2048 x len2048 outofplace", rocfft_placement_notinplace, 2048, 1, 2048, 2048, prec, a, b);
        run("strided(2048) 2048 x len2048 inplace", rocfft_placement_inplace, 2048, 2048, 1, 2048, prec, a, b);
        run("strided(2048) 2048 x len2048 outofplace", rocfft_placement_notinplace, 2048, 2048, 1, 2048, prec, a, b);
        run("contiguous 1024 x len4096 inplace", rocfft_placement_inplace, 4096, 1, 4096, 1024, prec, a, b);
        run("strided(4096) 4096 x len1024 inplace", rocfft_placement_inplace, 1024, 4096, 1, 4096, prec, a, b);
        run("full 2^22 c2c inplace", rocfft_placement_inplace, M, 1, M, 1, prec, a, b);
        run("full 2^22 c2c outofplace", rocfft_placement_notinplace, M, 1, M, 1, prec, a, b);
    }
    rocfft_cleanup();
    return 0;
}
