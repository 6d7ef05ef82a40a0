/* abi_layout.c — compiled and run by tests/test_host.py (CPU): pins the layout of every struct that crosses the C ABI of
 * include/covgram.h and prints it as a table ("struct field offset size" per line) that the test compares with the two
 * bindings' mirrors — the ctypes Structures of covgram/_ffi.py and the `struct` blocks of julia/CovGram.jl (parsed as text:
 * Julia is not in the image).  A drift between header, library and bindings fails here, not in a caller's memory. */
#include <stddef.h>
#include <stdio.h>

#include "../include/covgram.h"

_Static_assert(sizeof(covgram_kernel) == 40, "covgram_kernel: 4 x int32 + 3 x double");
_Static_assert(offsetof(covgram_kernel, family) == 0 && offsetof(covgram_kernel, trait) == 4 && offsetof(covgram_kernel, p) == 8 &&
               offsetof(covgram_kernel, power) == 12, "covgram_kernel: int32 head");
_Static_assert(offsetof(covgram_kernel, param) == 16 && offsetof(covgram_kernel, lengthscale) == 24 && offsetof(covgram_kernel, scale) == 32,
               "covgram_kernel: double tail");
_Static_assert(COVGRAM_COMPOSITE_MAX_TERMS == 8 && COVGRAM_COMPOSITE_MAX_FACTORS == 8, "composite limits are ABI");
_Static_assert(offsetof(covgram_kernel_composite, head) == 0, "a composite is passed as a pointer to its head");
_Static_assert(offsetof(covgram_kernel_composite, nterms) == 40 && offsetof(covgram_kernel_composite, nfactors) == 44, "composite counters");
_Static_assert(offsetof(covgram_kernel_composite, factors) == 80, "factors start 8-byte aligned after 1 + 8 int32 (+ 4 bytes padding)");
_Static_assert(sizeof(covgram_kernel_composite) == 80 + 8 * 40, "composite size");

#define ROW(S, F) printf(#S " " #F " %zu %zu\n", offsetof(S, F), sizeof(((S*)0)->F))

int main(void) {
    ROW(covgram_kernel, family); ROW(covgram_kernel, trait); ROW(covgram_kernel, p); ROW(covgram_kernel, power);
    ROW(covgram_kernel, param); ROW(covgram_kernel, lengthscale); ROW(covgram_kernel, scale);
    printf("covgram_kernel sizeof %zu %zu\n", sizeof(covgram_kernel), _Alignof(covgram_kernel));
    ROW(covgram_kernel_composite, head); ROW(covgram_kernel_composite, nterms); ROW(covgram_kernel_composite, nfactors);
    ROW(covgram_kernel_composite, factors);
    printf("covgram_kernel_composite sizeof %zu %zu\n", sizeof(covgram_kernel_composite), _Alignof(covgram_kernel_composite));
    printf("enum COVGRAM_EQ %d\nenum COVGRAM_EXP %d\nenum COVGRAM_RQ %d\nenum COVGRAM_GAMMAEXP %d\nenum COVGRAM_CAUCHY %d\nenum COVGRAM_IMQ %d\n"
           "enum COVGRAM_MATERNP %d\nenum COVGRAM_DOT %d\nenum COVGRAM_EXPDOT %d\nenum COVGRAM_CONSTANT %d\nenum COVGRAM_COMPOSITE %d\n"
           "enum COVGRAM_ISOTROPIC %d\nenum COVGRAM_DOTPRODUCT %d\nenum COVGRAM_F32 %d\nenum COVGRAM_F64 %d\nenum COVGRAM_HOST %d\nenum COVGRAM_DEVICE %d\n",
           COVGRAM_EQ, COVGRAM_EXP, COVGRAM_RQ, COVGRAM_GAMMAEXP, COVGRAM_CAUCHY, COVGRAM_IMQ, COVGRAM_MATERNP, COVGRAM_DOT, COVGRAM_EXPDOT,
           COVGRAM_CONSTANT, COVGRAM_COMPOSITE, COVGRAM_ISOTROPIC, COVGRAM_DOTPRODUCT, COVGRAM_F32, COVGRAM_F64, COVGRAM_HOST, COVGRAM_DEVICE);
    return 0;
}
