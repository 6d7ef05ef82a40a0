// kron_inst.hip — one translation unit per (element type, kernel) of the Kronecker mode-product kernels (kron_kernels.hpp):
// built as kron_f32_0 ... kron_f64_2 with -DKRON_T=float|double -DKRON_PART=0 (pair) | 1 (mode) | 2 (modet).
#include "kron_kernels.hpp"

namespace covgram {
namespace kron {
#if KRON_PART == 0
template int run_pair<KRON_T>(covgram_ctx*, const KRON_T*, KRON_T*, const KRON_T*, int64_t, int64_t, int64_t, const KRON_T*, int64_t, int64_t, int64_t, int64_t, KRON_T, KRON_T);
#elif KRON_PART == 1
template int run_mode<KRON_T>(covgram_ctx*, const KRON_T*, KRON_T*, const KRON_T*, int64_t, int64_t, int64_t, int64_t, int64_t, KRON_T, KRON_T);
#else
template int run_modet<KRON_T>(covgram_ctx*, const KRON_T*, KRON_T*, const KRON_T*, int64_t, int64_t, int64_t, int64_t, KRON_T, KRON_T);
#endif
}  // namespace kron
}  // namespace covgram
