// kron_limits.hpp — what the Kronecker mode-product kernels (kron_kernels.hpp) accept; the planner (kron.hip) routes anything else
// to the library GEMM.  The tile loaders address with 32-bit element offsets from a wave-uniform base, so the rows of one tile and
// the columns of one factor fragment must lie within 2^31 elements.
#pragma once
#include <stdint.h>

namespace covgram {
namespace kron {
constexpr int PAIR_MAX_K2 = 128;   // slab rows (c_q) the fused pass holds in its accumulators
constexpr int PAIR_MIN_SIDE = 48;  // r_{q-1} and c_q from which the fused pass pays (three of its four strips / blocks busy)
inline bool span_ok(int64_t rows, int64_t row_stride) { return rows * row_stride < ((int64_t)1 << 31); }
inline bool pair_ok(int64_t K1, int64_t K2, int64_t ld2, int64_t ld3) { return K2 <= PAIR_MAX_K2 && span_ok(128, ld3) && span_ok(K1, ld2); }
inline bool mode_ok(int64_t K, int64_t post, int64_t ld) { return span_ok(16, post) && span_ok(K, ld); }
inline bool modet_ok(int64_t K, int64_t ld) { return span_ok(128, K) && span_ok(K, ld); }
}  // namespace kron
}  // namespace covgram
