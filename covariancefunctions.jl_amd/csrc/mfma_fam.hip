// mfma_fam.hip — compiled once per profile family that has a matrix-core dense path (-DCOVGRAM_FAM=<covgram_family>);
// exports launch_mfma_family_<FAM>.
#include "dense_mfma.hpp"

#ifndef COVGRAM_FAM
#error "compile with -DCOVGRAM_FAM=<0, 2, 4, 5, 6, 7, 8>"
#endif

namespace covgram {
#define CG_CAT2(a, b) a##b
#define CG_CAT(a, b) CG_CAT2(a, b)
int CG_CAT(launch_mfma_family_, COVGRAM_FAM)(const MfmaArgs& a, bool query) { return launch_mfma_family<COVGRAM_FAM>(a, query); }
}  // namespace covgram
