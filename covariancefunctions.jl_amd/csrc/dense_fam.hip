// dense_fam.hip — compiled once per kernel family (-DCOVGRAM_FAM=<covgram_family>) so the nine families
// build in parallel; exports launch_dense_family_<FAM>.
#include "dense_mvm.hpp"
#include "dense_sym32.hpp"
#include "dense_bcast.hpp"
#include "dense_wide.hpp"

#ifndef COVGRAM_FAM
#error "compile with -DCOVGRAM_FAM=<0..12>"
#endif

namespace covgram {
#define CG_CAT2(a, b) a##b
#define CG_CAT(a, b) CG_CAT2(a, b)
int CG_CAT(launch_dense_family_, COVGRAM_FAM)(const DenseArgs& a, int dtype) {
    return launch_dense_family<COVGRAM_FAM>(a, dtype);
}
int CG_CAT(launch_dense_wide_family_, COVGRAM_FAM)(const DenseArgs& a, int dtype) {
    return launch_dense_wide_family<COVGRAM_FAM>(a, dtype);
}
}  // namespace covgram
