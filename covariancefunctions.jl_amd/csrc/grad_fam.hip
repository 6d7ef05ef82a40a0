// grad_fam.hip — compiled once per kernel family (-DCOVGRAM_FAM=<covgram_family>); exports
// launch_grad_family_<FAM>.
#include "grad_mvm.hpp"
#include "grad_bcast.hpp"
#include "grad_wide.hpp"

#ifndef COVGRAM_FAM
#error "compile with -DCOVGRAM_FAM=<0..12>"
#endif

namespace covgram {
#define CG_CAT2(a, b) a##b
#define CG_CAT(a, b) CG_CAT2(a, b)
int CG_CAT(launch_grad_family_, COVGRAM_FAM)(const GradArgs& a, int dtype) {
    return launch_grad_family<COVGRAM_FAM>(a, dtype);
}
int CG_CAT(launch_grad_wide_family_, COVGRAM_FAM)(const GradWideArgs& a, int dtype) {
    return launch_grad_wide_family<COVGRAM_FAM>(a, dtype);
}
}  // namespace covgram
