#!/bin/bash
# Where the Kronecker kernels' time goes (README.md:205-210 case): the KRON_DIAG build of the library with parts switched off
#   COVGRAM_KRON_DIAG bit 1 = no MFMAs, 2 = no global stores, 4 = no tile loads after the prologue
# build: cd covariancefunctions.jl_amd && cp -r build build_diag && rm build_diag/kron_f* && make BUILD=build_diag LIBDIR=lib_diag KRON_EXTRA=-DKRON_DIAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
export COVGRAM_LIB=$R/covariancefunctions.jl_amd/lib_diag/libcovgram.so
for d in 0 1 2 4 3 5 6 7; do
  export COVGRAM_KRON_DIAG=$d
  rocprofv3 --kernel-trace -d $OUT/d$d -o k -- python3 $R/tools/pmc_run.py ${2:-kron64} 20 > /dev/null 2>&1
  echo "== COVGRAM_KRON_DIAG=$d"; python3 $R/tools/rocpd_stats.py $OUT/d$d/k_results.db kron_
done
