#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/tools/pmc_run.py toeplitz 20 > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/c5_kernel_stats.csv; cut -d, -f1-4 $OUT/c5_kernel_stats.csv | cut -c1-150
