#!/bin/bash
# rocprofv3 --kernel-trace --stats over the contract command (python3 bench.py), summary -> gpurun_out/<prefix>_bench_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --no-cpu-baseline --no-configs > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/bench_kernel_stats.csv; head -12 $OUT/bench_kernel_stats.csv | cut -c1-200
