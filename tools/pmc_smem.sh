#!/bin/bash
# usage: tools/pmc_smem.sh <config of tools/pmc_run.py> <kernel regex> <out dir under gpurun_out/>: scalar-cache counters of one kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; CFG=$1; OUT=$R/gpurun_out/$3; mkdir -p $OUT
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQC_TC_STALL SQC_DCACHE_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 $R/tools/pmc_run.py $CFG 3 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/p2 -- python3 $R/tools/pmc_run.py $CFG 3 > $OUT/p2.log 2>&1
python3 $R/tools/pmc_table.py $OUT "$2"
