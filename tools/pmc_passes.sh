#!/bin/bash
# usage: tools/pmc_passes.sh <config of tools/pmc_run.py> <kernel name regex> <out prefix under gpurun_out/>
# Three rocprofv3 --pmc passes (SQ: 8 counters per pass; TCC: FETCH_SIZE and WRITE_SIZE do not fit one pass) + a kernel-trace pass.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; CFG=$1; OUT=$R/gpurun_out/$3; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/p1 -- python3 $R/tools/pmc_run.py $CFG 3 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT/p2 -- python3 $R/tools/pmc_run.py $CFG 3 > $OUT/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -- python3 $R/tools/pmc_run.py $CFG 3 > $OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p4 -- python3 $R/tools/pmc_run.py $CFG 3 > $OUT/p4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/tools/pmc_run.py $CFG 10 > $OUT/kt.log 2>&1
python3 - "$OUT" "$2" <<'PY'
import csv, glob, sys, re, collections
out, pat = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if re.search(pat, row["Kernel_Name"]):
            tot[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("# mean per dispatch of kernels matching", pat)
for k, v in sorted(tot.items()):
    print(f"{k:28s} {sum(v)/len(v):16.0f}   ({len(v)} dispatches)")
for f in glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        print("stats:", row["Name"][:90], row["Calls"], "calls avg", row["AverageNs"], "ns")
PY
