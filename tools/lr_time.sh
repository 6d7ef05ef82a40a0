cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/lr2 && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/lr2 -- python3 $GRAFT_REPO_ROOT/tools/pmc_run.py lowrank 20 > /dev/null 2>&1; f=$(find $GRAFT_REPO_ROOT/gpurun_out/lr2 -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'EOP'
import csv,sys
for r in list(csv.reader(open(sys.argv[1])))[1:8]:
    if "covgram" in r[0]: print(r[0][:50], r[1], r[3])
EOP
