#!/bin/bash
# usage: tools/trace_case.sh <case> ...   rocprofv3 kernel trace of tools/trace_case.py: per-kernel averages and the gaps between consecutive kernels of one call
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rm -rf /tmp/tc_$c; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tc_$c -- python3 $GRAFT_REPO_ROOT/tools/trace_case.py $c > /dev/null 2>&1
  f=$(find /tmp/tc_$c -name "*kernel_stats.csv" | head -1); t=$(find /tmp/tc_$c -name "*kernel_trace.csv" | head -1)
  echo "== $c"
  python3 - "$f" "$t" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "covgram" in r["Name"] or "rocfft" in r["Name"].lower()]
for x in rows[:8]: print(f"  {x['Name'][:90]:90s} calls {x['Calls']:>4s}  avg {float(x['AverageNs'])/1e3:8.2f} us  min {float(x['MinNs'])/1e3:8.2f}")
tr = [r for r in csv.DictReader(open(sys.argv[2])) if "covgram" in r["Kernel_Name"] or "rocfft" in r["Kernel_Name"].lower()]
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = tr[-12:]
for a, b in zip(tail[:-1], tail[1:]):
    print(f"    {a['Kernel_Name'][:60]:60s} {(int(a['End_Timestamp'])-int(a['Start_Timestamp']))/1e3:8.2f} us, then {(int(b['Start_Timestamp'])-int(a['End_Timestamp']))/1e3:6.2f} us to the next start")
PY
done
