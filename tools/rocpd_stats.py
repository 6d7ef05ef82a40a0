"""Per-kernel summary of a rocprofv3 results .db (the default sqlite output): calls, mean / min / max duration (us), grid, workgroup, LDS, scratch.
usage: python tools/rocpd_stats.py <results.db> [substring]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {ks}")}
agg = collections.defaultdict(list)
for kid, s, e, gx, wx, lds, scr in cur.execute(f"select kernel_id, start, end, grid_size_x, workgroup_size_x, group_segment_size, private_segment_size from {kd}"):
    agg[(names[kid], gx, wx, lds, scr)].append((e - s) * 1e-3)
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for (n, gx, wx, lds, scr), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if sub in n:
        print(f"{len(v):5d} calls  mean {sum(v)/len(v):9.2f} us  min {min(v):9.2f}  max {max(v):9.2f}  grid {gx//wx:6d} x {wx:4d}  lds {lds:6d} scratch {scr:4d}  {n[:110]}")
