"""Per-kernel means of the rocprofv3 --pmc passes written by tools/pmc_passes.sh.   usage: pmc_table.py <dir under gpurun_out> <kernel regex>"""
import csv, glob, sys, re, collections
out, pat = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if re.search(pat, row["Kernel_Name"]):
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void covgram::", "")
            tot[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
names = sorted(tot)
ctrs = sorted({c for n in names for c in tot[n]})
print(f"{'counter (mean per dispatch)':28s}" + "".join(f"{n[:34]:>36s}" for n in names))
for c in ctrs:
    print(f"{c:28s}" + "".join(f"{(sum(tot[n][c]) / len(tot[n][c]) if tot[n][c] else float('nan')):36.0f}" for n in names))
