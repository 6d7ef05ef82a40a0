"""profiles/rNN_dense_mfma_pmc.json from the counter CSVs of tools/pmc_passes.sh (so that the file bench.py reads for roofline.traffic is produced
by a script from the passes, not typed).   usage: pmc_json.py <dir under gpurun_out> <kernel regex> <label> <algorithmic bytes> <out.json>"""
import csv, glob, json, re, sys, collections
out, pat, label, alg, dst = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), sys.argv[5]
tot = collections.defaultdict(list)
names = set()
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if re.search(pat, row["Kernel_Name"]):
            tot[row["Counter_Name"]].append(float(row["Counter_Value"]))
            names.add(row["Kernel_Name"].split("(")[0].replace("void ", "").strip())
assert len(names) == 1, f"the pattern matches {len(names)} different kernels: {sorted(names)}"
fetch = sum(tot["FETCH_SIZE"]) / len(tot["FETCH_SIZE"]); write = sum(tot["WRITE_SIZE"]) / len(tot["WRITE_SIZE"])
json.dump({"kernel": label, "kernel_name": sorted(names)[0], "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (tools/pmc_passes.sh, {out.split('/')[-1]}), mean of {len(tot['FETCH_SIZE'])} dispatches; written by tools/pmc_json.py",
           "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
           "fetch_correction": "x2 (gfx950 counts 128-B requests at 64 B for 16 B/lane streaming reads; MI355X_MICROARCH.md, HBM)",
           "hbm_bytes_per_launch": (2 * fetch + write) * 1024.0, "algorithmic_bytes_per_launch": alg}, open(dst, "w"), indent=1)
print(open(dst).read())
