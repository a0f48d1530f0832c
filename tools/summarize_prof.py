"""Summarise rocprofv3 CSV output (kernel stats + PMC counters) of tools/gpu_profile.sh."""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]
for f in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats", f)
    for i, row in enumerate(csv.DictReader(open(f))):
        if i < 8:
            print({k: row[k] for k in list(row)[:8]})
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        agg = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("== counters", f)
        for kname, ctrs in agg.items():
            if "gemm" not in kname and "quant" not in kname:
                continue
            print(" ", kname, {c: (len(v), sum(v) / len(v)) for c, v in ctrs.items()})
