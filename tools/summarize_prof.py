"""Summarise rocprofv3 CSV output (kernel stats + PMC counters) of tools/gpu_profile.sh."""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]
for f in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats (rocprofv3 --kernel-trace --stats)")
    for i, row in enumerate(csv.DictReader(open(f))):
        if i < 6:
            name = row["Name"][:90]
            print(f"  {name:90s} calls={row['Calls']:>5s} avg_ns={float(row['AverageNs']):>12.1f} "
                  f"min_ns={row['MinNs']:>9s} max_ns={row['MaxNs']:>9s} pct={row['Percentage']}")
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        agg = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print(f"== counters ({sub}): per-launch averages")
        for kname, ctrs in agg.items():
            if not any(k in kname for k in ("gemm", "quant", "skinny", "repack", "fnuz", "strip")):
                continue
            print("  ", kname)
            for c, v in sorted(ctrs.items()):
                print(f"      {c:28s} n={len(v):4d} avg={sum(v) / len(v):16.1f}")
