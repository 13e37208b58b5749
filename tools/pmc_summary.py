#!/usr/bin/env python3
"""Per-wave(-step) instruction mix from tools/pmc_rollout.sh output.  Usage: pmc_summary.py <tag> [kernel substr] [steps per launch]"""
import csv, collections, glob, sys
tag = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "rollout"; T = float(sys.argv[3]) if len(sys.argv) > 3 else 51.0
tot = {}
for part in "ab":
    for f in glob.glob(f"gpurun_out/pmc_{tag}_{part}/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list); grid = None
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"])); grid = int(r["Grid_Size"])
        for c, x in agg.items():
            tot[c] = sum(x) / len(x) / (grid / 64) / T
for c in sorted(tot):
    print(f"{c:28s} {tot[c]:9.1f}")
