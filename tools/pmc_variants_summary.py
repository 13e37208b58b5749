#!/usr/bin/env python3
"""Per-wave-step SQ counters of the rollout kernel from tools/pmc_variants.sh output.  Usage: pmc_variants_summary.py <dir> variant ..."""
import collections, csv, glob, sys
d, variants = sys.argv[1], sys.argv[2:]
T = 51.0
rows = {}
for v in variants:
    vv = v.replace(":", "_")
    tot = {}
    for part in "abc":
        for f in glob.glob(f"{d}/{part}_{vv}/*/*_counter_collection.csv"):
            agg = collections.defaultdict(list); grid = None
            for r in csv.DictReader(open(f)):
                if "mr_rollout_kernel" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"])); grid = int(r["Grid_Size"])
            for c, x in agg.items():
                x = x[len(x) // 2:]  # skip the settle launches' first half
                tot[c] = sum(x) / len(x) / (grid / 64) / T
    rows[v] = tot
names = sorted({c for t in rows.values() for c in t})
print("counter (per wave-step)".ljust(30) + "".join(v.rjust(14) for v in variants))
for c in names:
    print(c.ljust(30) + "".join(f"{rows[v].get(c, float('nan')):14.1f}" for v in variants))
