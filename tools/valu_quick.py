#!/usr/bin/env python3
"""Mean SQ_INSTS_VALU (or any one counter) per wave and env step of the fused rollout kernels in rocprofv3 --pmc output directories:
python tools/valu_quick.py <dir> [<dir> ...]   (later half of the dispatches of every kernel / counter / grid; T = 51)"""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "mr_rollout_kernel" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:60], r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v = v[len(v)//2:]
        print(d.split("/")[-1], k, len(v), sum(v)/len(v)/(int(k[2])/64)/51)
