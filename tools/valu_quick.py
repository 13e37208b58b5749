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
