# rocprofv3 kernel trace of DDPG.train's iteration (tools/train_loop_probe.py, 256 envs): durations of the launches of one iteration (step kernel with actor + replay sink, update with the policy upload)
O=$GRAFT_REPO_ROOT/gpurun_out/train_trace; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $GRAFT_REPO_ROOT/tools/train_loop_probe.py --envs 256 --steps 600 --forms step --learners fused > $O/probe.out 2> $O/probe.log
echo "rc $?"
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if 'mrsim' in r['Kernel_Name']][-2 * 600:]            # the timed run: two launches per iteration
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    k = a["Kernel_Name"].split("(")[0][:60]
    dur[k].append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
    gap[k].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
with open("$O/summary.txt", "w") as out:
    for k in dur:
        d, g = sorted(dur[k]), sorted(gap[k])
        line = "%-62s n=%4d  duration median %6.2f us   gap to the next launch median %6.2f us" % (k, len(d), d[len(d) // 2], g[len(g) // 2])
        print(line); out.write(line + "\n")
PY
rm -f $O/*/*kernel_trace.csv $O/*kernel_trace.csv
