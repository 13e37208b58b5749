#!/bin/bash
# Size sweep of both bench modes (run on the GPU box via gpurun):  bash tools/size_sweep.sh > gpurun_out/size_sweep.txt
# Steps are scaled so that every timed region lasts >= ~150 ms (settled clocks).
R=${GRAFT_REPO_ROOT:-.}
echo "# python bench.py --no-cpu-baseline --no-step-path --no-mixed-set --streams 1 --mode {rollout,step} --envs-per-gpu N   (sigma=1, noise_math=fast, carry=f64, one MI355X)"
echo "# value = end-to-end env-steps/s of the timed region; kernel = HIP events attached to the dispatches (rollout: the one-stream sustained region);"
echo "# rollout in-kernel = N x 51 / kernel time; step frac = 97 B x N / kernel time / 8 TB/s (algorithmic bytes; the step kernel moves 106.6)"
for spec in "4096 102000 10200" "65536 102000 10200" "262144 51000 5100" "2097152 10200 1020" "16777216 2040 510"; do
  set -- $spec
  for mode in rollout step; do
    python3 $R/bench.py --no-cpu-baseline --no-step-path --no-mixed-set --streams 1 --mode $mode --envs-per-gpu $1 --steps $2 --warmup $3 --sustained-steps $(( $2 / 5 )) > /tmp/ss.json 2>/tmp/ss.err || { echo "N=$1 $mode FAILED"; tail -2 /tmp/ss.err; continue; }
    python3 - "$1" "$mode" <<'PY'
import json, sys
d = json.loads([l for l in open("/tmp/ss.json") if l.startswith("{")][-1])
r = d["roofline"]
n = int(sys.argv[1])
if sys.argv[2] == "rollout":
    extra = f"in-kernel {r['achieved']:7.2f} G env-steps/s"
else:
    extra = f"algorithmic {r['achieved']:7.1f} GB/s = {r['frac']:.3f} of 8 TB/s"
print(f"N={n:9d} {sys.argv[2]:8s} {d['value']/1e9:8.2f} G env-steps/s {d['ms_per_step']*1e3:9.2f} us/step | kernel avg {r['avg_kernel_us']:10.2f} us  {extra}")
PY
  done
done
