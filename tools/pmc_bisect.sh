#!/bin/bash
# Round-2 diagnosis of the rocprofv3 --pmc SIGSEGV seen in round 1 (rollout-mode bench only): ONE pass per bench
# variant, each toggling one suspect, with bench.py's phase markers on stderr.  Run on the GPU box via gpurun:
#   tools/pmc_bisect.sh > gpurun_out/pmc_bisect/summary.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_bisect; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export MRSIM_BENCH_TRACE=1
run() {
  tag=$1; shift
  timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$tag -- python3 $R/bench.py --no-cpu-baseline --steps 102 "$@" > $O/$tag.json 2> $O/$tag.log
  rc=$?
  n=$(cat $O/$tag/*/*counter_collection.csv 2>/dev/null | wc -l)
  echo "$tag rc=$rc csv_rows=$n last_phase=[$(grep '^\[bench' $O/$tag.log | tail -1)] args: $*"
}
run base
run nostep --no-step-path
run nomixed --no-mixed-set
run nosettle --settle-episodes 0
run shortwarm --warmup 102
run nostep_nomixed --no-step-path --no-mixed-set
run all_off --no-step-path --no-mixed-set --settle-episodes 0 --warmup 102
echo bisect done
