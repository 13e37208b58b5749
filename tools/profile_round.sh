#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run via gpurun):  tools/profile_round.sh r01
# 1) kernel trace + stats of the default bench command; 2) PMC passes (separate, as the guide
# prescribes): FETCH_SIZE / WRITE_SIZE for HBM traffic, plus a calibration run on tools/membench
# whose bytes are known exactly.
R=$GRAFT_REPO_ROOT; tag=${1:-r01}; O=$R/gpurun_out/prof_$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline > $O/kt_bench.json 2> $O/kt.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_step -- python3 $R/bench.py --no-cpu-baseline --mode step > $O/kt_step_bench.json 2> $O/kt_step.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --no-cpu-baseline --steps 102 > /dev/null 2> $O/pmc_$c.log
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_step_$c -- python3 $R/bench.py --no-cpu-baseline --steps 102 --mode step --launch eager > /dev/null 2> $O/pmc_step_$c.log
  rocprofv3 --pmc $c --output-format csv -d $O/cal_$c -- $R/tools/membench 16777216 20 > /dev/null 2> $O/cal_$c.log
  rocprofv3 --pmc $c --output-format csv -d $O/cal262k_$c -- $R/tools/membench 262144 20 > /dev/null 2> $O/cal262k_$c.log
done
echo profile_round done
