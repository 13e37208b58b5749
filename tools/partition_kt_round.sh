#!/bin/bash
# Evidence for the exit-time crash of `rocprofv3 --kernel-trace` after CU-masked streams were created: three profiled
# programs, each run ONCE, exit codes and the tail of stderr kept.  A run that times out or is killed ends the script.
#   1 cumask   tools/cumask_probe: pure HIP, no torch -- CU-masked streams, kernels, streams destroyed in order
#   2 close    tools/partition_kt_probe.py: train_collected(learner_cus=1) then DDPG.close() (dependents first, streams last)
#   3 noclose  the same without close(): what round 4 did (streams never destroyed, wrappers alive at interpreter exit)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/partition_kt; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
: > $O/status.txt
run() { name=$1; shift
  timeout -k 10 240 "$@" > $O/$name.out 2> $O/$name.err; rc=$?
  echo "$name rc=$rc" >> $O/status.txt
  tail -c 1500 $O/$name.err > $O/$name.err.tail
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "stopping: $name timed out / was killed" >> $O/status.txt; cat $O/status.txt; exit 1; fi
}
run cumask rocprofv3 --kernel-trace --output-format csv -d $O/kt_cumask -- $R/tools/cumask_probe
run close rocprofv3 --kernel-trace --output-format csv -d $O/kt_close -- python3 $R/tools/partition_kt_probe.py
run noclose rocprofv3 --kernel-trace --output-format csv -d $O/kt_noclose -- python3 $R/tools/partition_kt_probe.py --no-close
run plain_noclose python3 $R/tools/partition_kt_probe.py --no-close
find $O -name "*.csv" | xargs -r ls -la >> $O/status.txt
find $O -name "*.csv" -size +200k -delete    # keep the scratch small: the stats are not what this round is about
cat $O/status.txt
