#!/bin/bash
# Busy / wait cycle counters of the rollout kernel (run on the GPU box via gpurun).  Usage: tools/pmc_busy.sh <tag> [bench args]
R=$GRAFT_REPO_ROOT; tag=$1; shift
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_${tag}_a -- python3 $R/bench.py --no-cpu-baseline --no-step-path --no-mixed-set --steps 102 --settle-episodes 0 --warmup 102 "$@" > $R/gpurun_out/pmc_${tag}_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/pmc_${tag}_b -- python3 $R/bench.py --no-cpu-baseline --no-step-path --no-mixed-set --steps 102 --settle-episodes 0 --warmup 102 "$@" > $R/gpurun_out/pmc_${tag}_b.log 2>&1
echo done
