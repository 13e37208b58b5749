#!/bin/bash
# Collect one round's rocprofv3 evidence on the GPU box (run via gpurun):  bash tools/profile_round.sh r02
# Every pass writes its exit code to $O/status.txt; tools/collect_profiles.py refuses to build profiles/<tag>/ from a
# round in which a pass failed or whose CSVs are older than the library / bench.py they claim to describe.
#  1) kernel trace + stats of the default bench command (rollout, 2 sub-shard streams), of `--streams 1` (one launch per
#     episode: the kernel duration bench.py's roofline quotes; also --noise-law per_stage and --mismatched), of --mode step, and of
#     the N = 2 097 152 streaming point (rollout both laws, step kernel);
#  2) PMC passes, each counter group in its own run with no trace flags (MI355X_MICROARCH.md, rocprofv3 PMC slots):
#     FETCH_SIZE / WRITE_SIZE of the rollout kernel (both carries) and of the eager step path, calibrated on
#     tools/membench whose bytes are known exactly; SQ instruction-mix and busy / wait counters of the rollout kernel;
#  3) tools/instbench --json: per-instruction issue costs that price the instruction mix.
# The PMC passes use the bench's own flags to skip its hipGraph leg (graph capture under --pmc crashes rocprofv3 on
# ROCm 7.2 -- bench.py also skips it by itself when it sees ROCPROF_COUNTER_COLLECTION).
# A round is longer than one gpurun call may last (20 min): `profile_round.sh <tag> a` (kernel traces) and `... b` / `... c` (counter
# passes) run in separate calls; each part writes status_<part>.txt / manifest_<part>.txt / sha_<part>.txt, which
# tools/collect_profiles.py unites (and refuses if the parts were taken with different builds).
R=$GRAFT_REPO_ROOT; tag=${1:-r02}; part=${2:-all}; O=$R/gpurun_out/prof_$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
: > $O/status_$part.txt
want() { [ "$part" = all ] || [ "$part" = "$1" ]; }   # want <part>: does this invocation run the passes of that part?
PART_OF_NEXT=a
pass() {  # pass <name> <cmd...>: run (if the pass belongs to this invocation's part), record the exit code
  name=$1; shift
  want $PART_OF_NEXT || return 0
  timeout -k 10 400 "$@" > $O/$name.out 2> $O/$name.log
  echo "$name $?" >> $O/status_$part.txt
  # gpurun copies back at most 64 MiB: per-dispatch traces are kept only where tools/collect_profiles.py reads them (the rollout
  # kernels' rows of the one-stream run and of the actor probes); everywhere else the --stats summary is what is committed
  for f in $(find $O/$name -name "*kernel_trace.csv" 2>/dev/null); do
    case $name in
      kt_s1|kt_actor|kt_actor_bf|kt_actor_b1) (head -1 $f; grep mr_rollout $f) > $f.tmp; mv $f.tmp $f ;;
      *) rm -f $f ;;
    esac
  done
}
PMCARGS="--no-cpu-baseline --no-step-path --no-mixed-set --no-power --no-actor-leg --no-learner-leg --no-other-law --no-facade-leg --no-streaming-point --no-consumers-leg --steps 102 --warmup 102 --settle-episodes 20 --sustained-steps 0"
S1ARGS="--no-cpu-baseline --no-power --streams 1 --no-step-path --no-mixed-set --no-other-law --no-actor-leg --no-learner-leg --no-facade-leg --no-streaming-point --no-consumers-leg"
pass kt rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --no-power
pass kt_s1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_s1 -- python3 $R/bench.py $S1ARGS
pass kt_s1_ps rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_s1_ps -- python3 $R/bench.py $S1ARGS --noise-law per_stage
pass kt_s1_mis rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_s1_mis -- python3 $R/bench.py $S1ARGS --mismatched
pass kt_step rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_step -- python3 $R/bench.py --no-cpu-baseline --no-power --mode step --no-facade-leg --no-consumers-leg
# ---- SURVEY H4's streaming point: N = 2 097 152 envs on this one GPU (rollout both laws, step kernel)
BIG="--envs-per-gpu 2097152 --settle-episodes 40 --warmup 510 --steps 2040 --sustained-steps 2040"
pass kt_2m rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_2m -- python3 $R/bench.py $S1ARGS $BIG
pass kt_2m_ps rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_2m_ps -- python3 $R/bench.py $S1ARGS $BIG --noise-law per_stage
pass kt_2m_step rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_2m_step -- python3 $R/bench.py --no-cpu-baseline --no-power --no-facade-leg --mode step --launch eager --envs-per-gpu 2097152 --settle-episodes 10 --warmup 102 --steps 510
PART_OF_NEXT=b
for c in FETCH_SIZE WRITE_SIZE; do
  for carry in f64 f32; do
    pass pmc_${carry}_$c rocprofv3 --pmc $c --output-format csv -d $O/pmc_${carry}_$c -- python3 $R/bench.py $PMCARGS --carry $carry
  done
  pass pmc_ps_f64_$c rocprofv3 --pmc $c --output-format csv -d $O/pmc_ps_f64_$c -- python3 $R/bench.py $PMCARGS --carry f64 --noise-law per_stage
  pass pmc_mis_f64_$c rocprofv3 --pmc $c --output-format csv -d $O/pmc_mis_f64_$c -- python3 $R/bench.py $PMCARGS --carry f64 --mismatched
  pass pmc_2m_f64_$c rocprofv3 --pmc $c --output-format csv -d $O/pmc_2m_f64_$c -- python3 $R/bench.py $PMCARGS --carry f64 --envs-per-gpu 2097152
  pass pmc_2m_ps_f64_$c rocprofv3 --pmc $c --output-format csv -d $O/pmc_2m_ps_f64_$c -- python3 $R/bench.py $PMCARGS --carry f64 --envs-per-gpu 2097152 --noise-law per_stage
  pass pmc_2m_step_$c rocprofv3 --pmc $c --output-format csv -d $O/pmc_2m_step_$c -- python3 $R/bench.py $PMCARGS --mode step --launch eager --envs-per-gpu 2097152
  pass pmc_step_$c rocprofv3 --pmc $c --output-format csv -d $O/pmc_step_$c -- python3 $R/bench.py $PMCARGS --mode step --launch eager
  pass cal_$c rocprofv3 --pmc $c --output-format csv -d $O/cal_$c -- $R/tools/membench 16777216 20
  pass cal262k_$c rocprofv3 --pmc $c --output-format csv -d $O/cal262k_$c -- $R/tools/membench 262144 20
done
PART_OF_NEXT=c
for carry in f64 f32; do
  pass valu_a_$carry rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/valu_a_$carry -- python3 $R/bench.py $PMCARGS --carry $carry
  pass valu_b_$carry rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VMEM_WR --output-format csv -d $O/valu_b_$carry -- python3 $R/bench.py $PMCARGS --carry $carry
  pass valu_c_$carry rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD --output-format csv -d $O/valu_c_$carry -- python3 $R/bench.py $PMCARGS --carry $carry
done
# the per-stage law (the library's default / parity mode) beside bench.py's default collapsed law
pass valu_a_ps rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/valu_a_ps -- python3 $R/bench.py $PMCARGS --carry f64 --noise-law per_stage
pass valu_b_ps rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VMEM_WR --output-format csv -d $O/valu_b_ps -- python3 $R/bench.py $PMCARGS --carry f64 --noise-law per_stage
pass valu_c_ps rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD --output-format csv -d $O/valu_c_ps -- python3 $R/bench.py $PMCARGS --carry f64 --noise-law per_stage
# the mismatched model (collapsed law: two Philox calls, three Box-Muller pairs)
pass valu_a_mis rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/valu_a_mis -- python3 $R/bench.py $PMCARGS --carry f64 --mismatched
pass valu_b_mis rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VMEM_WR --output-format csv -d $O/valu_b_mis -- python3 $R/bench.py $PMCARGS --carry f64 --mismatched
pass valu_c_mis rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD --output-format csv -d $O/valu_c_mis -- python3 $R/bench.py $PMCARGS --carry f64 --mismatched
# ---- the fused rollout with the DDPG actor as its policy source (tools/actor_probe.py: 262 144 envs, 51 steps per launch)
pass kt_actor rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_actor -- python3 $R/tools/actor_probe.py --launches 200
pass pmc_actor_a rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/pmc_actor_a -- python3 $R/tools/actor_probe.py --launches 6 --discard 0
pass pmc_actor_b rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_actor_b -- python3 $R/tools/actor_probe.py --launches 6 --discard 0
for c in FETCH_SIZE WRITE_SIZE; do
  pass pmc_actor_$c rocprofv3 --pmc $c --output-format csv -d $O/pmc_actor_$c -- python3 $R/tools/actor_probe.py --launches 6 --discard 0
done
# ---- the same with the actor's 64 x 64 layer in bf16 x 3 arithmetic
pass kt_actor_bf rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_actor_bf -- python3 $R/tools/actor_probe.py --math bf16x3 --launches 200
pass pmc_actor_bf_a rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/pmc_actor_bf_a -- python3 $R/tools/actor_probe.py --math bf16x3 --launches 6 --discard 0
pass kt_actor_b1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_actor_b1 -- python3 $R/tools/actor_probe.py --math bf16 --launches 200
pass pmc_actor_b1_a rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/pmc_actor_b1_a -- python3 $R/tools/actor_probe.py --math bf16 --launches 6 --discard 0
# ---- the mixed trajectory set (BASELINE config 5's workload): instruction count per wave-step
pass valu_a_mixed rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/valu_a_mixed -- python3 $R/bench.py $PMCARGS --workload mixed --carry f64
pass instbench $R/tools/instbench --json
sha256sum $R/bench.py $R/mr_rl_amd/libmrsim.so > $O/sha_$part.txt
(cd $O && find . -type f ! -name "manifest*.txt" | sed 's|^\./||' | sort) > $O/manifest_$part.txt   # what THIS invocation wrote
cat $O/status_$part.txt
echo profile_round $part done
