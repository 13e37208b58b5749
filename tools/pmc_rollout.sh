#!/bin/bash
# PMC instruction-mix passes for the rollout kernel (run on the GPU box via gpurun).  Usage: tools/pmc_rollout.sh <tag> [bench args]
R=$GRAFT_REPO_ROOT; tag=$1; shift
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d $R/gpurun_out/pmc_${tag}_a -- python3 $R/bench.py --no-cpu-baseline --steps 102 "$@" > $R/gpurun_out/pmc_${tag}_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_${tag}_b -- python3 $R/bench.py --no-cpu-baseline --steps 102 "$@" > $R/gpurun_out/pmc_${tag}_b.log 2>&1
echo done
