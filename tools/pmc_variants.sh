#!/bin/bash
# SQ counters of the fused rollout kernel for A/B variants (run on the GPU box via gpurun):
#   tools/pmc_variants.sh <outdir-tag> [--mismatched] -- variant[:f64] ...
# One rocprofv3 --pmc pass per variant and counter group (the guide: counters in their own runs, no trace flags).
R=$GRAFT_REPO_ROOT; tag=$1; shift
extra=""
while [ "$1" != "--" ]; do extra="$extra $1"; shift; done; shift
O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  vv=${v/:/_}
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/a_$vv -- python3 $R/tools/ab_rollout.py --rounds 1 --launches 6 --discard 0 $extra $v > $O/a_$vv.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VMEM_WR --output-format csv -d $O/b_$vv -- python3 $R/tools/ab_rollout.py --rounds 1 --launches 6 --discard 0 $extra $v > $O/b_$vv.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD --output-format csv -d $O/c_$vv -- python3 $R/tools/ab_rollout.py --rounds 1 --launches 6 --discard 0 $extra $v > $O/c_$vv.log 2>&1
done
python3 $R/tools/pmc_variants_summary.py $O "$@"
