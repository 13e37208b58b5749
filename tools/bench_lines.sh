# the bench lines committed under profiles/<tag>/ beside the profiler's (one box, one after another):  bash tools/bench_lines.sh
O=$GRAFT_REPO_ROOT/gpurun_out/lines; mkdir -p $O; cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 300 "$@" > $O/$name.json 2> $O/$name.err; echo "$name $?"; }
run bench_default python bench.py &&
run bench_driver_flags python bench.py --gpus 1 --steps 20 --warmup 5 &&
run bench_mismatched python bench.py --mismatched &&
run bench_mixed python bench.py --workload mixed &&
run bench_gpus2_gloo_rehearsal python bench.py --gpus 2 --share-gpu --dist-backend gloo --steps 510 --warmup 102 &&
MRSIM_BENCH_FORCE_DIST=1 run bench_one_rank_rccl python bench.py --dist-backend nccl
