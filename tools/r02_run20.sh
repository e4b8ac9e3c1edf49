R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run20
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 600 python tools/skin_jitter.py > $OUT/skin_jitter.txt 2>&1; grep -v "^\[" $OUT/skin_jitter.txt | cut -c1-330 | tail -24
HUMOR_AMD_BENCH_BACKEND=gloo HUMOR_AMD_BENCH_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err; tail -c 1500 $OUT/bench_2rank_gloo.json; tail -3 $OUT/bench_2rank_gloo.err | cut -c1-300
