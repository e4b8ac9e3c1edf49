# round 2, GPU session 6: full GPU tests + full bench (incl. lbfgs, rccl self-check, c5, cpu baseline) + 2-rank gloo bench path
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run6
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|FAILED\|chamfer\|near pi" $OUT/pytest_gpu.txt | cut -c1-300
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_run6/bench.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','closure_mode','lbfgs','rccl','cpu_baseline','roofline'):
    print(k, json.dumps(d.get(k))[:900])
print('c5', json.dumps(d.get('c5_rooflines'))[:900])
PY
tail -3 $OUT/bench.err | cut -c1-300
HUMOR_AMD_BENCH_BACKEND=gloo HUMOR_AMD_BENCH_ONE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err; tail -c 1200 $OUT/bench_2rank_gloo.json; tail -3 $OUT/bench_2rank_gloo.err | cut -c1-300
