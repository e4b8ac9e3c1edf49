R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run17
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt | cut -c1-300
timeout 900 python bench.py --no-cpu-baseline --no-c5 --no-rccl-check > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_run17/bench.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','closure_mode','lbfgs'):
    print(k, json.dumps(d.get(k))[:2000])
PY
tail -2 $OUT/bench.err | cut -c1-300
timeout 300 python tools/closure_ops.py > $OUT/closure_ops.txt 2>&1; tail -12 $OUT/closure_ops.txt | cut -c1-200
