R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run8
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.txt 2>&1; grep -n "short run\|passed\|failed\|FAILED" $OUT/pytest_gpu.txt | cut -c1-700
timeout 900 python bench.py --no-cpu-baseline --no-c5 > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_run8/bench.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','closure_mode','lbfgs','rccl'):
    print(k, json.dumps(d.get(k))[:1800])
PY
tail -3 $OUT/bench.err | cut -c1-300
