R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run33
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_fitting_gpu.py tests/test_e2e_gpu.py -q -x > $OUT/pytest_fit.txt 2>&1; tail -6 $OUT/pytest_fit.txt | cut -c1-300
timeout 300 python tools/lbfgs_eval_breakdown.py > $OUT/lbfgs_eval_breakdown.txt 2>&1; tail -4 $OUT/lbfgs_eval_breakdown.txt | cut -c1-400
timeout 600 python bench.py --no-cpu-baseline --no-c5 --no-rccl-check > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_run33/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'])
lb=d['lbfgs']
for k,v in lb['phases'].items(): print(k, v)
print('whole fit', lb['whole_fit_seconds_for_30_80_70_schedule'])
PY
tail -2 $OUT/bench.err | cut -c1-200
