R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run10
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|FAILED\|rollout post\|fit pre" $OUT/pytest_gpu.txt | cut -c1-300
timeout 900 python bench.py --no-cpu-baseline --no-c5 --no-rccl-check > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_run10/bench.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','closure_mode','lbfgs'):
    print(k, json.dumps(d.get(k))[:1800])
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-c5 --eager --no-lbfgs --no-rccl-check > $OUT/prof_stdout.txt 2> $OUT/prof_stderr.txt
find $OUT -name "*.db" -delete
rm -f $OUT/prof/*kernel_trace.csv
