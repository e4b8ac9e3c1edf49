R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run18
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt | cut -c1-300
timeout 300 python tools/smpl_dense_bwd_timing.py 1920 22 > $OUT/dense_bwd_1920.txt 2>&1; tail -9 $OUT/dense_bwd_1920.txt | cut -c1-250
timeout 300 python tools/smpl_dense_bwd_timing.py 1920 52 > $OUT/dense_bwd_1920_hands.txt 2>&1; tail -9 $OUT/dense_bwd_1920_hands.txt | cut -c1-250
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_dense -o dn -- python $R/tools/smpl_dense_bwd_timing.py 1920 22 > $OUT/prof_stdout.txt 2>&1; cd $R
f=$(find $OUT/prof_dense -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-220
timeout 900 python bench.py --no-cpu-baseline --no-c5 --no-rccl-check > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_run18/bench.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','closure_mode','lbfgs'):
    print(k, json.dumps(d.get(k))[:2500])
PY
tail -2 $OUT/bench.err | cut -c1-300
timeout 300 python tools/closure_ops.py > $OUT/closure_ops.txt 2>&1; grep -A8 "forward dispatches" $OUT/closure_ops.txt | cut -c1-200
