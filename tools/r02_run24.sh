R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run24
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 300 python tools/lbfgs_eval_breakdown.py > $OUT/lbfgs_eval_breakdown.txt 2>&1; tail -14 $OUT/lbfgs_eval_breakdown.txt | cut -c1-250
timeout 300 python tools/closure_ops.py > $OUT/closure_ops.txt 2>&1; tail -45 $OUT/closure_ops.txt | cut -c1-200
timeout 300 python tools/closure_hostprof.py > $OUT/closure_hostprof.txt 2>&1; grep -A34 "tottime" $OUT/closure_hostprof.txt | head -40 | cut -c1-200
timeout 600 python -m pytest tests/test_rollout_gpu.py -q -x -k "accumulate or determinism" 2>&1 | tail -3
