R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run25
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 300 python tools/lbfgs_eval_breakdown.py > $OUT/lbfgs_eval_breakdown.txt 2>&1; tail -5 $OUT/lbfgs_eval_breakdown.txt | cut -c1-400
