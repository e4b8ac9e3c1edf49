R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run40
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_fitting_gpu.py tests/test_e2e_gpu.py -q -x > $OUT/pytest_fit.txt 2>&1; tail -3 $OUT/pytest_fit.txt | cut -c1-200
timeout 300 python tools/stage_lbfgs_n.py 1 10 2>&1 | tail -1
timeout 300 python tools/stage_lbfgs_n.py 2 10 2>&1 | tail -1
timeout 300 python tools/lbfgs_eval_breakdown.py 2>&1 | tail -2 | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o t -- python $R/tools/stage_lbfgs_n.py 2 10 > $OUT/p.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
head -12 $(find $OUT/p -name '*kernel_stats.csv') | cut -c1-130
