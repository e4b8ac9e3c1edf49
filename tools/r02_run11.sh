R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run11
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_rollout_gpu.py -m gpu -q -x > $OUT/pytest_rollout.txt 2>&1; tail -3 $OUT/pytest_rollout.txt | cut -c1-300
timeout 300 python tools/rollout_ab.py 32 59 "4,1,0" "4,1,1" "4,1,0" "4,1,1" > $OUT/rollout_ab_32.txt 2>&1; cat $OUT/rollout_ab_32.txt
