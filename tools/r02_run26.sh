R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run26
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 300 python tools/rollout_ab.py 256 119 4,1,1,0,1 4,1,1,0,2 4,1,1,0,4 4,1,1,0,8 > $OUT/rollout_ab_256.txt 2>&1; grep "B=" $OUT/rollout_ab_256.txt | cut -c1-200
timeout 300 python tools/rollout_ab.py 64 59 4,1,1,0,1 4,1,1,0,2 > $OUT/rollout_ab_64.txt 2>&1; grep "B=" $OUT/rollout_ab_64.txt | cut -c1-200
timeout 300 python tools/rollout_ab.py 128 59 4,1,1,0,1 4,1,1,0,2 4,1,1,0,4 > $OUT/rollout_ab_128.txt 2>&1; grep "B=" $OUT/rollout_ab_128.txt | cut -c1-200
timeout 900 python -m pytest tests/test_rollout_gpu.py -q -x > $OUT/pytest_rollout.txt 2>&1; tail -5 $OUT/pytest_rollout.txt | cut -c1-300
