R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run34
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 300 python tools/rollout_ab.py 32 59 4,1,1,0,0,0 4,1,1,0,0,1 4,1,1,0,0,0 4,1,1,0,0,1 > $OUT/rollout_ab_32.txt 2>&1; grep "B=" $OUT/rollout_ab_32.txt | cut -c1-220
timeout 300 python tools/rollout_ab.py 256 119 4,1,1,0,0,0 4,1,1,0,0,1 > $OUT/rollout_ab_256.txt 2>&1; grep "B=" $OUT/rollout_ab_256.txt | cut -c1-220
timeout 300 python tools/rollout_ab.py 64 59 4,1,1,0,0,0 4,1,1,0,0,1 > $OUT/rollout_ab_64.txt 2>&1; grep "B=" $OUT/rollout_ab_64.txt | cut -c1-220
timeout 900 python -m pytest tests/test_rollout_gpu.py -q -x > $OUT/pytest_rollout.txt 2>&1; tail -4 $OUT/pytest_rollout.txt | cut -c1-300
