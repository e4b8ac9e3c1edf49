R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run29
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 300 python tools/smpl_dense_bwd_timing.py 1920 22 > $OUT/dense_bwd_1920.txt 2>&1; tail -9 $OUT/dense_bwd_1920.txt | cut -c1-250
timeout 300 python tools/smpl_dense_bwd_timing.py 30720 22 > $OUT/dense_bwd_30720.txt 2>&1; tail -9 $OUT/dense_bwd_30720.txt | cut -c1-250
timeout 600 python -m pytest tests/test_rollout_gpu.py -q -x -k "groups or accumulate" 2>&1 | tail -3
