R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run21
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_humor_loss_gpu.py -q -x -s > $OUT/pytest_humor_loss.txt 2>&1; tail -12 $OUT/pytest_humor_loss.txt | cut -c1-600
