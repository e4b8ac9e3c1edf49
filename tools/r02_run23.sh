R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run23
rm -rf $OUT && mkdir -p $OUT
cd $R
ACC=0 timeout 300 python tools/layer_timing.py > $OUT/layer_timing_acc0.txt 2>&1; tail -34 $OUT/layer_timing_acc0.txt
ACC=1 timeout 300 python tools/layer_timing.py > $OUT/layer_timing_acc1.txt 2>&1; tail -34 $OUT/layer_timing_acc1.txt
