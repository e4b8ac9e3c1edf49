R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run30
rm -rf $OUT && mkdir -p $OUT
cd $R
WAVES=6400,12800,25600,51200,102400 timeout 300 python tools/smpl_dense_bwd_timing.py 30720 22 > $OUT/dense_bwd_30720.txt 2>&1; tail -6 $OUT/dense_bwd_30720.txt | cut -c1-250
WAVES=6400,12800,25600 timeout 300 python tools/smpl_dense_bwd_timing.py 7680 22 > $OUT/dense_bwd_7680.txt 2>&1; tail -4 $OUT/dense_bwd_7680.txt | cut -c1-250
cd /tmp && export TMPDIR=/tmp
WAVES=25600 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o d -- python $R/tools/smpl_dense_bwd_timing.py 30720 22 > $OUT/prof.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
head -12 $(find $OUT/prof -name '*kernel_stats.csv') | cut -c1-180
