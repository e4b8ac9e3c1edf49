R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run39
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 300 python tools/stage_lbfgs_n.py 2 10 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o t -- python $R/tools/stage_lbfgs_n.py 2 10 > $OUT/p.txt 2>&1
tail -1 $OUT/p.txt
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
head -40 $(find $OUT/p -name '*kernel_stats.csv') | cut -c1-150
