R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run28
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/p10 -o t -- python $R/tools/closure_n.py 10 > $OUT/p10.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/p30 -o t -- python $R/tools/closure_n.py 30 > $OUT/p30.txt 2>&1
find $OUT -name "*.db" -delete
cd $R
python tools/closure_trace_diff.py $(find $OUT/p10 -name '*kernel_trace.csv') 10 $(find $OUT/p30 -name '*kernel_trace.csv') 30 > $OUT/closure_census.txt 2>&1
cat $OUT/closure_census.txt | cut -c1-190
find $OUT -name '*kernel_trace.csv' -delete
