# round 2, GPU session 4: where the decoder-chain launches spend their time
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run4
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 300 python tools/layer_timing.py > $OUT/layer_timing.txt 2>&1; tail -50 $OUT/layer_timing.txt
timeout 300 python tools/rollout_ab.py 32 59 "4,1" "6,1" "8,1" "12,1" "17,1" > $OUT/rollout_ab_32.txt 2>&1; cat $OUT/rollout_ab_32.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o ro -- python $R/tools/rollout_ab.py 32 59 "4,1" > $OUT/prof_stdout.txt 2> $OUT/prof_stderr.txt
find $OUT -name "*.db" -delete
python $R/tools/dispatch_by_grid.py $OUT/prof/*kernel_trace.csv ha:: > $OUT/by_grid.txt 2>&1; cat $OUT/by_grid.txt
rm -f $OUT/prof/*kernel_trace.csv
