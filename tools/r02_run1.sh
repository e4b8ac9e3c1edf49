# round 2, GPU session 1: tests + roll-out timing after taking the prior off the recurrence
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run1
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; tail -15 $OUT/pytest_gpu.txt
timeout 300 python tools/rollout_ab.py 32 59 "4,1" > $OUT/rollout_ab_32.txt 2>&1; cat $OUT/rollout_ab_32.txt
timeout 300 python tools/rollout_ab.py 256 119 "4,1" > $OUT/rollout_ab_256.txt 2>&1; cat $OUT/rollout_ab_256.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-c5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o ro -- python $R/tools/rollout_ab.py 32 59 "4,1" > $OUT/prof_stdout.txt 2> $OUT/prof_stderr.txt
find $OUT -name "*.db" -delete
rm -f $OUT/prof/*kernel_trace.csv
head -12 $OUT/prof/*kernel_stats.csv | cut -c1-200
ls $OUT $OUT/prof
