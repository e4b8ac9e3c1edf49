# rocprofv3 kernel stats of the default bench run (hipGraph closure) -> gpurun_out/prof_bench/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_bench
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_stdout.txt 2> $OUT/bench_stderr.txt
find $OUT -name "*.db" -delete
rm -f $OUT/*/*kernel_trace.csv $OUT/*kernel_trace.csv
ls -la $OUT $OUT/* | head -20
tail -1 $OUT/bench_stdout.txt | cut -c1-200
