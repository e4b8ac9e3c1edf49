# round 2, GPU session 3: fused fit loss -- tests, bench, kernel census of the closure
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run3
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.txt 2>&1; grep -n "short run\|passed\|failed\|FAILED\|^c[2345] \|fused fit" $OUT/pytest_gpu.txt | cut -c1-600
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-c5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 1700 $OUT/bench.json | head -c 900; tail -3 $OUT/bench.err | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-c5 --eager > $OUT/prof_stdout.txt 2> $OUT/prof_stderr.txt
find $OUT -name "*.db" -delete
rm -f $OUT/prof/*kernel_trace.csv
ls $OUT/prof
