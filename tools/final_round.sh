# End-of-round measurement session on one MI355X: tests, smoke, bench (+CPU baseline), rocprofv3 stats, C5 sizes, phase timing.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final
rm -rf $OUT && mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json
python tools/bench_c5.py > $OUT/bench_c5.txt 2>&1; tail -12 $OUT/bench_c5.txt
python tools/layer_timing.py > $OUT/layer_timing.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/prof_bench_stdout.txt 2> $OUT/prof_bench_stderr.txt
find $OUT -name "*.db" -delete
rm -f $OUT/prof/*kernel_trace.csv
ls $OUT $OUT/prof
