# Round-2 artefact session: GPU tests, smoke, bench (+CPU baseline), rocprofv3 --kernel-trace --stats of the same bench command.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_final
rm -rf $OUT && mkdir -p $OUT
cd $R
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $OUT/pytest_gpu.txt 2>&1; tail -25 $OUT/pytest_gpu.txt | cut -c1-200
echo "pytest: $(( $(date +%s) - t0 )) s"; t0=$(date +%s)
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
echo "smoke: $(( $(date +%s) - t0 )) s"; t0=$(date +%s)
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json; tail -3 $OUT/bench.err | cut -c1-300
echo "bench: $(( $(date +%s) - t0 )) s"; t0=$(date +%s)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-lbfgs --no-rccl-check > $OUT/prof_bench_stdout.txt 2> $OUT/prof_bench_stderr.txt
echo "rocprof: $(( $(date +%s) - t0 )) s"
find $OUT -name "*.db" -delete
find $OUT/prof -name "*kernel_trace.csv" -delete
find $OUT/prof -type f | head; f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-200
