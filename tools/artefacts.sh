# Artefact session of a round: GPU tests, smoke, bench (+CPU baseline, parity), rocprofv3 --kernel-trace --stats of the same bench command.
# usage (on the GPU box, via gpurun):  bash tools/artefacts.sh r03_final [skip-tests]
R=$GRAFT_REPO_ROOT
RUN=${1:-r04_mid}
OUT=$R/gpurun_out/$RUN
rm -rf $OUT && mkdir -p $OUT
cd $R
t0=$(date +%s)
if [ "$2" != "skip-tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 > $OUT/pytest_gpu.txt 2>&1; tail -22 $OUT/pytest_gpu.txt | cut -c1-200
  echo "pytest: $(( $(date +%s) - t0 )) s"; t0=$(date +%s)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
  echo "smoke: $(( $(date +%s) - t0 )) s"; t0=$(date +%s)
fi
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; head -c 2500 $OUT/bench.json; echo; tail -2 $OUT/bench.err | cut -c1-300
echo "bench: $(( $(date +%s) - t0 )) s"; t0=$(date +%s)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-lbfgs --no-rccl-check > $OUT/prof_bench_stdout.txt 2> $OUT/prof_bench_stderr.txt
echo "rocprof: $(( $(date +%s) - t0 )) s"
find $OUT -name "*.db" -delete
find $OUT/prof -name "*kernel_trace.csv" -delete
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/bench_kernel_stats.csv && head -16 "$f" | cut -c1-180
