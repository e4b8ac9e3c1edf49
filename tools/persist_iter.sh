# One iteration of the persistent roll-out work on the GPU box: parity of the persistent kernels (against the oracle at 32 x 59 and against
# the launch chain), then timing at the metric's batch, then (if the profiling build exists) the phase tables.
#   gpurun --timeout 600 -- 'bash tools/persist_iter.sh <tag> [full]'
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/persist_iter_${1:-x}
rm -rf $OUT && mkdir -p $OUT
cd $R
K="persistent and not 256 and not chain_bwd"
[ "$2" == "full" ] && K="persistent or rotation_representations or full_length"
timeout 500 python -m pytest tests/test_rollout_gpu.py -q -x -k "$K" > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt | cut -c1-250
timeout 200 python tools/persist_timing.py quick > $OUT/timing.txt 2>&1; grep -v "^{" $OUT/timing.txt | tail -4 | cut -c1-250
if [ -f tools/microbench/libhumor_amd_ptiming.so ]; then
  HUMOR_AMD_LIB=tools/microbench/libhumor_amd_ptiming.so timeout 200 python tools/persist_phase_timing.py 1 > $OUT/phase.txt 2>&1
  grep -E "median step|sum of medians" $OUT/phase.txt
fi
