# round 5, GPU session 2: phase timestamps of the pipelined forward; the c2 short-run test again
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_run2
mkdir -p $OUT
cd $R
HUMOR_AMD_LIB=$R/tools/microbench/libhumor_amd_ptiming.so timeout 300 python tools/pipe_phase_timing.py 256 30 > $OUT/phase_256.txt 2>&1; cat $OUT/phase_256.txt
HUMOR_AMD_LIB=$R/tools/microbench/libhumor_amd_ptiming.so timeout 300 python tools/pipe_phase_timing.py 64 30 > $OUT/phase_64.txt 2>&1; cat $OUT/phase_64.txt
timeout 600 python -m pytest tests/test_fitting_gpu.py -x -q -k "short_run_at_baseline_sizes or earlier_persistent_failure" > $OUT/pytest_fit.txt 2>&1; tail -8 $OUT/pytest_fit.txt
