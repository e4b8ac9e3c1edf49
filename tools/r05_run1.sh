# round 5, GPU session 1: first runs of the pipelined forward (tools/pipe_debug.py) + the two new fitting tests
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_run1
mkdir -p $OUT
cd $R
export HUMOR_AMD_LIB=$R/tools/microbench/libhumor_amd_pdebug.so
timeout 300 python tools/pipe_debug.py fwd 40 3 > $OUT/fwd_40_3.txt 2>&1; tail -40 $OUT/fwd_40_3.txt
timeout 200 python tools/pipe_debug.py fwd 256 5 > $OUT/fwd_256_5.txt 2>&1; tail -30 $OUT/fwd_256_5.txt
unset HUMOR_AMD_LIB
timeout 200 python tools/pipe_debug.py grad 64 6 > $OUT/grad_64_6.txt 2>&1; tail -12 $OUT/grad_64_6.txt
timeout 300 python tools/pipe_debug.py time 256 119 > $OUT/time_256_119.txt 2>&1; tail -6 $OUT/time_256_119.txt
timeout 300 python tools/pipe_debug.py time 64 59 > $OUT/time_64_59.txt 2>&1; tail -6 $OUT/time_64_59.txt
timeout 600 python -m pytest tests/test_fitting_gpu.py -x -q -k "short_run_at_baseline_sizes or earlier_persistent_failure" > $OUT/pytest_fit.txt 2>&1; tail -8 $OUT/pytest_fit.txt
