# round 5, GPU session 9: roll-out GPU tests on the pipelined paths; early first poll in the forward roles
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_run9
mkdir -p $OUT
cd $R
timeout 300 python tools/pipe_debug.py time 256 119 > $OUT/time_256_119.txt 2>&1; tail -4 $OUT/time_256_119.txt
timeout 300 python tools/pipe_debug.py time 64 59 > $OUT/time_64_59.txt 2>&1; tail -4 $OUT/time_64_59.txt
HUMOR_AMD_LIB=$R/tools/microbench/libhumor_amd_ptiming.so timeout 300 python tools/pipe_phase_timing.py 256 30 > $OUT/phase_256.txt 2>&1; cat $OUT/phase_256.txt
timeout 1200 python -m pytest tests/test_rollout_gpu.py -x -q --durations=8 > $OUT/pytest_rollout.txt 2>&1; tail -25 $OUT/pytest_rollout.txt
