# round 5, GPU session 14: batched k-block reductions in the pipelined kernels' publish; dense backward default (compressed dL/dA)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_run14
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_rollout_gpu.py -x -q -k "pipelined or forward_backward or c5_size" > $OUT/pytest_rollout.txt 2>&1; tail -3 $OUT/pytest_rollout.txt
timeout 600 python -m pytest tests/test_smpl_gpu.py -x -q > $OUT/pytest_smpl.txt 2>&1; tail -3 $OUT/pytest_smpl.txt
for i in 1 2; do timeout 300 python tools/pipe_debug.py time 256 119 2>&1 | grep "pipe fwd + pipe bwd" | tee -a $OUT/time.txt; done
timeout 300 python tools/pipe_debug.py time 64 59 2>&1 | grep "pipe fwd + pipe bwd" | tee -a $OUT/time.txt
HUMOR_AMD_LIB=$R/tools/microbench/libhumor_amd_ptiming.so timeout 300 python tools/pipe_phase_timing.py 256 30 > $OUT/phase_fwd_256.txt 2>&1; grep "role\|group 3" $OUT/phase_fwd_256.txt
HUMOR_AMD_LIB=$R/tools/microbench/libhumor_amd_ptiming.so timeout 300 python tools/pipe_phase_timing.py 256 30 bwd > $OUT/phase_bwd_256.txt 2>&1; grep "role\|group 3" $OUT/phase_bwd_256.txt
