# round 5, GPU session 15: no wait states in front of VGPR-weight MFMAs (pipelined kernels)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_run15
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_rollout_gpu.py -x -q -k "pipelined or forward_backward or c5_size" > $OUT/pytest_rollout.txt 2>&1; tail -3 $OUT/pytest_rollout.txt
for i in 1 2; do timeout 300 python tools/pipe_debug.py time 256 119 2>&1 | grep "pipe fwd + pipe bwd" | tee -a $OUT/time.txt; done
timeout 300 python tools/pipe_debug.py time 64 59 2>&1 | grep "pipe fwd + pipe bwd" | tee -a $OUT/time.txt
