# round 5, GPU session 10: A/B of the early first poll (forward roles), same box, alternating
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_run10
mkdir -p $OUT
cd $R
for i in 1 2 3; do
timeout 300 python tools/pipe_debug.py time 256 119 2>&1 | grep "pipe fwd + pipe bwd" | sed 's/^/prefetch    /' | tee -a $OUT/ab.txt
HUMOR_AMD_LIB=$R/tools/microbench/libhumor_amd_nopf.so timeout 300 python tools/pipe_debug.py time 256 119 2>&1 | grep "pipe fwd + pipe bwd" | sed 's/^/no prefetch /' | tee -a $OUT/ab.txt
done
