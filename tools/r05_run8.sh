# round 5, GPU session 7: pipelined adjoint, deeper prefetch on the glue CU
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_run8
mkdir -p $OUT
cd $R
timeout 200 python tools/pipe_debug.py grad 64 6 > $OUT/grad_64_6.txt 2>&1; tail -9 $OUT/grad_64_6.txt
timeout 200 python tools/pipe_debug.py grad 256 9 > $OUT/grad_256_9.txt 2>&1; tail -9 $OUT/grad_256_9.txt
timeout 300 python tools/pipe_debug.py time 256 119 > $OUT/time_256_119.txt 2>&1; tail -5 $OUT/time_256_119.txt
timeout 300 python tools/pipe_debug.py time 64 59 > $OUT/time_64_59.txt 2>&1; tail -5 $OUT/time_64_59.txt
HUMOR_AMD_LIB=$R/tools/microbench/libhumor_amd_ptiming.so timeout 300 python tools/pipe_phase_timing.py 256 30 bwd > $OUT/phase_bwd_256.txt 2>&1; cat $OUT/phase_bwd_256.txt
