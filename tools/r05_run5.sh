# round 5, GPU session 5: dedicated glue CU, layer 0 on five CUs
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_run5
mkdir -p $OUT
cd $R
timeout 300 python tools/pipe_debug.py fwd 40 3 > $OUT/fwd_40_3.txt 2>&1; head -8 $OUT/fwd_40_3.txt
timeout 200 python tools/pipe_debug.py fwd 256 7 > $OUT/fwd_256_7.txt 2>&1; head -8 $OUT/fwd_256_7.txt
timeout 200 python tools/pipe_debug.py fwd 100 4 > $OUT/fwd_100_4.txt 2>&1; head -8 $OUT/fwd_100_4.txt
timeout 200 python tools/pipe_debug.py grad 64 6 > $OUT/grad_64_6.txt 2>&1; tail -6 $OUT/grad_64_6.txt
timeout 300 python tools/pipe_debug.py time 256 119 > $OUT/time_256_119.txt 2>&1; tail -5 $OUT/time_256_119.txt
timeout 300 python tools/pipe_debug.py time 64 59 > $OUT/time_64_59.txt 2>&1; tail -5 $OUT/time_64_59.txt
HUMOR_AMD_LIB=$R/tools/microbench/libhumor_amd_ptiming.so timeout 300 python tools/pipe_phase_timing.py 256 30 > $OUT/phase_256.txt 2>&1; cat $OUT/phase_256.txt
HUMOR_AMD_LIB=$R/tools/microbench/libhumor_amd_ptiming.so timeout 300 python tools/pipe_phase_timing.py 64 30 > $OUT/phase_64.txt 2>&1; cat $OUT/phase_64.txt
