# round 5, GPU session 11: L-BFGS with speculative issue (trajectory pins, phases), B <= 32 adjoint variant A/B, MFMA PMC passes
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_run11
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_fitting_gpu.py -x -q -k "lbfgs or short_run or graphed" > $OUT/pytest_lbfgs.txt 2>&1; tail -5 $OUT/pytest_lbfgs.txt
timeout 600 python tools/lbfgs_phase_profile.py 5 1 > $OUT/phase_profile_spec1.txt 2>&1; tail -12 $OUT/phase_profile_spec1.txt
timeout 600 python tools/lbfgs_phase_profile.py 5 0 > $OUT/phase_profile_spec0.txt 2>&1; tail -6 $OUT/phase_profile_spec0.txt
for i in 1 2; do
timeout 300 python tools/pipe_debug.py time 32 59 2>&1 | grep "pipe fwd + pipe bwd" | sed 's/^/default  /' | tee -a $OUT/ab_gnblate.txt
HUMOR_AMD_LIB=$R/tools/microbench/libhumor_amd_gnblate.so timeout 300 python tools/pipe_debug.py time 32 59 2>&1 | grep "pipe fwd + pipe bwd" | sed 's/^/gnb late /' | tee -a $OUT/ab_gnblate.txt
done
