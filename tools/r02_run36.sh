R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run36
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_smpl_gpu.py -q -x 2>&1 | tail -2
for i in 1 2; do
HUMOR_AMD_LIB=$R/tools/microbench/libhumor_amd_prev.so timeout 300 python tools/dense_fwd_timing.py 2>&1 | grep "N=" | sed 's/^/prev: /'
timeout 300 python tools/dense_fwd_timing.py 2>&1 | grep "N=" | sed 's/^/new:  /'
done
