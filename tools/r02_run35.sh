R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run35
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_smpl_gpu.py -q -x 2>&1 | tail -2
HUMOR_AMD_LIB=$R/tools/microbench/libhumor_amd_prev.so timeout 300 python tools/smpl_frame_timing.py 1920 2>&1 | grep "N="
timeout 300 python tools/smpl_frame_timing.py 1920 2>&1 | grep "N="
HUMOR_AMD_LIB=$R/tools/microbench/libhumor_amd_prev.so timeout 300 python tools/smpl_frame_timing.py 1920 2>&1 | grep "N="
timeout 300 python tools/smpl_frame_timing.py 1920 2>&1 | grep "N="
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o f -- python $R/tools/smpl_frame_timing.py 1920 > $OUT/prof.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
head -4 $(find $OUT/prof -name '*kernel_stats.csv') | cut -c1-150
