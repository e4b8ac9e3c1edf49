R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run31
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 300 python tools/skin_sustained.py 30720 > $OUT/skin_sustained.txt 2>&1; grep -v amdgpu.ids $OUT/skin_sustained.txt | cut -c1-220
timeout 300 python tools/smpl_dense_bwd_timing.py 30720 22 2>&1 | grep "N=" | cut -c1-200
