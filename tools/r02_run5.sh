R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run5
rm -rf $OUT && mkdir -p $OUT
cd $R
for v in 0 1; do
echo "HIP_FORCE_DEV_KERNARG=$v"
HIP_FORCE_DEV_KERNARG=$v timeout 300 python tools/rollout_ab.py 32 59 "4,1" 2>&1 | grep fwd
HIP_FORCE_DEV_KERNARG=$v B=32 timeout 300 python tools/layer_timing.py 2>&1 | tail -9
done
