# The measurements round 3 left open, in one GPU call (gpurun --timeout 900 -- 'bash tools/next_round_first.sh'):
#   1. dense backward with the chunk-compressed dL/dA (ha_tune_set("dense_gA_sparse", 2)) against the default and the dense product
#   2. closure census + launch list of the final code
#   3. phase timestamps of the SMPL frame kernels and of the persistent roll-out (variant builds must exist: tools/build_variant.sh
#      stiming -DHA_SMPL_TIMING ; tools/build_variant.sh ptiming -DHA_PERSIST_TIMING)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/next_round_first
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 1920 30720; do timeout 250 python $R/tools/smpl_dense_bwd_timing.py $n 2>&1 | tee -a $OUT/dense_bwd.txt | tail -6; done
timeout 200 python $R/tools/closure_launch_list.py > $OUT/launch_list.txt 2>&1; tail -3 $OUT/launch_list.txt
cd $R && bash tools/closure_census.sh next_round_census > /dev/null 2>&1; head -12 $R/gpurun_out/next_round_census/closure_census.txt
[ -f tools/microbench/libhumor_amd_stiming.so ] && for n in 32 1920; do HUMOR_AMD_LIB=tools/microbench/libhumor_amd_stiming.so timeout 100 python tools/smpl_phase_timing.py $n | tee -a $OUT/smpl_phase.txt; done
[ -f tools/microbench/libhumor_amd_ptiming.so ] && HUMOR_AMD_LIB=tools/microbench/libhumor_amd_ptiming.so timeout 200 python tools/persist_phase_timing.py 1 > $OUT/persist_phase.txt 2>&1
