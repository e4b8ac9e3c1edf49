# Round-4 GPU session 1 (no code changes): MFMA PMC evidence for the round-3 kernels + the phase tables of the persistent kernels.
#   gpurun --timeout 900 -- 'bash tools/r04_run2.sh'
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_run2
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
t0=$(date +%s)
PMC="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA"
# C4 (the metric's batch: 32 x 60): ten stage-3 closures -> rollout_persist_fwd/bwd, prior_gemm, the VPoser GEMMs
timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/c4 -- python $R/tools/closure_n.py 10 > $OUT/c4.log 2>&1
echo "pmc c4: $(( $(date +%s) - t0 )) s"; t0=$(date +%s)
# C5 (256 x 120): pose_blend_mfma, prior_gemm at M = 30 464, mlp_layer
timeout 400 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/c5 -- python $R/tools/bench_c5.py > $OUT/c5.log 2>&1
echo "pmc c5: $(( $(date +%s) - t0 )) s"; t0=$(date +%s)
python $R/tools/pmc_mfma_summary.py $OUT/c4 $OUT/c5 > $OUT/SUMMARY.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name '*kernel_trace.csv' -delete
# the raw counter CSVs are large: keep only the rows of the kernels that issue MFMAs
for d in c4 c5; do
  for f in $(find $OUT/$d -name '*counter_collection.csv'); do
    (head -1 $f; grep -E "rollout_persist|prior_gemm|pose_blend|mlp_layer|dense_g" $f) > $OUT/${d}_mfma_kernels_counter_collection.csv; rm -f $f
  done
done
cat $OUT/SUMMARY.txt | cut -c1-260
cd $R
HUMOR_AMD_LIB=tools/microbench/libhumor_amd_ptiming.so timeout 200 python tools/persist_phase_timing.py 1 > $OUT/persist_phase.txt 2>&1
tail -32 $OUT/persist_phase.txt
echo "phase: $(( $(date +%s) - t0 )) s"
