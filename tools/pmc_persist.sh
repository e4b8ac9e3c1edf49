# MFMA-side PMC counters of the B <= 32 persistent roll-out kernels at the metric's batch (tools/persist_timing.py quick = 32 x 59)
# -> gpurun_out/pmc_persist/ ; summary with tools/pmc_mfma_summary.py.   usage (on the GPU box): bash tools/pmc_persist.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_persist
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/run -- python $R/tools/persist_timing.py quick > $OUT/run.log 2>&1
find $OUT -name "*.db" -delete
python $R/tools/pmc_mfma_summary.py $OUT/run > $OUT/SUMMARY.txt 2>&1
rm -f $OUT/run/*/*kernel_trace.csv
grep -i "persist\|prior_gemm" $OUT/SUMMARY.txt | cut -c1-300
