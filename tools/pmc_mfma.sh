# MFMA-side PMC counters for the pose-blend GEMM and the roll-out layer kernel at BASELINE C5 sizes -> gpurun_out/pmc_mfma/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_mfma
rm -rf $OUT && mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $OUT/mfma_counters_available.txt
cat $OUT/mfma_counters_available.txt | head -20
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/run -- python $R/tools/bench_c5.py > $OUT/run.log 2>&1
find $OUT -name "*.db" -delete
rm -f $OUT/run/*/*kernel_trace.csv
ls $OUT/run/* | head; tail -2 $OUT/run.log | cut -c1-300
