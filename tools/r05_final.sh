# end-of-round GPU session: artefacts (tests, smoke, bench, rocprofv3 of the bench), MFMA PMC pass of the C5 workload, pipelined-kernel timings
R=$GRAFT_REPO_ROOT
bash $R/tools/artefacts.sh r05_final
OUT=$R/gpurun_out/r05_final
cd $R
timeout 900 bash tools/pmc_mfma.sh > $OUT/pmc_mfma.log 2>&1
mkdir -p $OUT/pmc_mfma && cp -r $R/gpurun_out/pmc_mfma/run $OUT/pmc_mfma/ 2>/dev/null
python tools/pmc_mfma_summary.py $R/gpurun_out/pmc_mfma/run > $OUT/pmc_mfma_summary.txt 2>&1; grep -i "pipe\|prior_gemm" $OUT/pmc_mfma_summary.txt | cut -c1-260
for i in 1 2; do timeout 300 python tools/pipe_debug.py time 256 119 2>&1 | grep "pipe fwd\|chain  " | tee -a $OUT/pipe_time.txt; done
timeout 300 python tools/pipe_debug.py time 64 59 2>&1 | grep "pipe fwd\|chain  " | tee -a $OUT/pipe_time.txt
