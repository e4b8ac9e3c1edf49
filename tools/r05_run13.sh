# round 5, GPU session 13: fitting tests with the branch sets, dense SMPL backward variants (+ kernel stats), MFMA PMC passes at C5 (pipelined kernels)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_run13
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_fitting_gpu.py -q > $OUT/pytest_fitting.txt 2>&1; tail -4 $OUT/pytest_fitting.txt
for n in 1920 30720; do timeout 250 python tools/smpl_dense_bwd_timing.py $n 2>&1 | grep -v amdgpu | tee -a $OUT/dense_bwd.txt | tail -7; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dense -- python $R/tools/smpl_dense_bwd_timing.py 1920 > $OUT/prof_dense.log 2>&1
find $OUT/prof_dense -name "*.db" -delete; find $OUT/prof_dense -name "*kernel_trace.csv" -delete
head -12 $(find $OUT/prof_dense -name "*kernel_stats.csv" | head -1) | cut -c1-200
cd $R
timeout 900 bash tools/pmc_mfma.sh > $OUT/pmc_mfma.log 2>&1; tail -3 $OUT/pmc_mfma.log | cut -c1-400
mkdir -p $OUT/pmc_mfma && cp -r $R/gpurun_out/pmc_mfma/* $OUT/pmc_mfma/ 2>/dev/null
python tools/pmc_mfma_summary.py $R/gpurun_out/pmc_mfma/run > $OUT/pmc_mfma_summary.txt 2>&1; grep -i "pipe\|mlp_layer\|prior_gemm\|persist" $OUT/pmc_mfma_summary.txt | cut -c1-300
