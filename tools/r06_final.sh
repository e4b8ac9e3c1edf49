# end-of-round GPU session on the HEAD the round ends on: artefacts (GPU tests under NaN poison, smoke, bench, rocprofv3 of the bench) + the second
# poison mode of the GPU tier (1e30 instead of NaN)
R=$GRAFT_REPO_ROOT
bash $R/tools/artefacts.sh r06_final
OUT=$R/gpurun_out/r06_final
cd $R
t0=$(date +%s)
HUMOR_AMD_TEST_POISON_VALUE=big timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu_big.txt 2>&1; tail -3 $OUT/pytest_gpu_big.txt | cut -c1-200
echo "pytest (1e30 poison): $(( $(date +%s) - t0 )) s"
