cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc6
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc6/$c -- python $R/tools/skin_once.py -1 1920 > $R/gpurun_out/pmc6/$c.log 2>&1
done
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc6/SQ -- python $R/tools/skin_once.py -1 1920 > $R/gpurun_out/pmc6/SQ.log 2>&1
find $R/gpurun_out/pmc6 -name "*.db" -delete
ls -R $R/gpurun_out/pmc6 | head -30
