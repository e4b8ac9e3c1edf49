# PMC passes for the LBS skinning kernel at the metric's batch (N = 1920), launches cycling over 4 operand sets (1.27 GB > the 256 MiB
# Infinity Cache): HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes), then the SQ / LDS counters.  Counter passes carry
# --kernel-trace only.   usage (on the GPU box): bash tools/pmc_lbs.sh [outdir] [sets]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc_lbs}
SETS=${2:-4}
rm -rf $OUT && mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python $R/tools/skin_once.py -1 1920 $SETS > $OUT/$c.log 2>&1
done
timeout 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/SQ -- python $R/tools/skin_once.py -1 1920 $SETS > $OUT/SQ.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/LDS -- python $R/tools/skin_once.py -1 1920 $SETS > $OUT/LDS.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/VMEM -- python $R/tools/skin_once.py -1 1920 $SETS > $OUT/VMEM.log 2>&1
find $OUT -name "*.db" -delete
python $R/tools/pmc_lbs_summary.py $OUT > $OUT/SUMMARY.txt 2>&1
cat $OUT/SUMMARY.txt
