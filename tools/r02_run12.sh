R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run12
rm -rf $OUT && mkdir -p $OUT
cd $R
SKIN_VARIANTS=5,4,6,1,0,2,13,14,21 timeout 600 python tools/skin_ab.py 7680 30720 > $OUT/skin_ab.txt 2>&1; cat $OUT/skin_ab.txt
cd /tmp && export TMPDIR=/tmp
for N in 1920 30720; do
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$N/$c -- python $R/tools/skin_once.py -1 $N > $OUT/pmc_${N}_$c.log 2>&1
done
done
find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -delete
ls -R $OUT | head -40
