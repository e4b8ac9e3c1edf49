cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_lbfgs_own
mkdir -p $O
python tools/lbfgs_own_work.py 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o own -- python $GRAFT_REPO_ROOT/tools/lbfgs_own_work.py > $O/stdout.txt 2>&1
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv; find $O -name "*.db" -delete; find $O/prof -name "*kernel_trace.csv" -delete
head -22 $O/kernel_stats.csv | cut -c1-200
