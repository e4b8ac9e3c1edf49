R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run27
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 300 python tools/rollout_graph_groups.py 256 119 1 2 4 8 > $OUT/graph_groups_256.txt 2>&1; grep "B=" $OUT/graph_groups_256.txt | cut -c1-250
timeout 300 python tools/rollout_graph_groups.py 64 59 1 2 > $OUT/graph_groups_64.txt 2>&1; grep "B=" $OUT/graph_groups_64.txt | cut -c1-250
