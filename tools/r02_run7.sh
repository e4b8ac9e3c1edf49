R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run7
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "rotation or points3d or stitched or all_phases or amass_style" > $OUT/pytest_sel.txt 2>&1; tail -5 $OUT/pytest_sel.txt | cut -c1-300
timeout 600 python tools/stage_closures.py > $OUT/stage_closures.txt 2>&1; grep graphs $OUT/stage_closures.txt
