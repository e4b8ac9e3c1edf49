# round 5, GPU session 12: fitting GPU tests (speculation off by default), short-run amass with and without
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_run12
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_fitting_gpu.py -q > $OUT/pytest_fitting.txt 2>&1; tail -15 $OUT/pytest_fitting.txt
