R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r02_run15
timeout 600 python tools/lbfgs_eval_breakdown.py 2>&1 | grep -v Warning | tail -5 | tee gpurun_out/r02_run15/breakdown.txt
