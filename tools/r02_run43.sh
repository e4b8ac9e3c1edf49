R=$GRAFT_REPO_ROOT
cd $R
timeout 40 python -m pytest tests/test_rollout_gpu.py -q -x -k "k_split" 2>&1 | tail -1
GEMM_KS=0 timeout 30 python tools/stage_lbfgs_n.py 2 6 2>&1 | tail -1
GEMM_KS=2 timeout 30 python tools/stage_lbfgs_n.py 2 6 2>&1 | tail -1
