R=$GRAFT_REPO_ROOT
cd $R
timeout 45 python -m pytest tests/test_rollout_gpu.py -q -x 2>&1 | tail -2 | cut -c1-200
