R=$GRAFT_REPO_ROOT
cd $R
timeout 60 python -m pytest tests/test_humor_loss_gpu.py -q -x 2>&1 | tail -2 | cut -c1-200
