R=$GRAFT_REPO_ROOT
cd $R
timeout 70 python -m pytest tests/test_fitting_gpu.py -q -x -s -k "lbfgs_kernels" 2>&1 | grep -E "direction|passed|failed|Error|error" | cut -c1-200
