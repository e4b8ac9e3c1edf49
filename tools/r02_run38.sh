R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_smpl_gpu.py -q -x -k "skin" 2>&1 | tail -3
timeout 300 python tools/skin_sustained.py 30720 2>&1 | grep -v amdgpu.ids | grep "first" | cut -c1-200
timeout 300 python tools/skin_sustained.py 1920 2>&1 | grep -v amdgpu.ids | grep "first" | cut -c1-200
