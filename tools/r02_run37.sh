R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python tools/skin_sustained.py 1920 2>&1 | grep -v amdgpu.ids | cut -c1-200
