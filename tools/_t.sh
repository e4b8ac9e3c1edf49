cd $GRAFT_REPO_ROOT
HUMOR_AMD_LIB=tools/microbench/libhumor_amd_ptiming.so timeout 200 python tools/persist_phase_timing.py 1 2>&1 | grep -E "extra polls|median step"
