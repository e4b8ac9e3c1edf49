cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_lbfgs
mkdir -p $O
timeout 600 python tools/lbfgs_phase_profile.py 5 0 2>&1 | grep -v Warn | tail -8 > $O/head.txt; cat $O/head.txt | cut -c1-260
# the same problem with the glue of round 5's last kernels (archived tree) -- trajectory comparison
d=tools/microbench/bisect/96914f9
mkdir -p $d/tools; cp $d/../../../lbfgs_phase_profile.py $d/tools/ 2>/dev/null || cp tools/lbfgs_phase_profile.py $d/tools/
git -C . show 96914f9:bench.py > /dev/null 2>&1
(cd $d && timeout 600 python tools/lbfgs_phase_profile.py 5 0 2>&1 | grep -v Warn | tail -8 > $O/r05_96914f9.txt); cat $O/r05_96914f9.txt | cut -c1-260
