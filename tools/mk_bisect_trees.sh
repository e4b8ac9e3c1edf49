# Archives the given commits (humor_amd/, include/, oracle/, the shared test helpers) under tools/microbench/bisect/<sha>/ and builds each library there,
# so that one gpurun call can run the same probe in every tree (tools/r06_bisect.sh, r06_accuracy_ab.sh, r06_lbfgs_cmp.sh).  usage: bash tools/mk_bisect_trees.sh <sha> ...
set -e
cd /root/repo
for sha in "$@"; do
  d=tools/microbench/bisect/$sha
  rm -rf $d; mkdir -p $d
  git archive $sha humor_amd include oracle tests/rollout_checks.py tests/conftest.py tests/golden/rollout_kink_flags.npz 2>/dev/null | tar -x -C $d || git archive $sha humor_amd include oracle tests/rollout_checks.py tests/conftest.py | tar -x -C $d
  mkdir -p $d/tools; cp tools/nan_hunt.py $d/tools/
  (cd $d && python -m humor_amd.build > build.log 2>&1 && rm -f humor_amd/csrc/*.o && echo built $sha) 
done
