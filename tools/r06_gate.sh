# The driver's gate, as the driver runs it (fresh process each): pytest -m gpu -x -q, then smoke().  usage: bash tools/r06_gate.sh <outdir-name> [pytest args]
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06_gate}
mkdir -p $O
shift
t0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q --durations=10 "$@" > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -45 $O/pytest_gpu.txt | cut -c1-260
echo "pytest: $(( $(date +%s) - t0 )) s"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -4 $O/smoke.txt
(git rev-parse HEAD 2>/dev/null || true) > $O/head.txt; true
