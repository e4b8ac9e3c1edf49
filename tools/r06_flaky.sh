# flakiness check: the driver's smoke 30 x in fresh processes, the e2e file 6 x, the roll-out failure / graph tests 6 x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_flaky; mkdir -p $O
ok=0; bad=0
for i in $(seq 1 30); do
  if timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$i.txt 2>&1; then ok=$((ok+1)); rm $O/smoke_$i.txt; else bad=$((bad+1)); fi
done
echo "smoke: $ok ok, $bad failed" | tee $O/summary.txt
for i in $(seq 1 6); do
  timeout 600 python -m pytest tests/test_e2e_gpu.py -m gpu -q -x 2>&1 | tail -1 | tee -a $O/summary.txt
done
for i in $(seq 1 6); do
  timeout 600 python -m pytest tests/test_rollout_gpu.py -m gpu -q -x -k "failure or determinism or stash_mode or small_batches" 2>&1 | tail -1 | tee -a $O/summary.txt
done
