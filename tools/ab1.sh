mkdir -p gpurun_out/ab3
for i in 1 2 3; do
for v in old new; do
  if [ $v = new ]; then export HUMOR_AMD_LIB=; else export HUMOR_AMD_LIB=$PWD/ab_old/libhumor_amd_old.so; fi
  echo "== $v $i $(timeout 300 python tools/persist_timing.py quick 2>&1 | tail -1)" >> gpurun_out/ab3/ab.txt
done
done
export HUMOR_AMD_LIB=
timeout 900 python -m pytest tests/test_rollout_gpu.py -q 2>&1 | tail -3 >> gpurun_out/ab3/ab.txt
