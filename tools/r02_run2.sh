# round 2, GPU session 2: all GPU tests (no -x), spb A/B of the decoder chain, bench
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run2
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.txt 2>&1; grep -n "short run\|passed\|failed\|FAILED\|^c[2345] " $OUT/pytest_gpu.txt | cut -c1-900
timeout 300 python tools/rollout_ab.py 32 59 "4,1" "6,1" "8,1" "12,1" "17,1" "4,2" "8,2" "17,2" > $OUT/rollout_ab_32.txt 2>&1; cat $OUT/rollout_ab_32.txt
timeout 300 python tools/rollout_ab.py 256 119 "4,1" > $OUT/rollout_ab_256.txt 2>&1; cat $OUT/rollout_ab_256.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-c5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err
