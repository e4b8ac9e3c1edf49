# old (round-5 mid, d89fb75: before the half-granule sweeps) vs new kernels against fp64 on 36 inputs of the random network + 12 of the contractive one
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_accuracy
mkdir -p $O
C1=$(python -c "print(','.join([f'32x12x{s}' for s in range(1,13)]+[f'4x10x{s}' for s in range(1,13)]+[f'8x8x{s}' for s in range(1,13)]))")
C2=$(python -c "print(','.join([f'32x12x{s}' for s in range(1,7)]+[f'8x30x{s}' for s in range(1,7)]))")
d=tools/microbench/bisect/d89fb75
cp tools/accuracy_probe.py $d/tools/
(cd $d && HUMOR_AMD_TEST_POISON=0 timeout 900 python tools/accuracy_probe.py --cases $C1 > $O/random_old_d89fb75.txt 2>&1); tail -1 $O/random_old_d89fb75.txt
timeout 900 python tools/accuracy_probe.py --cases $C1 > $O/random_new.txt 2>&1; tail -1 $O/random_new.txt
(cd $d && timeout 900 python tools/accuracy_probe.py --contractive --cases $C2 > $O/contractive_old_d89fb75.txt 2>&1); tail -1 $O/contractive_old_d89fb75.txt
timeout 900 python tools/accuracy_probe.py --contractive --cases $C2 > $O/contractive_new.txt 2>&1; tail -1 $O/contractive_new.txt
