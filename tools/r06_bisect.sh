cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_bisect
mkdir -p $O
for sha in d89fb75 bf9ba7c 0378e1f 59112f5 1af6acc 96914f9; do
  d=tools/microbench/bisect/$sha
  cp tools/accuracy_probe.py $d/tools/
  (cd $d && HUMOR_AMD_TEST_POISON=0 timeout 300 python tools/accuracy_probe.py > $O/acc_$sha.txt 2>&1); echo "== $sha"; tail -9 $O/acc_$sha.txt | cut -c1-330
done
echo "== HEAD"; timeout 300 python tools/accuracy_probe.py > $O/acc_head.txt 2>&1; tail -9 $O/acc_head.txt | cut -c1-330
echo "== HEAD glue_ieee"; HUMOR_AMD_LIB=$GRAFT_REPO_ROOT/tools/microbench/libhumor_amd_glue_ieee.so timeout 300 python tools/accuracy_probe.py > $O/acc_head_ieee.txt 2>&1; tail -9 $O/acc_head_ieee.txt | cut -c1-330
echo "== HEAD contractive"; timeout 300 python tools/accuracy_probe.py --contractive --cases 130x2x130,288x2x288,70x3x70,33x5x33,32x12x32 --paths persistent,mixed,chain > $O/acc_head_c.txt 2>&1; tail -7 $O/acc_head_c.txt | cut -c1-330
echo "== HEAD glue_ieee contractive"; HUMOR_AMD_LIB=$GRAFT_REPO_ROOT/tools/microbench/libhumor_amd_glue_ieee.so timeout 300 python tools/accuracy_probe.py --contractive --cases 130x2x130,288x2x288,70x3x70,33x5x33,32x12x32 --paths persistent,mixed,chain > $O/acc_head_ieee_c.txt 2>&1; tail -7 $O/acc_head_ieee_c.txt | cut -c1-330
