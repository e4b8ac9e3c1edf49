# Per-closure kernel census: two rocprofv3 kernel traces of tools/closure_n.py (5 and 25 evaluations) -> tools/closure_trace_diff.py
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-census}
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 5 25; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t$n -o t$n -- python $R/tools/closure_n.py $n > $OUT/t$n.log 2>&1
done
a=$(find $OUT/t5 -name '*kernel_trace.csv' | head -1); b=$(find $OUT/t25 -name '*kernel_trace.csv' | head -1)
python $R/tools/closure_trace_diff.py $a 5 $b 25 > $OUT/closure_census.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name '*kernel_trace.csv' -delete
head -70 $OUT/closure_census.txt | cut -c1-170
