# L2 request counters of the wave-per-frame SMPL kernels at N = 1920 (subset path): separate --pmc pass with --kernel-trace only.
# usage (on the GPU box): bash tools/pmc_smpl_frame.sh [outdir]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc_smpl_frame}
rm -rf $OUT && mkdir -p $OUT
timeout 200 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/TCC -- python $R/tools/smpl_frame_timing.py 1920 > $OUT/TCC.log 2>&1
timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/TCP -- python $R/tools/smpl_frame_timing.py 1920 > $OUT/TCP.log 2>&1
find $OUT -name "*.db" -delete
python - <<PY
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'smpl_frame' in k:
            rows['fwd' if 'fwd' in k else 'bwd'][r['Counter_Name']].append(float(r['Counter_Value']))
for k in rows:
    for c, v in sorted(rows[k].items()):
        print(f'smpl_frame_{k}  {c:32s} n={len(v):3d}  mean {sum(v)/len(v):14.1f}')
PY
tail -3 $OUT/TCC.log | cut -c1-200
