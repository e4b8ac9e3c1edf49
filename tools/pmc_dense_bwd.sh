# SQ / LDS / VMEM / MFMA counters of the dense SMPL backward's kernels at N = 1920 (tools/smpl_dense_bwd_timing.py): separate --pmc passes with
# --kernel-trace only.   usage (on the GPU box): bash tools/pmc_dense_bwd.sh [outdir]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc_dense_bwd}
rm -rf $OUT && mkdir -p $OUT
run() { n=$1; shift; timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -- python $R/tools/smpl_dense_bwd_timing.py > $OUT/$n.log 2>&1; }
run SQ SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES
run LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS
run VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
find $OUT -name "*.db" -delete
python - <<PY
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if any(s in k for s in ('compressed_gA', 'dense_gco', 'dense_gvp')):
            rows[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(rows):
    for c, v in sorted(rows[k].items()):
        print(f'{k:40s} {c:28s} n={len(v):3d}  mean {sum(v)/len(v):16.1f}')
PY
