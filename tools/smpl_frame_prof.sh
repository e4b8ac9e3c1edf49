# Kernel durations of the wave-per-frame SMPL kernels at the closure's two sizes (32 and 1920 frames): rocprofv3 kernel stats of
# tools/smpl_frame_timing.py.  usage (on the GPU box): bash tools/smpl_frame_prof.sh [outdir]
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-smpl_frame_prof}
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 32 1920; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n$n -o n$n -- python $R/tools/smpl_frame_timing.py $n > $OUT/n$n.log 2>&1
  tail -1 $OUT/n$n.log
  f=$(find $OUT/n$n -name '*kernel_stats.csv' | head -1)
  grep smpl_frame $f | cut -d, -f1-8
  find $OUT/n$n -name '*.db' -delete; find $OUT/n$n -name '*kernel_trace.csv' -delete
done
