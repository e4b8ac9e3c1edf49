"""A stage-3 closure evaluation as L-BFGS sees it: evaluate, read the loss on the host (a device synchronise), evaluate again.  Prints the
wall time per evaluation with and without the host read between evaluations, and -- with a torch profiler trace of a few synchronised
evaluations -- when, after the host read, the first kernel and the persistent forward START on the GPU (the head the host has to issue
before the GPU has work for 0.5 ms).   usage: closure_sync_gap.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                             # noqa: E402
from humor_amd import synth                              # noqa: E402

dev = torch.device('cuda:0')
npz = synth.write_smplh_npz('/tmp/model_sg.npz', seed=0)
fc = bench.FitClosure(dev, npz, 1, 0, None, use_graphs=False)
for _ in range(5):
    fc.step()
torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
for _ in range(n):
    fc.step()
torch.cuda.synchronize()
back = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n):
    float(fc.step())
sync = (time.perf_counter() - t0) / n
print('closure evaluation: back to back %.3f ms, with a host read after each %.3f ms (exposed host time %.3f ms)' % (1e3 * back, 1e3 * sync, 1e3 * (sync - back)))
# host-side: time from the start of an evaluation to the return of the Python call (all launches issued), after a synchronise
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fc.step()
    ts.append(time.perf_counter() - t0)
print('host time to ISSUE one evaluation (forward + backward launches) after a synchronise: median %.3f ms' % (1e3 * sorted(ts)[len(ts) // 2]))
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(4):
        float(fc.step())
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
ev.sort(key=lambda e: e.time_range.start)
names = [e.name for e in ev]
idx = [i for i, nm in enumerate(names) if 'rollout_persist_fwd' in nm]
for a, b in zip(idx[:-1], idx[1:]):
    seg = ev[a:b + 1]
    # the evaluation boundary = the largest gap between two consecutive kernels of the segment (the host read)
    gaps = [(seg[i + 1].time_range.start - seg[i].time_range.end, i) for i in range(len(seg) - 1)]
    g, i = max(gaps)
    head = seg[i + 1:]
    print('after the host read: GPU idle %.0f us; then %d kernels in %.0f us (busy %.0f us) before the persistent forward starts' % (
        g, len(head) - 1, head[-1].time_range.start - head[0].time_range.start, sum(e.time_range.end - e.time_range.start for e in head[:-1])))
