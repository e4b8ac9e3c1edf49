#!/usr/bin/env python
"""Every GPU kernel of ONE eager stage-3 closure evaluation in launch order, with the ATen op that launched it and (forward ops) the
innermost humor_amd / bench frame: which line of the host code pays for each small launch.  usage: closure_launch_list.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                             # noqa: E402
from humor_amd import synth                              # noqa: E402


def main():
    dev = torch.device('cuda:0')
    npz = synth.write_smplh_npz('/tmp/model_ll.npz', seed=0)
    fc = bench.FitClosure(dev, npz, 1, 0, None, use_graphs=False)
    for _ in range(3):
        fc.step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        fc.step()
        torch.cuda.synchronize()
    rows = []
    for e in prof.events():
        if not e.kernels:
            continue
        where = ''
        for fr in (e.stack or []):
            if ('humor_amd/' in fr or 'bench.py' in fr) and 'torch/' not in fr:
                where = fr.split('/')[-1]
                break
        for k in e.kernels:
            rows.append((e.time_range.start, e.name, k.name, k.duration, where))
    rows.sort()
    tot = 0.0
    print(f'{len(rows)} kernels')
    for i, (_, op, kn, dur, where) in enumerate(rows):
        tot += dur
        print(f'{i:3d} {dur:7.1f} us  {op[:44]:44s} {kn[:60]:60s} {where[:70]}')
    print(f'kernel time {tot / 1e3:.3f} ms')


if __name__ == '__main__':
    main()
