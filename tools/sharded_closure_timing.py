#!/usr/bin/env python
"""The N > 1 closure path on ONE GPU: a world-size-1 RCCL group, the sharded stage-3 closure (halo all-gather + packed gradient
all-reduce, eager launches) timed back to back beside the un-sharded eager closure, and its launch list.  usage: sharded_closure_timing.py"""
import os
import sys
import tempfile
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def timed(fc, n=40):
    for _ in range(5):
        fc.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fc.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    from humor_amd import synth
    dev = torch.device('cuda:0')
    npz = synth.write_smplh_npz(os.path.join(tempfile.mkdtemp(), 'm.npz'), seed=0)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(29400 + os.getpid() % 500))
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    plain = bench.FitClosure(dev, npz, 1, 0, None, use_graphs=False)
    shard = bench.FitClosure(dev, npz, 1, 0, dist.group.WORLD, use_graphs=False)
    for r in range(3):
        print(f'un-sharded eager {timed(plain):.3f} ms   sharded (world 1, RCCL) {timed(shard):.3f} ms', flush=True)
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        shard.step()
        torch.cuda.synchronize()
    rows = []
    for e in prof.events():
        for k in (e.kernels or []):
            rows.append((e.time_range.start, e.name, k.name, k.duration))
    rows.sort()
    print(len(rows), 'kernels in one sharded evaluation')
    for i, (_, op, kn, dur) in enumerate(rows):
        if 'ha::' not in kn:
            print(f'{i:3d} {dur:7.1f} us  {op[:40]:40s} {kn[:70]}')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
