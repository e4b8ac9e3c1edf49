#!/usr/bin/env python
"""Where does one stage-3 closure go?  GPU-busy time vs wall time, per-section wall times (synchronised), top kernels."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                             # noqa: E402
from humor_amd import synth                              # noqa: E402


def main():
    dev = torch.device('cuda:0')
    npz = synth.write_smplh_npz('/tmp/model_bd.npz', seed=0)
    fc = bench.FitClosure(dev, npz, 1, 0, None, use_graphs='--graph' in sys.argv)
    for _ in range(3):
        fc.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fc.step()
    torch.cuda.synchronize()
    print(f'closure wall: {(time.perf_counter() - t0) * 100:.2f} ms')
    # forward only / backward only
    o = fc.opt
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        loss, _ = o._stage3_objective(fc.obs_local, None, fc.prior_params, False, 15, 1.0, fc.og_w, True, 'neutral')
    torch.cuda.synchronize()
    print(f'objective forward wall: {(time.perf_counter() - t0) * 100:.2f} ms')
    # roll-out alone
    hm = o.motion_prior
    past = torch.randn(32, 339, device=dev, requires_grad=True)
    z = torch.randn(32, 59, 48, device=dev, requires_grad=True)
    for _ in range(2):
        out, (pm, pv) = hm.roll_out(past, None, 59, z_seq=z, return_prior=True)
        (out['trans'].sum() + pm.sum()).backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out, (pm, pv) = hm.roll_out(past, None, 59, z_seq=z, return_prior=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(10):
        out, (pm, pv) = hm.roll_out(past, None, 59, z_seq=z, return_prior=True)
        (out['trans'].sum() + pm.sum()).backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'roll-out forward wall: {(t1 - t0) * 100:.2f} ms ; forward+backward: {(t2 - t1) * 100:.2f} ms')
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            fc.step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by='self_cuda_time_total', row_limit=14, max_name_column_width=60))
    print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=14, max_name_column_width=60))


if __name__ == '__main__':
    main()
