#!/usr/bin/env python
"""Where a stage-3 closure evaluation inside the fused L-BFGS spends its wall time: host time to ISSUE the closure (forward +
backward launches), host wait for the GPU at the line search's read, and the optimiser's own work."""
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from humor_amd import synth
    dev = torch.device('cuda:0')
    npz = synth.write_smplh_npz(os.path.join(tempfile.mkdtemp(), 'm.npz'), seed=0)
    fc = bench.FitClosure(dev, npz, 1, 0, None, use_graphs=False)
    for _ in range(5):
        fc.step()
    torch.cuda.synchronize()
    # 1. issue time vs total time of a bare closure evaluation, one at a time (a sync after each, like the line search)
    issue, total = [], []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = fc.step()
        t1 = time.perf_counter()
        float(loss.detach())
        t2 = time.perf_counter()
        issue.append(t1 - t0)
        total.append(t2 - t0)
    print(f'synchronised closure: host issue {1e3 * sum(issue) / 20:.2f} ms, until the loss is on the host {1e3 * sum(total) / 20:.2f} ms')
    # 2. back-to-back (the bench's regime)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fc.step()
    torch.cuda.synchronize()
    print(f'back-to-back closure: {1e3 * (time.perf_counter() - t0) / 20:.2f} ms')
    # 3. inside the fused L-BFGS
    from humor_amd.lbfgs import LBFGS
    opt = LBFGS(fc.params, max_iter=20, lr=1.0, line_search_fn='strong_wolfe')
    opt.step(fc.closure)          # first step: steepest descent + history set-up, kept out of the timeline
    opt.profile = {}
    e0 = fc.opt.closure_evals
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        opt.step(fc.closure)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ne = fc.opt.closure_evals - e0
    print(f'inside LBFGS.step: {1e3 * dt / ne:.2f} ms per closure evaluation ({ne} evaluations)')
    print('host timeline per evaluation (ms): ' + ', '.join(f'{k} {1e3 * v / ne:.3f}' for k, v in opt.profile.items()) +
          f' | sum {1e3 * sum(opt.profile.values()) / ne:.3f}')


if __name__ == '__main__':
    main()
