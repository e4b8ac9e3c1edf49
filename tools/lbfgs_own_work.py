#!/usr/bin/env python
"""The fused L-BFGS's OWN GPU work per inner iteration at the stage-3 size (n = 94 752 variables, history 100, full): a closure that costs one
elementwise kernel, so that a rocprofv3 --kernel-trace --stats of this script lists what the optimiser itself launches.
usage: python tools/lbfgs_own_work.py [n] [outer steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from humor_amd.lbfgs import LBFGS      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 94752
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
a = (0.5 + torch.rand(n, generator=g)).to(dev)
b = torch.randn(n, generator=g).to(dev)
p = torch.zeros(n, device=dev, requires_grad=True)
opt = LBFGS([p], max_iter=20, lr=1.0, line_search_fn='strong_wolfe', history_size=100)
evals = [0]


def closure():
    p.grad = None
    evals[0] += 1
    l = (0.5 * a * p * p - b * p + 0.05 * torch.sin(3.0 * p)).sum()
    l.backward()
    return l


for _ in range(6):            # fill the history
    opt.step(closure)
torch.cuda.synchronize()
e0 = evals[0]
t0 = time.perf_counter()
for _ in range(steps):
    opt.step(closure)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'n = {n}: {steps} step() calls, {evals[0] - e0} evaluations, {1e3 * dt / max(1, evals[0] - e0):.3f} ms wall per evaluation (closure ~ 6 small kernels), '
      f'history pairs {len(opt._hist["order"])}')
