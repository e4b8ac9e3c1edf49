#!/usr/bin/env python
"""Loss traces of a short three-stage fit (k outer iterations per stage / phase, eager closures) with and without the stage-3 composite
nodes: where the two L-BFGS trajectories part (gradient contributions are summed in a different order: fp32 rounding, amplified by the
line search) and how both objectives keep falling.  usage: lbfgs_trace_ab.py"""
import json, os, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from humor_amd import synth
dev = torch.device('cuda:0')
npz = synth.write_smplh_npz(os.path.join(tempfile.mkdtemp(), 'm.npz'), seed=0)
traces = []
k = 3
for nodes in (1, 0):
    opt = bench.build_optimizer(dev, npz, bench.B_SEQ, use_graphs=False)
    opt.fused_stage3 = bool(nodes)
    opt.fitting_loss.fold_init_prior = bool(nodes)
    opt.stage3_tune_init_freeze_start, opt.stage3_tune_init_freeze_end = k, 2 * k
    opt.loss_trace = []
    obs, _ = bench.make_problem(bench.B_SEQ, bench.T_SEQ, seed=100, device=dev)
    opt.run(obs, data_fps=30, lr=1.0, num_iter=[k, k, 3 * k], lbfgs_max_iter=20)
    traces.append(opt.loss_trace)
    print('nodes', nodes, 'evaluations', len(opt.loss_trace))
a, b = traces
n = min(len(a), len(b))
first = None
for i in range(n):
    rel = abs(a[i][1] - b[i][1]) / max(1.0, abs(b[i][1]))
    if a[i][0] != b[i][0] or rel > 1e-5:
        first = i
        break
print('first differing evaluation', first, 'of', n)
lo = max(0, (first or 0) - 3)
for i in range(lo, min(n, lo + 40)):
    print(i, a[i], b[i])
