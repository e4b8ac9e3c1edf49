#!/usr/bin/env python
"""bench.lbfgs_profile (real L-BFGS outer iterations of every stage / stage-3 phase) with the stage-3 composite nodes and the cached
unit seed switched on and off, same process (the first configuration also pays the process warm-up).  usage: lbfgs_nodes_ab.py"""
import json, os, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from humor_amd import synth, motion_optimizer as MO, fit_kernels as FK
dev = torch.device('cuda:0')
npz = synth.write_smplh_npz(os.path.join(tempfile.mkdtemp(), 'm.npz'), seed=0)
orig_build = bench.build_optimizer
orig_finish = MO.MotionOptimizer._finish_closure
def plain_finish(self, loss, params, stats=None):
    loss.backward()
    self.closure_evals += 1
    if self.loss_trace is not None:
        self.loss_trace.append((self.fitting_loss.cur_stage_idx, float(loss.detach())))
    return loss
for nodes, unit in ((1, 1), (0, 1), (0, 0), (1, 0)):
    def build(*a, **k):
        o = orig_build(*a, **k)
        o.fused_stage3 = bool(nodes)
        o.fitting_loss.fold_init_prior = bool(nodes)
        return o
    bench.build_optimizer = build
    MO.MotionOptimizer._finish_closure = orig_finish if unit else plain_finish
    r = bench.lbfgs_profile(dev, npz)
    print(f'nodes={nodes} unit_seed={unit}: whole fit {r["whole_fit_seconds_for_30_80_70_schedule"]} s; ' + '; '.join(
        f'{k} {v["outer_iters_per_sec"]:.1f}/s {v["closure_evals_per_outer_iter"]} ev/it {v["ms_per_closure_eval"]} ms' for k, v in r['phases'].items()), flush=True)
