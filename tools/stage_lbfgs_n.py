#!/usr/bin/env python
"""Runs K outer L-BFGS iterations (humor_amd.lbfgs.LBFGS, max_iter 20, strong Wolfe) of the stage-1 or stage-2 closure of the C4 problem
with hipGraph replay (kernel-trace driver: per-evaluation kernel census of the short stages).  usage: stage_lbfgs_n.py STAGE K"""
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
    from humor_amd.lbfgs import LBFGS, flat_arena
    stage, K = int(sys.argv[1]), int(sys.argv[2])
    dev = torch.device('cuda:0')
    npz = synth.write_smplh_npz(os.path.join(tempfile.mkdtemp(), 'm.npz'), seed=0)
    B = bench.B_SEQ
    from humor_amd import _lib
    _lib.get_lib().call('ha_tune_set', b'gemm_ks', int(os.environ.get('GEMM_KS', 0)))      # experiment knob: K split of the small GEMMs
    opt = bench.build_optimizer(dev, npz, B, use_graphs=True)
    obs, init = bench.make_problem(B, bench.T_SEQ, seed=100, device=dev)
    names = ['trans', 'root_orient'] if stage == 1 else ['trans', 'root_orient', 'betas', 'latent_pose']
    flat, views = flat_arena([init[n].shape for n in ['trans', 'root_orient', 'betas', 'latent_pose']], dev)
    for n, v in zip(['trans', 'root_orient', 'betas', 'latent_pose'], views):
        v.copy_(init[n])
        setattr(opt, n, v)
    opt.initialize(obs)
    ol = opt._local_obs(obs)
    opt.fitting_loss.set_stage(stage - 1)
    params = [getattr(opt, n) for n in names]
    for p in params:
        p.requires_grad_(True)
    obj = (lambda: opt._stage1_objective(ol, True)) if stage == 1 else (lambda: opt._stage2_objective(ol, True))
    closure = opt.make_closure(obj, params, None)
    optim = LBFGS(params, max_iter=20, lr=1.0, line_search_fn='strong_wolfe')
    optim.step(closure)                     # capture + first step
    torch.cuda.synchronize()
    e0 = opt.closure_evals
    t0 = time.perf_counter()
    for _ in range(K):
        optim.step(closure)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ne = opt.closure_evals - e0
    print(f'stage {stage}: {K} outer iterations, {ne} closure evaluations, {dt / ne * 1e3:.3f} ms per evaluation, {K / dt:.1f} outer iterations/s', flush=True)


if __name__ == '__main__':
    main()
