#!/usr/bin/env python
"""Bare stage-1 / stage-2 / stage-3 closure times (eager and hipGraph replay) next to the time per closure evaluation inside
torch.optim.LBFGS.step: how much of an outer iteration is the optimiser itself."""
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    from humor_amd import synth
    dev = torch.device('cuda:0')
    npz = synth.write_smplh_npz(os.path.join(tempfile.mkdtemp(), 'm.npz'), seed=0)
    B = bench.B_SEQ
    for graphs in (False, True):
        opt = bench.build_optimizer(dev, npz, B, use_graphs=graphs)
        obs, init = bench.make_problem(B, bench.T_SEQ, seed=100, device=dev)
        opt.trans, opt.root_orient, opt.latent_pose, opt.betas = (init[k].clone() for k in ('trans', 'root_orient', 'latent_pose', 'betas'))
        opt.initialize(obs)
        ol = opt._local_obs(obs)
        for stage, names in ((0, ['trans', 'root_orient']), (1, ['trans', 'root_orient', 'betas', 'latent_pose'])):
            opt.fitting_loss.set_stage(stage)
            params = [getattr(opt, n) for n in names]
            for p in params:
                p.requires_grad_(True)
            obj = (lambda: opt._stage1_objective(ol, True)) if stage == 0 else (lambda: opt._stage2_objective(ol, True))
            closure = opt.make_closure(obj, params, None)
            t_bare = timeit(closure)
            optim = torch.optim.LBFGS(params, max_iter=20, lr=1.0, line_search_fn='strong_wolfe')
            e0 = opt.closure_evals
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(6):
                optim.step(closure)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ne = opt.closure_evals - e0
            print(f'graphs={graphs} stage {stage + 1}: bare closure {t_bare:.3f} ms; inside LBFGS.step: {dt / ne * 1e3:.3f} ms per closure evaluation '
                  f'({ne} evaluations in 6 outer iterations, {dt / 6 * 1e3:.1f} ms per outer iteration)', flush=True)


if __name__ == '__main__':
    main()
