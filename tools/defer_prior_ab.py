#!/usr/bin/env python
"""A/B of the side-stream conditional prior (MotionOptimizer(defer_prior=...)) on the C4 stage-3 closure, back to back, alternating
the variants several times on the same instance.  usage: defer_prior_ab.py [rounds]
variants: off = everything in stream order; fwd = forward on the side stream only; both = forward and adjoint"""
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
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device('cuda:0')
    npz = synth.write_smplh_npz(os.path.join(tempfile.mkdtemp(), 'm.npz'), seed=0)
    fc = bench.FitClosure(dev, npz, 1, 0, None, use_graphs=False)
    o = fc.opt
    hook = o.motion_prior.mark_prior_grad

    def set_variant(v):
        o.defer_prior = v in ('fwd', 'both')
        o.fitting_loss.prior_grad_hook = hook if v in ('both', 'bwd') else None

    res = {}
    for r in range(rounds):
        for v in ('off', 'fwd', 'bwd', 'both'):
            set_variant(v)
            for _ in range(5):
                fc.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(40):
                fc.step()
            torch.cuda.synchronize()
            res.setdefault(v, []).append((time.perf_counter() - t0) / 40 * 1e3)
    for v, t in res.items():
        print(f'{v:5s}: ' + ' '.join(f'{x:.3f}' for x in t) + f'   median {sorted(t)[len(t) // 2]:.3f} ms per evaluation')
    print('side-stream statistics (forwards deferred, adjoints at a mark):', o.motion_prior.prior_side_stats(o.trans))


if __name__ == '__main__':
    main()
