#!/usr/bin/env python
"""Which pieces of the stage-3 closure can be captured into a hipGraph?  (diagnostic)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                    # noqa: E402
from humor_amd import ops, synth                # noqa: E402


def try_capture(name, fn):
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        g.replay()
        torch.cuda.synchronize()
        print(f'[capture ok  ] {name}')
    except Exception as e:
        print(f'[capture FAIL] {name}: {str(e).splitlines()[0][:150]}')
        torch.cuda.synchronize()


def main():
    dev = torch.device('cuda:0')
    npz = synth.write_smplh_npz('/tmp/model_cb.npz', seed=0)
    fc = bench.FitClosure(dev, npz, 1, 0, None, use_graphs=False)
    o = fc.opt
    B, T = 32, 60
    lp = torch.randn(B, T, 32, device=dev, requires_grad=True)
    try_capture('vposer decode + R->aa (N=1920) fwd+bwd', lambda: o.latent2pose(lp).sum().backward())
    bp = torch.randn(B, T, 63, device=dev, requires_grad=True)
    try_capture('vposer encode fwd+bwd', lambda: o.pose2latent(bp).sum().backward())
    tr = torch.randn(B, T, 3, device=dev, requires_grad=True)
    ro = torch.randn(B, T, 3, device=dev, requires_grad=True)
    be = torch.randn(B, 16, device=dev, requires_grad=True)
    try_capture('smpl_results fwd+bwd', lambda: sum(v.sum() for k, v in o.smpl_results(tr, ro, bp, be)[0].items() if k != 'faces').backward())
    past = torch.randn(B, 339, device=dev, requires_grad=True)
    z = torch.randn(B, 59, 48, device=dev, requires_grad=True)

    def roll():
        out, (pm, pv) = o.motion_prior.roll_out(past, None, 59, z_seq=z, return_prior=True)
        (out['trans'].sum() + pm.sum()).backward()
    try_capture('roll-out fwd+bwd', roll)
    j = torch.randn(B, 1, 22, 3, device=dev, requires_grad=True)
    jv = torch.randn(B, 1, 22, 3, device=dev)
    tv = torch.randn(B, 1, 3, device=dev)
    try_capture('GMM init-state prior', lambda: o.fitting_loss.init_motion_prior_loss(j, jv, tv, tv).backward())
    fl = torch.tensor([[0.0, 0.5, 0.0]], device=dev).expand(B, 3).clone().requires_grad_(True)
    from humor_amd import frames
    try_capture('compute_cam2prior', lambda: sum(x.sum() for x in frames.compute_cam2prior(
        fl, tr[:, 0], ops.batch_rodrigues(ro[:, 0]), torch.randn(B, 22, 3, device=dev))).backward())
    obs = fc.obs_local
    j3 = (torch.randn(B, T, 22, 3, device=dev) + torch.tensor([0, 0, 5.0], device=dev)).requires_grad_(True)
    je = (torch.randn(B, T, 51, 3, device=dev) + torch.tensor([0, 0, 5.0], device=dev)).requires_grad_(True)
    try_capture('joints2d loss', lambda: o.fitting_loss.joints2d_loss(obs['joints2d'], j3, je).backward())
    v3 = torch.randn(B, T, 43, 3, device=dev, requires_grad=True)
    try_capture('overlap loss', lambda: sum(o.fitting_loss.overlap_verts_loss(obs['seq_interval'], v3)).backward())
    try_capture('full objective fwd', lambda: o._stage3_objective(fc.obs_local, None, fc.prior_params, False, 15, 1.0, fc.og_w, True, 'neutral'))
    def full():
        for p in fc.params:
            p.grad = None
        loss, _ = o._stage3_objective(fc.obs_local, None, fc.prior_params, False, 15, 1.0, fc.og_w, True, 'neutral')
        loss.backward()
    try_capture('full objective fwd+bwd', full)


if __name__ == '__main__':
    main()
