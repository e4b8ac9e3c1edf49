#!/usr/bin/env python
"""fp32 accuracy of the roll-out forward and adjoint vs an fp64 oracle, GPU kernels vs the CPU fp32 oracle, by chain length."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import rollout_checks as RC                       # noqa: E402
from humor_amd import _lib                        # noqa: E402
from oracle import humor_restated as H            # noqa: E402


def main():
    dev = torch.device('cuda:0')
    lib = _lib.get_lib()
    B = 32
    torch.set_num_threads(min(32, os.cpu_count()))
    for seed in (4, 7):
        hm, sd = RC.make_model(lib, dev, seed=seed)
        sd64 = {k: v.double() for k, v in sd.items()}
        for S in (1, 2, 4, 8, 12, 16):
            g = torch.Generator().manual_seed(seed + 5)
            past_c = RC.canonical_state(B, g)
            z_c = torch.randn(B, S, 48, generator=g)
            gw = torch.randn(B, S, 348, generator=g)
            res = {}
            for name in ('gpu', 'cpu32', 'f64'):
                if name == 'gpu':
                    p, z = past_c.to(dev).requires_grad_(True), z_c.to(dev).requires_grad_(True)
                    out, (pm, pv) = hm.roll_out(p, None, S, z_seq=z, return_prior=True)
                    w = RC.world_of(out)
                    loss = (w * gw.to(dev)).sum() + pm.sum()
                elif name == 'cpu32':
                    p, z = past_c.clone().requires_grad_(True), z_c.clone().requires_grad_(True)
                    w, (pm, pv) = H.roll_out(sd, p, z)
                    loss = (w * gw).sum() + pm.sum()
                else:
                    p, z = past_c.double().requires_grad_(True), z_c.double().requires_grad_(True)
                    w, (pm, pv) = H.roll_out(sd64, p, z)
                    loss = (w * gw.double()).sum() + pm.sum()
                gp, gz = torch.autograd.grad(loss, [p, z])
                res[name] = (w.detach().cpu().double(), gp.cpu().double(), gz.cpu().double())
            e = lambda a, i: (res[a][i] - res['f64'][i]).abs().max().item()
            print(f'seed {seed} S={S:2d}  fwd gpu {e("gpu", 0):.2e} cpu {e("cpu32", 0):.2e} | g_past gpu {e("gpu", 1):.2e} cpu {e("cpu32", 1):.2e} '
                  f'scale {res["f64"][1].abs().max().item():.1f} | g_z gpu {e("gpu", 2):.2e} cpu {e("cpu32", 2):.2e} scale {res["f64"][2].abs().max().item():.1f}',
                  flush=True)


if __name__ == '__main__':
    main()
