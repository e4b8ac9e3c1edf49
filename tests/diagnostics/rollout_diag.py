#!/usr/bin/env python
"""Per-step error growth of the roll-out chain: GPU fp32 vs CPU-oracle fp32 vs CPU-oracle fp64 (conditioning study)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from humor_amd import synth                        # noqa: E402
from humor_amd.humor_model import HumorModel       # noqa: E402
from oracle import humor_restated as H             # noqa: E402
from oracle.make_golden import canonical_state     # noqa: E402


def main():
    B, S, seed = 32, 59, 3
    dev = torch.device('cuda:0')
    torch.set_num_threads(min(32, os.cpu_count()))
    sd = synth.humor_state_dict(seed=seed)
    hm = HumorModel(in_rot_rep='mat', out_rot_rep='aa', model_data_config='smpl+joints+contacts')
    hm.load_state_dict(sd)
    hm = hm.to(dev).eval()
    g = torch.Generator().manual_seed(seed + 5)
    past = canonical_state(B, g)
    z = torch.randn(B, S, 48, generator=g)
    out, (pm, pv) = hm.roll_out(past.to(dev), None, S, z_seq=z.to(dev), return_prior=True)
    world = torch.cat([out[k] for k in ('trans', 'trans_vel', 'root_orient', 'root_orient_vel', 'pose_body', 'joints',
                                         'joints_vel', 'contacts')], 2).cpu()
    w32, (pm32, _) = H.roll_out(sd, past, z)
    w64, (pm64, _) = H.roll_out({k: v.double() for k, v in sd.items()}, past.double(), z.double())
    print('step  |gpu-f64|   |cpu32-f64|  |gpu-cpu32|   max|state|   |pm gpu-f64| |pm cpu32-f64|')
    for t in list(range(0, 12)) + list(range(12, S, 4)) + [S - 1]:
        e = lambda a, b: (a[:, t].double() - b[:, t].double()).abs().max().item()
        print(f'{t:4d}  {e(world, w64):.3e}  {e(w32, w64):.3e}   {e(world, w32):.3e}   {w64[:, t].abs().max().item():9.3f}'
              f'   {e(pm.cpu(), pm64):.3e}  {e(pm32, pm64):.3e}')


if __name__ == '__main__':
    main()
