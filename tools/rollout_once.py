#!/usr/bin/env python
"""Runs the HuMoR roll-out forward+backward a few times at 32x59 (for rocprofv3 --pmc runs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from humor_amd import synth                       # noqa: E402
from humor_amd.humor_model import HumorModel      # noqa: E402

dev = torch.device('cuda:0')
hm = HumorModel(in_rot_rep='mat', out_rot_rep='aa', model_data_config='smpl+joints+contacts')
hm.load_state_dict(synth.humor_state_dict(seed=0))
hm = hm.to(dev).eval()
past = torch.randn(32, 339, device=dev, requires_grad=True)
z = torch.randn(32, 59, 48, device=dev, requires_grad=True)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    out, (pm, pv) = hm.roll_out(past, None, 59, z_seq=z, return_prior=True)
    (out['trans'].sum() + pm.sum()).backward()
torch.cuda.synchronize()
print('done')
