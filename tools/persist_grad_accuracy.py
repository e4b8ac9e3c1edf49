"""Gradient accuracy of the roll-out paths against the float64 oracle on the inputs of tests/rollout_checks.check_persistent_vs_chain
(32 x 59, contractive weights): per-sequence error of dL/dpast and dL/dz (relative to the largest entry) for the launch chain and the
persistent kernels.  Evidence for the tolerance of test_persistent_forward_matches_launch_chain: which of two fp32 paths is nearer
to the exact gradient on the sequences where they differ.   usage: persist_grad_accuracy.py [B S]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch

import rollout_checks as RC
from humor_amd import _lib
from oracle import humor_restated as H

B, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 59)
dev = torch.device('cuda:0')
lib = _lib.get_lib()
hm, sd = RC.make_model(lib, dev, seed=B, contractive=True)      # (the test uses seed = B)
g = torch.Generator().manual_seed(77 + B + S)
past, z = RC.canonical_state(B, g), torch.randn(B, S, 48, generator=g)
gw = torch.randn(B, S, 348, generator=g)
gm, gv = torch.randn(B, S, 48, generator=g), torch.randn(B, S, 48, generator=g)
p64, z64 = past.double().requires_grad_(True), z.double().requires_grad_(True)
w64, (pm64, pv64) = H.roll_out({k: v.double() for k, v in sd.items()}, p64, z64)
((w64 * gw.double()).sum() + (pm64 * gm.double()).sum() + (pv64 * gv.double()).sum()).backward()
ref = (p64.grad, z64.grad)
print('oracle (float64) done; max |g_past| %.3g max |g_z| %.3g' % (ref[0].abs().max(), ref[1].abs().max()))
for name, knob, bwd in (('launch chain', 0, 0), ('persistent fwd + chain adjoint', 1, 0), ('persistent', 1, 1)):
    lib.call('ha_tune_set', b'rollout_persist', knob)
    lib.call('ha_tune_set', b'rollout_persist_bwd', bwd)
    p, zz = past.to(dev).requires_grad_(True), z.to(dev).requires_grad_(True)
    out, (pm, pv) = hm.roll_out(p, None, S, z_seq=zz, return_prior=True)
    w = RC.world_of(out)
    ((w * gw.to(dev)).sum() + (pm * gm.to(dev)).sum() + (pv * gv.to(dev)).sum()).backward()
    for gname, got, r in (('g_past', p.grad, ref[0]), ('g_z', zz.grad, ref[1])):
        e = (got.cpu().double() - r).abs().reshape(B, -1).amax(dim=1) / max(1.0, r.abs().max().item())
        top = sorted(((float(x), i) for i, x in enumerate(e)), reverse=True)[:6]
        print(f'{name:32s} {gname:6s} vs float64: median {e.median().item():.2e}  worst sequences ' + ' '.join(f'{i}:{x:.1e}' for x, i in top))
lib.call('ha_tune_set', b'rollout_persist', 1)
lib.call('ha_tune_set', b'rollout_persist_bwd', 1)
