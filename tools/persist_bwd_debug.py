"""Isolates parts of the persistent adjoint: dL/dworld restricted to channel ranges, short roll-outs; persistent vs launch chain."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from humor_amd import _lib
import rollout_checks as RC

dev = torch.device('cuda:0')
lib = _lib.get_lib()
hm, _ = RC.make_model(lib, dev, seed=0, contractive=True)
SEG = {'trans': (0, 3), 'tvel': (3, 6), 'rootR': (6, 15), 'rvel': (15, 18), 'bodyR': (18, 207), 'joints': (207, 273), 'jvel': (273, 339), 'contacts': (339, 348)}
PSEG = {k: v for k, v in SEG.items() if k != 'contacts'}


def run(B, S, mask, prior):
    g = torch.Generator().manual_seed(3)
    past, z = RC.canonical_state(B, g).to(dev), torch.randn(B, S, 48, generator=g).to(dev)
    gw = torch.randn(B, S, 348, generator=g).to(dev)
    m = torch.zeros(348, device=dev)
    for k in mask:
        m[SEG[k][0]:SEG[k][1]] = 1
    gw = gw * m
    gm, gv = torch.randn(B, S, 48, generator=g).to(dev), torch.randn(B, S, 48, generator=g).to(dev)
    out = []
    for bwd in (0, 1):
        lib.call('ha_tune_set', b'rollout_persist_bwd', bwd)
        p, zz = past.clone().requires_grad_(True), z.clone().requires_grad_(True)
        o, (pm, pv) = hm.roll_out(p, None, S, z_seq=zz, return_prior=True)
        loss = (RC.world_of(o) * gw).sum()
        if prior:
            loss = loss + (pm * gm).sum() + (pv * gv).sum()
        loss.backward()
        out.append((p.grad.clone(), zz.grad.clone()))
    lib.call('ha_tune_set', b'rollout_persist_bwd', 1)
    (gp0, gz0), (gp1, gz1) = out
    sc = max(1.0, gp0.abs().max().item())
    line = 'B=%d S=%d gw on %-28s prior=%d | g_z err %.2e (scale %.2e) per step %s | g_past err:' % (
        B, S, ','.join(mask), prior, (gz0 - gz1).abs().max().item(), gz0.abs().max().item(), ['%.1e' % (gz0[:, t] - gz1[:, t]).abs().max().item() for t in range(S)])
    for k, (a, b) in PSEG.items():
        line += ' %s %.1e' % (k, (gp0[:, a:b] - gp1[:, a:b]).abs().max().item() / sc)
    print(line, flush=True)

for B, S, mask, prior in ((4, 2, ['jvel'], 0), (4, 2, ['trans'], 0), (4, 2, ['tvel'], 0), (4, 2, ['rvel'], 0), (4, 2, ['rootR'], 0), (4, 2, ['bodyR'], 0), (4, 1, ['contacts'], 0), (4, 1, ['jvel'], 0), (4, 1, ['joints'], 0), (4, 1, ['bodyR'], 0), (4, 1, ['rvel'], 0), (4, 1, ['tvel'], 0),
                          (4, 1, ['trans'], 0), (4, 1, ['rootR'], 0), (4, 1, [], 1), (4, 2, ['contacts'], 0), (4, 2, ['joints'], 0), (4, 2, list(SEG), 1), (32, 3, list(SEG), 1)):
    run(B, S, mask, prior)
