"""Where a sequence's gradient differs from the fp64 oracle: per path, the worst entries of dL/dpast and dL/dz of the given rows.
usage: python tools/row_detail.py 130x2x130 43,93 [--contractive]      (inputs of tests/rollout_checks.check_rollout; test infrastructure)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import rollout_checks as RC          # noqa: E402
from humor_amd import _lib           # noqa: E402
from oracle import humor_restated as H   # noqa: E402

B, S, seed = (int(x) for x in sys.argv[1].split('x'))
rows = [int(x) for x in sys.argv[2].split(',')]
contractive = '--contractive' in sys.argv
lib = _lib.get_lib()
dev = torch.device('cuda:0')
hm, sd = RC.make_model(lib, dev, seed=seed, contractive=True) if contractive else RC.make_model(lib, dev, seed=seed)
g = torch.Generator().manual_seed(seed + 5)
past_c, z_c = RC.canonical_state(B, g), torch.randn(B, S, 48, generator=g)
gw = torch.randn(B, S, 348, generator=g)
gm, gv = torch.randn(B, S, 48, generator=g), torch.randn(B, S, 48, generator=g)
obj = lambda w, m, v: (w * gw.to(w)).sum() + (m * gm.to(m)).sum() + (v * gv.to(v)).sum()
_, _, _, g64 = RC.oracle_grads(sd, past_c, z_c, obj)
_, _, _, g32 = RC.oracle_grads(sd, past_c, z_c, obj, dtype=torch.float32)
sc = [max(1.0, a.abs().max().item()) for a in g64]
print('scales', sc)
res = {'oracle32': [x.double() for x in g32]}
for path, (kf, kb) in (('persistent', (1, 1)), ('mixed', (1, 0)), ('chain', (0, 0))):
    lib.call('ha_tune_set', b'rollout_persist', kf)
    lib.call('ha_tune_set', b'rollout_persist_bwd', kb)
    p, zz = past_c.to(dev).requires_grad_(True), z_c.to(dev).requires_grad_(True)
    out, (pm, pv) = hm.roll_out(p, None, S, z_seq=zz, return_prior=True)
    obj(RC.world_of(out), pm, pv).backward()
    res[path] = [p.grad.cpu().double(), zz.grad.cpu().double()]
for r in rows:
    for name, gr in res.items():
        line = f'row {r} {name:10s}'
        for i, nm in enumerate(('g_past', 'g_z')):
            e = (gr[i][r] - g64[i][r]).abs().reshape(-1) / sc[i]
            top = torch.topk(e, 3)
            line += f' | {nm} max {e.max().item():.1e} at {top.indices.tolist()} ref {[round(g64[i][r].reshape(-1)[j].item(), 3) for j in top.indices.tolist()]} got {[round(gr[i][r].reshape(-1)[j].item(), 3) for j in top.indices.tolist()]}'
        print(line)
