import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from humor_amd import _lib
import rollout_checks as RC
from oracle import humor_restated as H
dev = torch.device('cuda:0')
lib = _lib.get_lib()
for B, S, seed in [(33, 5, 33), (32, 12, 32), (33, 5, 1), (32, 12, 1), (70, 3, 70)]:
    hm, sd = RC.make_model(lib, dev, seed=seed)
    g = torch.Generator().manual_seed(seed + 5)
    past_c = RC.canonical_state(B, g).requires_grad_(True); z_c = torch.randn(B, S, 48, generator=g).requires_grad_(True)
    w_ref, (pm_r, pv_r) = H.roll_out(sd, past_c, z_c)
    gw = torch.randn(w_ref.shape, generator=g); gm, gv = torch.randn(pm_r.shape, generator=g), torch.randn(pv_r.shape, generator=g)
    g_ref = torch.autograd.grad((w_ref * gw).sum() + (pm_r * gm).sum() + (pv_r * gv).sum(), [past_c, z_c])
    p64, z64 = past_c.detach().double().requires_grad_(True), z_c.detach().double().requires_grad_(True)
    w64, (pm64, pv64) = H.roll_out({k: v.double() for k, v in sd.items()}, p64, z64)
    g64 = torch.autograd.grad((w64 * gw.double()).sum() + (pm64 * gm.double()).sum() + (pv64 * gv.double()).sum(), [p64, z64])
    for mode in (0, 3):
        lib.call('ha_tune_set', b'rollout_persist', mode)
        past = past_c.detach().to(dev).requires_grad_(True); z = z_c.detach().to(dev).requires_grad_(True)
        out, (pm, pv) = hm.roll_out(past, None, S, z_seq=z, return_prior=True)
        w = RC.world_of(out)
        go = torch.autograd.grad((w * gw.to(dev)).sum() + (pm * gm.to(dev)).sum() + (pv * gv.to(dev)).sum(), [past, z])
        scale = [max(1.0, a.abs().max().item()) for a in g_ref]
        print(B, S, seed, 'mode', mode, 'fwd vs cpu32 %.2e vs fp64 %.2e (cpu32 vs fp64 %.2e)' % ((w.detach().cpu() - w_ref).abs().max().item(), (w.detach().cpu().double() - w64).abs().max().item(), (w_ref.double() - w64).abs().max().item()),
              '| grads rel to scale: ' + ' '.join('%s vs cpu32 %.2e vs fp64 %.2e (cpu32 vs fp64 %.2e)' % (n, (a - b.cpu()).abs().max().item() / sc, (c - b.cpu().double()).abs().max().item() / sc, (c - a.double()).abs().max().item() / sc)
                                                  for n, a, b, c, sc in zip(('g_past', 'g_z'), g_ref, go, g64, scale)), flush=True)
