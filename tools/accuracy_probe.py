"""Forward / gradient distance of the roll-out paths from the fp64 oracle on the inputs of tests/rollout_checks.check_rollout (model seed =
case seed, inputs from seed + 5).  Stand-alone (runs inside archived trees of older commits too: tools/microbench/bisect/<sha>).
usage: python tools/accuracy_probe.py [--cases 32x12x32,4x10x4,...] [--paths persistent,chain] [--contractive]   (test infrastructure: imports oracle/)"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', default='32x12x32,4x10x4,3x5x29,4x6x0,8x8x1,16x8x2,32x8x4')
    ap.add_argument('--paths', default='persistent,chain')
    ap.add_argument('--contractive', action='store_true')
    a = ap.parse_args()
    import rollout_checks as RC
    from humor_amd import _lib
    from oracle import humor_restated as H
    lib = _lib.get_lib()
    dev = torch.device('cuda:0')
    knobs = {'persistent': (1, 1), 'mixed': (1, 0), 'chain': (0, 0)}
    tot = {p: [0.0, 0.0, 0.0] for p in a.paths.split(',')}
    for case in a.cases.split(','):
        B, S, seed = (int(x) for x in case.split('x'))
        hm, sd = RC.make_model(lib, dev, seed=seed, contractive=True) if a.contractive else RC.make_model(lib, dev, seed=seed)
        g = torch.Generator().manual_seed(seed + 5)
        past_c, z_c = RC.canonical_state(B, g), torch.randn(B, S, 48, generator=g)
        gw = torch.randn(B, S, 348, generator=g)
        gm, gv = torch.randn(B, S, 48, generator=g), torch.randn(B, S, 48, generator=g)
        obj = lambda w, m, v: (w * gw.to(w)).sum() + (m * gm.to(m)).sum() + (v * gv.to(v)).sum()
        res = {}
        for dt in (torch.float32, torch.float64):
            p, zz = past_c.to(dt).clone().requires_grad_(True), z_c.to(dt).clone().requires_grad_(True)
            w, (pm, pv) = H.roll_out({k: v.to(dt) for k, v in sd.items()}, p, zz)
            obj(w, pm, pv).backward()
            res[dt] = (w.detach().double(), p.grad.double(), zz.grad.double())
        w64, gp64, gz64 = res[torch.float64]
        sc = (max(1.0, gp64.abs().max().item()), max(1.0, gz64.abs().max().item()))
        line = f'{B}x{S} seed {seed}: oracle32 fwd {(res[torch.float32][0] - w64).abs().max().item():.1e} g {max((res[torch.float32][1] - gp64).abs().max().item() / sc[0], (res[torch.float32][2] - gz64).abs().max().item() / sc[1]):.1e}'
        for path in a.paths.split(','):
            lib.call('ha_tune_set', b'rollout_persist', knobs[path][0])
            lib.call('ha_tune_set', b'rollout_persist_bwd', knobs[path][1])
            p, zz = past_c.to(dev).requires_grad_(True), z_c.to(dev).requires_grad_(True)
            out, (pm, pv) = hm.roll_out(p, None, S, z_seq=zz, return_prior=True)
            w = RC.world_of(out)
            obj(w, pm, pv).backward()
            ef = (w.detach().cpu().double() - w64).abs().max().item()
            # the first three steps (before the chain amplifies anything)
            e3 = (w.detach().cpu().double() - w64)[:, :3].abs().max().item()
            eg = max((p.grad.cpu().double() - gp64).abs().max().item() / sc[0], (zz.grad.cpu().double() - gz64).abs().max().item() / sc[1])
            line += f' | {path}: fwd {ef:.1e} (first 3 steps {e3:.1e}) g {eg:.1e}'
            tot[path][0] += ef; tot[path][1] += e3; tot[path][2] += eg
        print(line, flush=True)
    print('SUM ' + ' | '.join(f'{p}: fwd {v[0]:.2e} first3 {v[1]:.2e} g {v[2]:.2e}' for p, v in tot.items()))


if __name__ == '__main__':
    main()
