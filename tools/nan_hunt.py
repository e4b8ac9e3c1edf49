"""Round-6 regression hunt: roll-out forward + backward at small (B, S) on the three paths with every torch.empty of the process
poisoned (NaN / huge / zero), compared with the oracle.  Prints which outputs hold non-finite values and where.
usage (GPU box):  python tools/nan_hunt.py [--poison nan|big|none] [--grid 8] [--paths persistent,mixed,chain]
(test infrastructure: imports oracle/)"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

_real_empty = torch.empty


def install_poison(kind):
    if kind == 'none':
        return

    def poisoned(*a, **k):
        t = _real_empty(*a, **k)
        if t.is_floating_point() and t.is_cuda:
            t.fill_(float('nan') if kind == 'nan' else 3.0e30)
        return t
    torch.empty = poisoned


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--poison', default='nan')
    ap.add_argument('--grid', type=int, default=8)
    ap.add_argument('--cases', default='')
    ap.add_argument('--paths', default='persistent,mixed,chain')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--reps', type=int, default=1)
    ap.add_argument('--cu-poison', type=lambda v: int(v, 0), default=0, help='ha_tune_set cu_poison (1 = NaN, else the bit pattern)')
    ap.add_argument('--smoke', action='store_true', help='the input of tests/smoke_rollout.py instead of the grid')
    a = ap.parse_args()
    install_poison(a.poison)
    import rollout_checks as RC
    from humor_amd import _lib
    from oracle import humor_restated as H
    lib = _lib.get_lib()
    dev = torch.device('cuda:0')
    if a.cu_poison:
        import ctypes as C
        n = C.c_uint()
        lib.call('ha_debug_cu_poison', a.cu_poison, C.byref(n), None)
        print(f'cu_poison probe: {n.value} of {256 * 40960} LDS words still hold the pattern in the next kernel', flush=True)
        lib.call('ha_tune_set', b'cu_poison', a.cu_poison)
    knobs = {'persistent': (1, 1), 'mixed': (1, 0), 'chain': (0, 0)}
    if a.cases:
        cases = [tuple(int(x) for x in c.split('x')) for c in a.cases.split(',')]
    else:
        cases = [(b, s) for b in range(1, a.grid + 1) for s in range(1, a.grid + 1)]
    hm, sd = RC.make_model(lib, dev, seed=a.seed)
    bad = 0
    if a.smoke:
        cases = [(4, 6)]
    sd64 = {k: v.double() for k, v in sd.items()}
    for B, S in cases:
        g = torch.Generator().manual_seed(0 if a.smoke else 1000 * B + S + a.seed)
        past_c = RC.canonical_state(B, g).requires_grad_(True)
        z_c = torch.randn(B, S, 48, generator=g).requires_grad_(True)
        gw = None if a.smoke else torch.randn(B, S, 348, generator=g)
        objective = (lambda w, pm, pv: w.square().sum() + pm.sum()) if a.smoke else (lambda w, pm, pv: (w * gw.to(w)).sum() + pm.sum() + pv.sum())
        w_ref, (pm_r, pv_r) = H.roll_out(sd, past_c, z_c)
        objective(w_ref, pm_r, pv_r).backward()
        if a.smoke:
            p64, z64 = past_c.detach().double().requires_grad_(True), z_c.detach().double().requires_grad_(True)
            w64, (pm64, pv64) = H.roll_out(sd64, p64, z64)
            objective(w64, pm64, pv64).backward()
            print('oracle fp32 vs fp64: g_past %.2e g_z %.2e (relative to the largest entry)' % (
                (past_c.grad.double() - p64.grad).abs().max().item() / max(1.0, p64.grad.abs().max().item()),
                (z_c.grad.double() - z64.grad).abs().max().item() / max(1.0, z64.grad.abs().max().item())))
        gscale = (max(1.0, past_c.grad.abs().max().item()), max(1.0, z_c.grad.abs().max().item()))
        for path in a.paths.split(','):
            lib.call('ha_tune_set', b'rollout_persist', knobs[path][0])
            lib.call('ha_tune_set', b'rollout_persist_bwd', knobs[path][1])
            for rep in range(a.reps):
                past = past_c.detach().to(dev).requires_grad_(True)
                z = z_c.detach().to(dev).requires_grad_(True)
                out, (pm, pv) = hm.roll_out(past, None, S, z_seq=z, return_prior=True)
                world = RC.world_of(out)
                objective(world, pm, pv).backward()
                torch.cuda.synchronize()
                st = RC.persist_status(lib, hm, dev)
                rows = []
                for name, got, ref, sc in (('world', world, w_ref, 1.0), ('pm', pm, pm_r, 1.0), ('pv', pv, pv_r, 1.0),
                                           ('g_past', past.grad, past_c.grad, gscale[0]), ('g_z', z.grad, z_c.grad, gscale[1])):
                    got = got.detach().cpu()
                    nf = (~torch.isfinite(got)).nonzero()
                    e = ((got - ref.detach()).abs() / sc)
                    e = e[torch.isfinite(e)].max().item() if torch.isfinite(e).any() else float('nan')
                    rows.append((name, nf.shape[0], e, nf[:3].tolist()))
                flag = any(r[1] for r in rows) or rows[0][2] > 1e-4 or rows[3][2] > 1e-3 or rows[4][2] > 1e-3
                bad += bool(flag)
                if flag or a.smoke or (B, S) in ((4, 6), (1, 1)):
                    print(f'{"BAD" if flag else "ok "} B={B} S={S} {path:10s} rep{rep} status={st} ' +
                          ' '.join(f'{n}:nf={c},err={e:.1e}{(",at=" + str(w)) if c else ""}' for n, c, e, w in rows), flush=True)
    print(f'nan_hunt poison={a.poison}: {bad} bad of {len(cases) * len(a.paths.split(",")) * a.reps}')


if __name__ == '__main__':
    main()
