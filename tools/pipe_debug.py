#!/usr/bin/env python
"""Pipelined roll-out (rollout_pipe.inc, 32 < B <= 256) against the launch chain on the same inputs.
  pipe_debug.py fwd  B S      world / prior outputs, per (row tile, team, step, channel block) error map; with a -DHA_PERSIST_DEBUG build
                              (HUMOR_AMD_LIB=tools/microbench/libhumor_amd_pdebug.so) also every pre-activation slab of step 0 against PyTorch
  pipe_debug.py grad B S      gradients: pipelined forward + launch-chain adjoint, pipelined forward + pipelined adjoint, vs the chain
  pipe_debug.py time B S      event-timed forward and forward + backward of the paths"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch                                        # noqa: E402
from humor_amd import _lib                          # noqa: E402
import rollout_checks as RC                         # noqa: E402

dev = torch.device('cuda:0')
lib = _lib.get_lib()
BLOCKS = [('trans', 0, 3), ('trans_vel', 3, 6), ('root_R', 6, 15), ('root_vel', 15, 18), ('body_R', 18, 207), ('joints', 207, 273), ('joints_vel', 273, 339),
          ('contacts', 339, 348)]


def tune(**kw):
    for k, v in kw.items():
        lib.call('ha_tune_set', k.encode(), v)


def inputs(B, S, seed=0):
    g = torch.Generator().manual_seed(100 + B + S + seed)
    return RC.canonical_state(B, g).to(dev), torch.randn(B, S, 48, generator=g).to(dev), g


def find_stash(t):
    seen, stack = set(), [t.grad_fn]
    while stack:
        n = stack.pop()
        if n is None or id(n) in seen:
            continue
        seen.add(id(n))
        if hasattr(n, 'stash'):
            return n.stash
        stack.extend(x[0] for x in n.next_functions)
    return None


def fwd(B, S):
    hm, _ = RC.make_model(lib, dev, seed=0, contractive=True)
    past, z, g = inputs(B, S)
    res = {}
    for name, knobs in (('chain', dict(rollout_persist=0)), ('pipe', dict(rollout_persist=1, rollout_pipe=1, rollout_pipe_bwd=0))):
        tune(**knobs)
        p, zz = past.clone().requires_grad_(True), z.clone().requires_grad_(True)
        out, (pm, pv) = hm.roll_out(p, None, S, z_seq=zz, return_prior=True)
        torch.cuda.synchronize()
        res[name] = (RC.world_of(out).detach(), pm.detach(), pv.detach(), find_stash(pm))
    tune(rollout_persist=1)
    print('persist status (available, error word, launches):', RC.persist_status(lib, hm, dev))
    a, b = res['chain'], res['pipe']
    for nm, x, y in zip(('world', 'prior_mu', 'prior_var'), a[:3], b[:3]):
        print(f'{nm}: max abs err {(x - y).abs().max().item():.3e} (scale {x.abs().max().item():.3e}) finite {bool(torch.isfinite(y).all())}')
    err = (a[0] - b[0]).abs()          # [B, S, 348]
    err = torch.nan_to_num(err, nan=1e9)
    RT = (B + 31) // 32
    print('worst error per step:', ' '.join(f'{v:.1e}' for v in err.amax(dim=(0, 2))[:12].tolist()), '...' if S > 12 else '')
    print('worst error per channel block (step 0 | all steps):')
    for nm, lo, hi in BLOCKS:
        print(f'   {nm:11s} {err[:, 0, lo:hi].max().item():.2e} | {err[:, :, lo:hi].max().item():.2e}')
    print('worst error per (row tile, team) at step 0 / all steps:')
    for tile in range(RT):
        row = []
        for team in range(8):
            r0 = 32 * tile + 4 * team
            if r0 >= B:
                row.append('   -   ')
                continue
            e = err[r0:min(r0 + 4, B)]
            row.append(f'{e[:, 0].max().item():.0e}/{e.max().item():.0e}')
        print(f'   tile {tile}: ' + ' '.join(row))
    # ---- slabs of step 0 (debug build only) ----------------------------------------------------------------------------------
    f = getattr(lib._dll, 'ha_debug_persist_layout', None)
    if f is None:
        print('(no ha_debug_persist_layout in this build: slab comparison skipped)')
        return
    tune(rollout_persist=1, rollout_pipe=1)
    h = hm._net_handle(dev)
    lay = (C.c_int64 * 15)()
    f.restype = C.c_int
    assert f(h.ptr, B, S, lay) == 0
    xT, steps, per_step, off_G, d0, d1, d2, d3, gn0, gn1, gn2, off_gl, dz_part, single, pws = list(lay)
    print('layout', list(lay))
    st = b[3].cpu()
    dec = hm.decoder
    mods = [m.cpu() for m in dec.net]
    for t in range(min(S, 2)):
        # input state of step t from the pipelined run's own xT slabs
        xs = st[xT + t * RT * 340 * 32: xT + (t + 1) * RT * 340 * 32].reshape(RT, 85, 32, 4).permute(0, 2, 1, 3).reshape(RT * 32, 340)[:B, :339]
        if t == 0:
            print('xT[0] vs past_in0:', (xs - past.cpu()).abs().max().item())
        zc = z[:, t].cpu()
        hs, a_ = [], torch.cat([xs, zc], 1)
        k = 0
        for mod in mods:
            if isinstance(mod, torch.nn.Linear):
                if k > 0:
                    a_ = torch.cat([a_, zc], 1)
                a_ = mod(a_)
                hs.append(a_.detach())
                k += 1
            else:
                a_ = mod(a_)
        stp = st[steps + t * per_step: steps + (t + 1) * per_step]
        for l, (off, Cn, pad) in enumerate(((d0, 1024, 1024), (d1, 1024, 1024), (d2, 512, 512), (d3, 216, 224))):
            sl = stp[off: off + RT * pad * 32].reshape(RT, pad // 4, 32, 4).permute(0, 2, 1, 3).reshape(RT * 32, pad)[:B, :Cn]
            e = (sl - hs[l]).abs()
            e = torch.nan_to_num(e, nan=1e9)
            # which column blocks of 16 / which rows are off
            badc = (e.amax(0) > 1e-3).nonzero().flatten()
            badr = (e.amax(1) > 1e-3).nonzero().flatten()
            print(f'step {t} layer {l} slab: max err {e.max().item():.3e} (scale {hs[l].abs().max().item():.2e}); bad columns {badc.numel()} '
                  f'{badc[:12].tolist()}{"..." if badc.numel() > 12 else ""}; bad rows {badr.numel()} {badr[:16].tolist()}')


def grad(B, S):
    hm, _ = RC.make_model(lib, dev, seed=0, contractive=True)
    past, z, g = inputs(B, S)
    gw = torch.randn(B, S, 348, generator=g).to(dev)
    gm, gv = torch.randn(B, S, 48, generator=g).to(dev), torch.randn(B, S, 48, generator=g).to(dev)
    res = {}
    paths = (('chain', dict(rollout_persist=0)), ('pipe+chain', dict(rollout_persist=1, rollout_pipe=1, rollout_pipe_bwd=0)),
             ('pipe+pipe', dict(rollout_persist=1, rollout_pipe=1, rollout_pipe_bwd=1)))
    for name, knobs in paths:
        tune(**knobs)
        p, zz = past.clone().requires_grad_(True), z.clone().requires_grad_(True)
        try:
            out, (pm, pv) = hm.roll_out(p, None, S, z_seq=zz, return_prior=True)
            w = RC.world_of(out)
            ((w * gw).sum() + (pm * gm).sum() + (pv * gv).sum()).backward()
            torch.cuda.synchronize()
            res[name] = (w.detach(), p.grad.clone(), zz.grad.clone())
        except Exception as ex:
            print(name, 'failed:', repr(ex)[:300])
    tune(rollout_persist=1, rollout_pipe_bwd=0)
    print('persist status:', RC.persist_status(lib, hm, dev))
    ref = res['chain']
    for name in ('pipe+chain', 'pipe+pipe'):
        if name not in res:
            continue
        r = res[name]
        for nm, x, y in zip(('world', 'g_past', 'g_z'), ref, r):
            e = torch.nan_to_num((x - y).abs(), nan=1e9).reshape(B, -1).amax(1) / max(1.0, x.abs().max().item())
            worst = e.argsort(descending=True)[:6]
            print(f'{name} {nm}: worst relative error per sequence {e.max().item():.3e}; sequences over 3e-4: {int((e > 3e-4).sum())} of {B}; worst rows {[(int(i), float("%.1e" % e[i])) for i in worst]}')
        if name == 'pipe+pipe':
            ez = torch.nan_to_num((ref[2] - r[2]).abs(), nan=1e9)
            print('   g_z worst error per step:', ' '.join(f'{v:.1e}' for v in ez.amax(dim=(0, 2))[:16].tolist()))
            print('   g_z worst error per step (last):', ' '.join(f'{v:.1e}' for v in ez.amax(dim=(0, 2))[-8:].tolist()))


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def time_(B, S):
    hm, _ = RC.make_model(lib, dev, seed=0, contractive=True)
    past, z, g = inputs(B, S)
    gw = torch.randn(B, S, 348, generator=g).to(dev)
    for name, knobs in (('chain', dict(rollout_persist=0)), ('pipe fwd + chain bwd', dict(rollout_persist=1, rollout_pipe=1, rollout_pipe_bwd=0)),
                        ('pipe fwd + pipe bwd', dict(rollout_persist=1, rollout_pipe=1, rollout_pipe_bwd=1))):
        tune(**knobs)

        def f_only(prior=True):
            with torch.no_grad():
                hm.roll_out(past, None, S, z_seq=z, return_prior=prior)

        def f_b():
            p, zz = past.clone().requires_grad_(True), z.clone().requires_grad_(True)
            out, (pm, pv) = hm.roll_out(p, None, S, z_seq=zz, return_prior=True)
            ((RC.world_of(out) * gw).sum() + pm.sum() + pv.sum()).backward()
        try:
            print(f'{name:22s} B={B} S={S}: forward {timeit(f_only):.3f} ms (without the prior {timeit(lambda: f_only(False)):.3f}), forward + backward {timeit(f_b):.3f} ms')
        except Exception as ex:
            print(name, 'failed:', repr(ex)[:200])
    tune(rollout_persist=1, rollout_pipe_bwd=0)
    print('persist status:', RC.persist_status(lib, hm, dev))


if __name__ == '__main__':
    what, B, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    {'fwd': fwd, 'grad': grad, 'time': time_}[what](B, S)
