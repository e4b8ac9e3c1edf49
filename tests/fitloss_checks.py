"""The fused fitting-objective kernel (ha_fit_loss) against the term-by-term PyTorch evaluation of FittingLoss, which the CPU
tier pins bit-for-bit to the reference's FittingLoss (tests/test_fitting_cpu.py): loss, every term value and every gradient."""
import numpy as np
import torch

from humor_amd import synth
from humor_amd.fitting_loss import FittingLoss
from humor_amd.tables import OP_IGNORE_JOINTS, SMPLH_TO_OPENPOSE25
from oracle import closure_cases as CC

ALL_ON = {k: 0.3 + 0.1 * i for i, k in enumerate(CC.KEYS)}
ALL_ON['points3d'] = 0.0


def make_inputs(B, T, device, seed=0, halo=False, nj=73, nv=43):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(device)
    jtr = r(B, T, nj, 3, sc=0.5)
    jtr[..., 2] += 4.0
    diff = {'cam_jtr': jtr, 'cam_verts': r(B, T, nv, 3, sc=0.5), 'pri_jtr': r(B, T, nj, 3, sc=0.3), 'ro_joints': r(B, T, 22, 3, sc=0.3),
            'contacts_conf': torch.rand(B, T, 22, generator=g).to(device), 'latent_pose': r(B, T, 32), 'betas': r(B, 16),
            'latent_motion': r(B, T - 1, 48), 'prior_mu': r(B, T - 1, 48, sc=0.5), 'prior_var': (0.5 + torch.rand(B, T - 1, 48, generator=g)).to(device),
            'floor': r(B, 3, sc=0.3), 'joints_vel': r(B, 1, 22, 3, sc=0.3), 'trans_vel': r(B, 1, 3, sc=0.3), 'root_orient_vel': r(B, 1, 3, sc=0.3)}
    if halo:
        diff.update(prev_tail=r(T, nv, 3, sc=0.5), prev_betas=r(16), prev_floor=r(3, sc=0.3))
    diff = {k: v.requires_grad_(True) for k, v in diff.items()}
    xy = torch.rand(B, T, 25, 2, generator=g) * torch.tensor([1900.0, 1000.0])
    obs = {'joints2d': torch.cat([xy, torch.rand(B, T, 25, 1, generator=g)], 3).to(device), 'joints3d': r(B, T, 22, 3, sc=0.5),
           'verts3d': r(B, T, nv, 3, sc=0.5), 'floor_plane': torch.tensor([[0.1, -1.0, 0.05, -0.5]]).expand(B, 4).contiguous().to(device)}
    obs['joints3d'][0, 1, 4, 1] = float('inf')
    obs['verts3d'][-1, 0, 2] = float('-inf')
    ov = min(3, T - 1)
    nrow = B + (1 if halo else 0)
    obs['seq_interval'] = torch.tensor([[b * (T - ov), b * (T - ov) + T] for b in range(nrow)])
    return diff, obs


def dicts(diff, halo):
    jtr, pj = diff['cam_jtr'], diff['pri_jtr']
    cam = {'joints3d': jtr[:, :, :22], 'joints3d_extra': jtr[:, :, 22:], 'jtr': jtr, 'verts3d': diff['cam_verts'], 'latent_pose': diff['latent_pose'],
           'betas': diff['betas'], 'floor_plane': diff['floor']}
    pred = {'joints3d': pj[:, :, :22], 'joints3d_extra': pj[:, :, 22:], 'jtr': pj, 'latent_motion': diff['latent_motion'],
            'joints_vel': diff['joints_vel'], 'trans_vel': diff['trans_vel'], 'root_orient_vel': diff['root_orient_vel'],
            'joints3d_rollout': diff['ro_joints'], 'contacts_conf': diff['contacts_conf']}
    h = None
    if halo:
        h = {'first': False, 'prev_tail': diff['prev_tail'], 'prev_betas': diff['prev_betas'], 'prev_floor': diff['prev_floor']}
    return cam, pred, h


def evaluate(fl, kind, diff, obs, halo, cond=True):
    cam, pred, h = dicts(diff, halo)
    if kind == 'root':
        loss, stats = fl.root_fit(obs, cam, halo=h)
    elif kind == 'smpl':
        loss, stats = fl.smpl_fit(obs, cam, 7, halo=h)
    else:
        cp = (diff['prior_mu'], diff['prior_var']) if cond else None
        loss, stats = fl.motion_fit(obs, pred, cam, 7, cond_prior=cp, init_motion_scale=1.7, halo=h)
    names = list(diff.keys())
    grads = torch.autograd.grad(loss, [diff[k] for k in names], allow_unused=True)
    return loss.detach(), {k: v.detach() for k, v in stats.items()}, dict(zip(names, grads))


def check_fused_vs_terms(lib, device, B=3, T=7, seed=0):
    w, mu, cov = synth.make_gmm(seed=0)
    gmm = {'gmm': (w.to(device), mu.to(device), cov.to(device))}
    cam_f = torch.tensor([[1060.5, 1060.4]]).expand(B, 2).to(device)
    cam_c = torch.tensor([[951.3, 536.8]]).expand(B, 2).to(device)
    mk = lambda fused: FittingLoss([ALL_ON] * 3, gmm, SMPLH_TO_OPENPOSE25, OP_IGNORE_JOINTS, cam_f, cam_c, 'bisquare', joints2d_sigma=100,
                                   fused=fused, _lib_override=lib)
    fused, terms = mk(True), mk(False)
    worst = 0.0
    for halo in (False, True):
        diff, obs = make_inputs(B, T, device, seed=seed + (1 if halo else 0), halo=halo)
        for kind, cond in (('root', True), ('smpl', True), ('motion', True), ('motion', False)):
            l1, s1, g1 = evaluate(fused, kind, diff, obs, halo, cond)
            l0, s0, g0 = evaluate(terms, kind, diff, obs, halo, cond)
            assert abs(l1.item() - l0.item()) <= 2e-5 * abs(l0.item()), (kind, halo, l1.item(), l0.item())
            assert set(s1.keys()) == set(s0.keys()), (kind, sorted(s1.keys()), sorted(s0.keys()))
            for k in s0:
                a, b_ = float(s1[k]), float(s0[k])
                assert abs(a - b_) <= 2e-5 * max(1.0, abs(b_)), (kind, halo, k, a, b_)
            for k in g0:
                if g0[k] is None:
                    assert g1[k] is None or float(g1[k].abs().max()) == 0.0, (kind, halo, k)
                    continue
                assert g1[k] is not None, (kind, halo, k)
                e = (g1[k] - g0[k]).abs().max().item() / max(1.0, g0[k].abs().max().item())
                worst = max(worst, e)
                assert e <= 1e-4, (kind, halo, cond, k, e)
    return worst
