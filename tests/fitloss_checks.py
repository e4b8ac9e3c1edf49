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


def check_rollout_post(lib, device, B=3, S=6, seed=0, cam=True):
    """ha_rollout_post (one kernel per direction) against the op-by-op chain it replaces in MotionOptimizer.rollout_latent_motion
    (R -> axis-angle kernels, cats, sigmoid / index_add contacts, apply_cam2prior(inverse=True)): every output and every gradient."""
    from humor_amd import ops
    from humor_amd.fit_kernels import RolloutPost
    from humor_amd.tables import CONTACT_INDS
    from oracle import lbs_restated as L
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: sc * torch.randn(*s, generator=g)
    T = S + 1
    world = r(B, S, 348, sc=0.5)
    world[:, :, 6:15] = L.batch_rodrigues(r(B * S, 3, sc=1.5)).reshape(B, S, 9)
    world[:, :, 18:207] = L.batch_rodrigues(r(B * S * 21, 3, sc=0.8)).reshape(B, S, 189)
    c2p_R = L.batch_rodrigues(r(B, 3, sc=1.0)).reshape(B, 3, 3) if cam else None
    base = dict(world=world, trans0=r(B, 3), root0=r(B, 3, sc=0.8), pose0=r(B, 63, sc=0.4), joints0=r(B, 22, 3), c2p_R=c2p_R, c2p_t=r(B, 3) if cam else None)

    def leaves():
        return {k: (None if v is None else v.clone().to(device).requires_grad_(True)) for k, v in base.items()}

    def reference(x):
        aa_root = ops.rotation_matrix_to_angle_axis(x['world'][:, :, 6:15].reshape(-1, 3, 3), _lib_override=lib).reshape(B, S, 3)
        aa_body = ops.rotation_matrix_to_angle_axis(x['world'][:, :, 18:207].reshape(-1, 3, 3), _lib_override=lib).reshape(B, S, 63)
        trans = torch.cat([x['trans0'].reshape(B, 1, 3), x['world'][:, :, 0:3]], 1)
        root = torch.cat([x['root0'].reshape(B, 1, 3), aa_root], 1)
        pose = torch.cat([x['pose0'].reshape(B, 1, 63), aa_body], 1)
        joints = torch.cat([x['joints0'].reshape(B, 1, 22, 3), x['world'][:, :, 207:273].reshape(B, S, 22, 3)], 1)
        conf9 = torch.sigmoid(x['world'][:, :, 339:348])
        idx = torch.as_tensor(CONTACT_INDS, dtype=torch.long, device=device)
        conf = torch.zeros(B, S, 22, device=device).index_add(2, idx, conf9)
        lab = torch.zeros(B, S, 22, device=device).index_add(2, idx, (conf9 > 0.5).float())
        conf, lab = torch.cat([conf[:, :1], conf], 1), torch.cat([lab[:, :1], lab], 1)
        ct = cr = None
        if cam:
            Rm = ops.batch_rodrigues(root.reshape(-1, 3), _lib_override=lib).reshape(B, T, 3, 3)
            Rl = x['c2p_R'].unsqueeze(1).transpose(3, 2)
            newR = (Rl.unsqueeze(-1) * Rm.unsqueeze(-3)).sum(-2)
            cr = ops.rotation_matrix_to_angle_axis(newR.reshape(-1, 3, 3), _lib_override=lib).reshape(B, T, 3)
            ct = (Rl * (trans - trans[:, 0:1]).unsqueeze(-2)).sum(-1) - x['c2p_t'].unsqueeze(1)
        return trans, root, pose, joints, conf, lab, ct, cr

    xa, xb = leaves(), leaves()
    oa = RolloutPost.apply(lib, *[xa[k] for k in ('world', 'trans0', 'root0', 'pose0', 'joints0', 'c2p_R', 'c2p_t')])
    ob = reference(xb)
    names = ['trans', 'root_orient', 'pose_body', 'joints', 'contacts_conf', 'contacts', 'cam_trans', 'cam_root_orient']
    la = lb = 0.0
    for i, (n, a, b_) in enumerate(zip(names, oa, ob)):
        if b_ is None:
            assert a is None, n
            continue
        assert a.shape == b_.shape, (n, a.shape, b_.shape)
        assert (a.detach() - b_.detach()).abs().max().item() < 1e-6, (n, (a.detach() - b_.detach()).abs().max().item())
        if n != 'contacts':
            wgt = CC.det_weights(a.shape, 0.1 * (i + 1)).to(device)
            la, lb = la + (a * wgt).sum(), lb + (b_ * wgt).sum()
    ks = [k for k in base if base[k] is not None]
    ga = torch.autograd.grad(la, [xa[k] for k in ks])
    gb = torch.autograd.grad(lb, [xb[k] for k in ks])
    worst = 0.0
    for k, p, q in zip(ks, ga, gb):
        e = (p - q).abs().max().item() / max(1.0, q.abs().max().item())
        worst = max(worst, e)
        assert e < 1e-5, (k, e)
    return worst


def check_fit_pre(lib, device, npz, B=3, seed=0):
    """ha_fit_pre (cam2prior + key frame in the prior frame + initial roll-out state from ONE SMPL evaluation) against the op chain it
    replaces, which evaluates SMPL three times (compute_cam2prior -> apply_cam2prior -> initial joints): every output and the
    gradients w.r.t. floor, trans, root orientation, body pose, betas and the initial velocities."""
    from humor_amd import frames, ops
    from humor_amd.body_model import BodyModel
    from humor_amd.fit_kernels import FitPre
    from humor_amd.tables import KEYPT_VERTS
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: sc * torch.randn(*s, generator=g)
    base = dict(floor=torch.tensor([[0.02, 0.5, 0.03]]) + r(B, 3, sc=0.03), trans0=torch.cat([r(B, 2, sc=0.3), 4.0 + r(B, 1, sc=0.3)], 1),
                root0=torch.tensor([np.pi, 0.0, 0.0]) + r(B, 3, sc=0.3), pose0=r(B, 63, sc=0.3), betas=r(B, 16, sc=0.5),
                trans_vel=r(B, 3, sc=0.3), joints_vel=r(B, 22, 3, sc=0.3), root_orient_vel=r(B, 3, sc=0.3))
    bm = BodyModel(npz, num_betas=16, batch_size=B, use_vtx_selector=True, vertex_subset=KEYPT_VERTS, _lib_override=lib)
    smpl_j = lambda tr, ro, po, be: bm(pose_body=po, pose_hand=None, betas=be, root_orient=ro, trans=tr).Jtr[:, :22]

    def leaves():
        return {k: v.clone().to(device).requires_grad_(True) for k, v in base.items()}

    def fused(x):
        jcam = smpl_j(x['trans0'], x['root0'], x['pose0'], x['betas'])
        return FitPre.apply(lib, x['floor'], x['trans0'], x['root0'], x['pose0'], jcam, x['trans_vel'], x['joints_vel'], x['root_orient_vel'])

    def chain(x):
        jcam = smpl_j(x['trans0'], x['root0'], x['pose0'], x['betas'])
        Rc = ops.batch_rodrigues(x['root0'], _lib_override=lib)
        R, t, h = frames.compute_cam2prior(x['floor'], x['trans0'], Rc, jcam)
        root_p = ops.rotation_matrix_to_angle_axis(torch.matmul(R, Rc), _lib_override=lib)
        tr = torch.matmul(R, (x['trans0'] + t).unsqueeze(-1)).squeeze(-1)
        cur_h = smpl_j(tr, root_p, x['pose0'], x['betas'])[:, 0, 2:3]
        tr = tr + torch.cat([torch.zeros(B, 2, device=device), h - cur_h], 1)
        joints_p = smpl_j(tr, root_p, x['pose0'], x['betas'])
        R_root = ops.batch_rodrigues(root_p, _lib_override=lib).reshape(B, 9)
        R_body = ops.batch_rodrigues(x['pose0'].reshape(-1, 3), _lib_override=lib).reshape(B, 189)
        past_in = torch.cat([tr, x['trans_vel'], R_root, x['root_orient_vel'], R_body, joints_p.reshape(B, 66), x['joints_vel'].reshape(B, 66)], 1)
        return past_in, tr, root_p, joints_p, R, t, h

    xa, xb = leaves(), leaves()
    oa, ob = fused(xa), chain(xb)
    la = lb = 0.0
    for i, (a, b_) in enumerate(zip(oa, ob)):
        assert a.shape == b_.shape, (i, a.shape, b_.shape)
        assert (a.detach() - b_.detach()).abs().max().item() < 2e-5, (i, (a.detach() - b_.detach()).abs().max().item())
        wgt = CC.det_weights(a.shape, 0.3 * (i + 1)).to(device)
        la, lb = la + (a * wgt).sum(), lb + (b_ * wgt).sum()
    ks = list(base.keys())
    ga = torch.autograd.grad(la, [xa[k] for k in ks])
    gb = torch.autograd.grad(lb, [xb[k] for k in ks])
    worst = 0.0
    for k, p, q in zip(ks, ga, gb):
        e = (p - q).abs().max().item() / max(1.0, q.abs().max().item())
        worst = max(worst, e)
        assert e < 2e-4, (k, e, q.abs().max().item())
    return worst


def check_rigid_image(lib, device, npz, N=5, seed=0, dense=False):
    """ha_rigid_image (the SMPL output under a second root pose as the rigid image of the first evaluation) against evaluating the
    body model twice: both outputs and the gradients w.r.t. pose, shape and both root poses."""
    from humor_amd.body_model import BodyModel
    from humor_amd.fit_kernels import RigidImage
    from humor_amd.tables import KEYPT_VERTS
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: sc * torch.randn(*s, generator=g)
    base = dict(trans=torch.cat([r(N, 2, sc=0.5), 1.0 + r(N, 1, sc=0.3)], 1), root=r(N, 3, sc=0.8), pose=r(N, 63, sc=0.3), betas=r(N, 16, sc=0.5),
                trans2=torch.cat([r(N, 2, sc=0.3), 4.0 + r(N, 1, sc=0.3)], 1), root2=torch.tensor([np.pi, 0.0, 0.0]) + r(N, 3, sc=0.4))
    bm = BodyModel(npz, num_betas=16, batch_size=N, use_vtx_selector=True, vertex_subset=None if dense else KEYPT_VERTS, _lib_override=lib)
    smpl = lambda tr, ro, x: bm(pose_body=x['pose'], pose_hand=None, betas=x['betas'], root_orient=ro, trans=tr)

    def leaves():
        return {k: v.clone().to(device).requires_grad_(True) for k, v in base.items()}

    def moved(x):
        b = smpl(x['trans'], x['root'], x)
        j2, v2 = RigidImage.apply(lib, b.Jtr, b.v, x['root'], x['trans'], x['root2'], x['trans2'])
        return b.Jtr, b.v, j2, v2

    def twice(x):
        b, b2 = smpl(x['trans'], x['root'], x), smpl(x['trans2'], x['root2'], x)
        return b.Jtr, b.v, b2.Jtr, b2.v

    xa, xb = leaves(), leaves()
    oa, ob = moved(xa), twice(xb)
    la = lb = 0.0
    for i, (a, b_) in enumerate(zip(oa, ob)):
        assert a.shape == b_.shape, (i, a.shape, b_.shape)
        e = (a.detach() - b_.detach()).abs().max().item()
        assert e < 1e-5, (i, e)
        wgt = CC.det_weights(a.shape, 0.3 * (i + 1)).to(device)
        la, lb = la + (a * wgt).sum(), lb + (b_ * wgt).sum()
    ks = list(base.keys())
    ga = torch.autograd.grad(la, [xa[k] for k in ks])
    gb = torch.autograd.grad(lb, [xb[k] for k in ks])
    worst = 0.0
    for k, p, q in zip(ks, ga, gb):
        e = (p - q).abs().max().item() / max(1.0, q.abs().max().item())
        worst = max(worst, e)
        assert e < 1e-4, (k, e, q.abs().max().item())
    return worst


def check_gmm_nll(lib, device, B=5, seed=0):
    """ha_gmm_nll (init-state mixture, value + gradient) against the op-by-op log-density and its autograd: contiguous segments, frame 0
    of a [B,T,J,3] tensor read in place (strided rows), a single sequence."""
    from humor_amd.fit_kernels import GmmNll
    from humor_amd.fitting_loss import _GMM
    w, mu, cov = synth.make_gmm(seed=0)
    gmm = _GMM(w.to(device), mu.to(device), cov.to(device))
    g = torch.Generator().manual_seed(seed)
    worst = 0.0
    for b in (B, 1):
        jtr = (0.3 * torch.randn(b, 4, 30, 3, generator=g)).to(device).requires_grad_(True)
        jv = (0.3 * torch.randn(b, 1, 22, 3, generator=g)).to(device).requires_grad_(True)
        tv = (0.3 * torch.randn(b, 1, 3, generator=g)).to(device).requires_grad_(True)
        rv = (0.3 * torch.randn(b, 1, 3, generator=g)).to(device).requires_grad_(True)
        j0 = jtr[:, :, :22][:, 0:1]                                   # a view with contiguous 66-float rows, row stride 4 * 30 * 3
        nll = GmmNll.apply(lib, gmm, j0, jv, tv, rv)
        state = torch.cat([j0.reshape(b, -1), jv.reshape(b, -1), tv.reshape(b, -1), rv.reshape(b, -1)], -1)
        ref = -gmm.log_prob(state).sum()
        assert abs(nll.item() - ref.item()) <= 1e-6 * abs(ref.item()), (nll.item(), ref.item())
        g1 = torch.autograd.grad(nll, [jtr, jv, tv, rv])
        g0 = torch.autograd.grad(ref, [jtr, jv, tv, rv])
        for a, c in zip(g1, g0):
            e = (a - c).abs().max().item() / max(1.0, c.abs().max().item())
            worst = max(worst, e)
            assert e < 2e-5, e
    return worst



def check_backward_addends(lib, device, B=3, seed=0):
    """The optional addends of ha_rigid_image_backward (g_joints_add / g_verts_add) and ha_fit_pre_backward (add_floor / add_pose0 / add_*_vel,
    jcam read with a row stride): out = adjoint + addend -- compared with the same call without addends plus the additions."""
    import ctypes as C
    from humor_amd import _lib
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(device).contiguous()
    st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream if device.type == 'cuda' else 0)
    worst = 0.0
    # ---- rigid image ----
    N, J, V = B * 2, 73, 43
    t = dict(joints=r(N, J, 3), verts=r(N, V, 3), root=r(N, 3, sc=0.8), trans=r(N, 3), root2=r(N, 3, sc=0.8), trans2=r(N, 3),
             g_joints2=r(N, J, 3), g_verts2=r(N, V, 3))
    outs = []
    adds = dict(g_joints_add=r(N, J, 3), g_verts_add=r(N, V, 3))
    for with_add in (False, True):
        o = {k: torch.empty_like(t[s]) for k, s in (('g_joints', 'joints'), ('g_verts', 'verts'), ('g_root', 'root'), ('g_trans', 'trans'),
                                                    ('g_root2', 'root2'), ('g_trans2', 'trans2'))}
        a = _lib.RigidImageArgs()
        a.N, a.J, a.V = N, J, V
        for k, v in dict(**t, **o, **(adds if with_add else {})).items():
            setattr(a, k, v.data_ptr())
        lib.call('ha_rigid_image_backward', C.byref(a), st)
        outs.append(o)
    for k in outs[0]:
        want = outs[0][k] + (adds['g_joints_add'] if k == 'g_joints' else adds['g_verts_add'] if k == 'g_verts' else 0.0)
        e = (outs[1][k] - want).abs().max().item() / max(1.0, want.abs().max().item())
        worst = max(worst, e)
        assert e < 1e-6, ('rigid', k, e)
    # ---- fit_pre: jcam as the first 22 rows of a [B,73,3] joint tensor (stride 219), five addends ----
    jfull = r(B, 73, 3)
    ins = dict(floor=torch.tensor([[0.05, -1.0, 0.1]]).repeat(B, 1).to(device) * (1.0 + 0.1 * r(B, 1).abs()), trans0=r(B, 3), root0=r(B, 3, sc=0.5), pose0=r(B, 63, sc=0.3),
               trans_vel=r(B, 3), joints_vel=r(B, 22, 3), root_orient_vel=r(B, 3))
    gin = dict(g_past_in=r(B, 339), g_trans_p=r(B, 3), g_root_p=r(B, 3), g_joints_p=r(B, 22, 3), g_c2p_R=r(B, 3, 3), g_c2p_t=r(B, 3), g_root_height=r(B, 1))
    adds = dict(add_floor=r(B, 3), add_pose0=r(B, 63), add_trans_vel=r(B, 3), add_joints_vel=r(B, 22, 3), add_root_orient_vel=r(B, 3))
    names = ('g_floor', 'g_trans0', 'g_root0', 'g_pose0', 'g_jcam', 'g_trans_vel', 'g_joints_vel', 'g_root_orient_vel')
    shapes = dict(g_floor=(B, 3), g_trans0=(B, 3), g_root0=(B, 3), g_pose0=(B, 63), g_jcam=(B, 22, 3), g_trans_vel=(B, 3), g_joints_vel=(B, 22, 3), g_root_orient_vel=(B, 3))
    outs = []
    for mode in ('compact', 'strided', 'strided+add'):
        o = {k: torch.empty(shapes[k], dtype=torch.float32, device=device) for k in names}
        a = _lib.FitPreArgs()
        a.B = B
        jc = jfull[:, :22].contiguous() if mode == 'compact' else jfull
        a.jcam_stride = 0 if mode == 'compact' else 73 * 3
        for k, v in dict(jcam=jc, **ins, **gin, **o, **(adds if mode.endswith('add') else {})).items():
            setattr(a, k, v.data_ptr())
        lib.call('ha_fit_pre_backward', C.byref(a), st)
        outs.append(o)
    for k in names:
        assert torch.equal(outs[0][k], outs[1][k]), ('fit_pre stride', k)
        ak = {'g_floor': 'add_floor', 'g_pose0': 'add_pose0', 'g_trans_vel': 'add_trans_vel', 'g_joints_vel': 'add_joints_vel', 'g_root_orient_vel': 'add_root_orient_vel'}.get(k)
        want = outs[0][k] + (adds[ak] if ak else 0.0)
        e = (outs[2][k] - want).abs().max().item() / max(1.0, want.abs().max().item())
        worst = max(worst, e)
        assert e < 1e-6, ('fit_pre', k, e)
    return worst
