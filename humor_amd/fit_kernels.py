"""Autograd binding of ``ha_fit_loss`` (include/humor_amd.h): every data / regularisation term of the fitting objective and
its gradient in one kernel launch (+ one single-block reduction), instead of ~250 element-wise / reduction launches of the
term-by-term PyTorch evaluation (forward + autograd backward).  The kernel produces d(loss)/d(input) during the forward
pass (every term is a plain sum), so the backward pass is one scaling of a flat buffer by the incoming gradient.

Term order = the HA_FIT_* indices of the header; ``TERM_NAMES`` maps them to the keys of the reference's ``stats_dict``
(humor/fitting/fitting_loss.py:94-309)."""
import ctypes as C

import torch

from . import _lib

NT = 17
(J2D, J3D, V3D, J3D_RO, POSE_PRIOR, SHAPE_PRIOR, SMOOTH, MOTION_PRIOR, JOINT_CONSIST, BONE_LEN, CONTACT_VEL, CONTACT_H, FLOOR_REG,
 OV_VPOS, OV_VVEL, OV_BETAS, OV_FLOOR) = range(NT)
TERM_NAMES = ['joints2d', 'joints3d', 'verts3d', 'joints3d_rollout', 'pose_prior', 'shape_prior', 'joints3d_smooth', 'motion_prior',
              'joint_consistency', 'bone_length', 'contact_vel', 'contact_height', 'floor_reg', 'rgb_overlap_consist_verts3d_pos',
              'rgb_overlap_consist_verts3d_vel', 'rgb_overlap_consist_betas', 'rgb_overlap_consist_floor']

# differentiable inputs, in the order of FusedFit.apply's tensor arguments; (field, gradient field)
DIFF_INPUTS = [('cam_jtr', 'g_cam_jtr'), ('cam_verts', 'g_cam_verts'), ('pri_joints', 'g_pri_joints'), ('ro_joints', 'g_ro_joints'),
               ('contacts_conf', 'g_contacts_conf'), ('latent_pose', 'g_latent_pose'), ('betas', 'g_betas'),
               ('latent_motion', 'g_latent_motion'), ('prior_mu', 'g_prior_mu'), ('prior_var', 'g_prior_var'), ('floor', 'g_floor'),
               ('prev_tail', 'g_prev_tail'), ('prev_betas', 'g_prev_betas'), ('prev_floor', 'g_prev_floor'),
               # inputs of the folded init-state prior only (spec['gmm']): their gradients come from ha_gmm_nll through ha_fit_loss
               ('joints_vel', None), ('trans_vel', None), ('root_orient_vel', None)]
GMM_INPUTS = ('joints_vel', 'trans_vel', 'root_orient_vel')
CONST_INPUTS = ['obs_j2d', 'smpl2op', 'op_mask', 'cam_f', 'cam_c', 'obs_j3d', 'obs_v3d', 'obs_floor', 'overlap']


_UNIT = {}


def unit_seed(ref):
    """The constant 1.0 the closure seeds loss.backward() with, one cached 0-dim tensor per device: backward(gradient=unit_seed(loss))
    saves autograd's ones_like launch, and FusedFit.backward recognises the tensor (by address) and skips its multiply-by-one launch."""
    key = (ref.device, ref.dtype)
    t = _UNIT.get(key)
    if t is None:
        t = _UNIT[key] = torch.ones((), dtype=ref.dtype, device=ref.device)
    return t


class FusedFit(torch.autograd.Function):
    """(differentiable tensors in DIFF_INPUTS order, None = absent) -> (loss [], terms [NT]).  `spec` carries the constant
    tensors (observations, tables), the NT weights and nsteps."""

    @staticmethod
    def forward(ctx, lib, spec, *tensors):
        assert len(tensors) == len(DIFF_INPUTS)
        ref = next(t for t in tensors if t is not None)
        dev = ref.device
        a = _lib.FitArgs()
        a.B, a.T = spec['B'], spec['T']
        keep = []
        sizes = []
        named = {}
        for (name, _), t in zip(DIFF_INPUTS, tensors):
            if t is None:
                sizes.append(0)
                continue
            t = t.detach()
            if not t.is_contiguous() or t.dtype != torch.float32:
                t = t.contiguous().float()
            keep.append(t)
            named[name] = t
            if name not in GMM_INPUTS:
                setattr(a, name, t.data_ptr())
            sizes.append(t.numel())
        for name in CONST_INPUTS:
            t = spec.get(name)
            if t is not None:
                assert t.is_contiguous() and t.device == dev, name
                setattr(a, name, t.data_ptr())
        a.pri_nj = spec.get('pri_nj', 22)
        a.nj, a.nv, a.dlp, a.nb, a.S, a.dz = spec.get('nj', 0), spec.get('nv', 0), spec.get('dlp', 0), spec.get('nb', 0), spec.get('S', 0), spec.get('dz', 0)
        a.sigma, a.nsteps = float(spec.get('sigma', 0.0)), float(spec['nsteps'])
        for k in range(NT):
            a.w[k] = float(spec['w'][k])
        F = a.B * a.T
        # one flat buffer: [gradients of every present input | terms NT | loss 1 | init-state prior total 1 | per-frame partial sums F*NT]
        ng = sum(sizes)
        flat = torch.empty(ng + NT + 2 + F * NT, dtype=torch.float32, device=dev)
        base, esz, o = flat.data_ptr(), 4, 0
        views, goff = [], {}
        for (name, gname), t, n in zip(DIFF_INPUTS, tensors, sizes):
            if n == 0:
                views.append(None)
                continue
            if gname is not None:
                setattr(a, gname, base + o * esz)
            goff[name] = base + o * esz
            views.append((o, n, tuple(t.shape)))
            o += n
        a.terms, a.loss, a.partial = base + ng * esz, base + (ng + NT) * esz, base + (ng + NT + 2) * esz
        gmm = spec.get('gmm')
        if gmm is not None:
            # init-state prior (FittingLoss.init_motion_prior_loss): ha_gmm_nll on [frame 0 of the prior-frame joints | joints_vel |
            # trans_vel | root_orient_vel], its value and gradient folded into the objective by ha_fit_loss's reduction kernel
            pj = named['pri_joints']
            segs = [(pj, 66, a.T * a.pri_nj * 3)] + [(named[k], named[k][0].numel(), named[k][0].numel()) for k in GMM_INPUTS]
            g = gmm['gmm']
            K, D = g.means.shape
            lp, gpart, nll, g_x = (torch.empty(sh, dtype=torch.float32, device=dev) for sh in ((a.B, K), (a.B, K, D), (a.B,), (a.B, D)))
            ga = _lib.GmmArgs()
            ga.B, ga.K, ga.D, ga.nseg = a.B, K, D, len(segs)
            for i, (sgt, w_, st_) in enumerate(segs):
                ga.seg[i], ga.seg_width[i], ga.seg_stride[i] = sgt.data_ptr(), w_, st_
            tb = g.device_tables(dev)
            for k, v in dict(means=tb['means'], Linv=tb['Linv'], LinvT=tb['LinvT'], cst=tb['const'], lp=lp, gpart=gpart, nll=nll, g_x=g_x).items():
                setattr(ga, k, v.data_ptr())
            lib.call('ha_gmm_nll', C.byref(ga), _lib.stream_ptr(ref))
            keep += [lp, gpart, nll, g_x]
            a.gmm_nll, a.gmm_gx, a.gmm_w, a.gmm_D, a.gmm_nseg = nll.data_ptr(), g_x.data_ptr(), float(gmm['w']), D, len(segs)
            a.gmm_g[0], a.gmm_seg_width[0], a.gmm_g_stride[0], a.gmm_g_acc[0] = goff['pri_joints'], 66, a.T * a.pri_nj * 3, 1
            for i, k in enumerate(GMM_INPUTS):
                a.gmm_g[i + 1], a.gmm_seg_width[i + 1], a.gmm_g_stride[i + 1], a.gmm_g_acc[i + 1] = goff[k], segs[i + 1][1], segs[i + 1][1], 0
            a.gmm_total = base + (ng + NT + 1) * esz
        lib.call('ha_fit_loss', C.byref(a), _lib.stream_ptr(ref))
        ctx.flat, ctx.views, ctx.ng = flat, views, ng
        ctx.set_materialize_grads(False)
        loss, terms, gmm_total = flat[ng + NT], flat[ng:ng + NT], flat[ng + NT + 1]
        ctx.mark_non_differentiable(terms, gmm_total)
        return loss, terms, gmm_total

    @staticmethod
    def backward(ctx, g_loss, _g_terms, _g_gmm=None):
        if g_loss is None:
            return (None, None) + (None,) * len(DIFF_INPUTS)
        unit = _UNIT.get((g_loss.device, g_loss.dtype))
        if unit is not None and g_loss.dim() == 0 and g_loss.data_ptr() == unit.data_ptr():
            # (the cached seed is shared by every closure of the process: an in-place write to it would mis-scale all gradients silently;
            # its version counter -- host side, no device read -- says whether anybody has written to it since it was made)
            assert unit._version == 0, 'fit_kernels.unit_seed(): the cached 1.0 was modified in place'
            scaled = ctx.flat[:ctx.ng]                # seeded with the cached 1.0 (MotionOptimizer._finish_closure): nothing to scale
        else:
            scaled = ctx.flat[:ctx.ng] * g_loss       # one launch for all inputs
        out = [None if v is None else scaled[v[0]:v[0] + v[1]].view(v[2]) for v in ctx.views]
        return (None, None) + tuple(out)


class RolloutPost(torch.autograd.Function):
    """(world [B,S,348], trans0 [B,3], root0 [B,3], pose0 [B,63], joints0 [B,22,3], c2p_R [B,3,3] | None, c2p_t [B,3] | None) ->
    (trans, root_orient, pose_body [B,T,.], joints [B,T,22,3], contacts_conf, contacts [B,T,22], cam_trans, cam_root_orient [B,T,3] | None):
    ha_rollout_post_forward / _backward (include/humor_amd.h), one launch per direction (+ a per-sequence reduction)."""

    @staticmethod
    def forward(ctx, lib, world, trans0, root0, pose0, joints0, c2p_R, c2p_t):
        c = lambda x: None if x is None else x.detach().contiguous().float()
        world, trans0, root0, pose0, joints0, c2p_R, c2p_t = (c(x) for x in (world, trans0, root0, pose0, joints0, c2p_R, c2p_t))
        B, S = world.shape[0], world.shape[1]
        T, dev = S + 1, world.device
        new = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        out = dict(trans=new(B, T, 3), root_orient=new(B, T, 3), pose_body=new(B, T, 63), joints=new(B, T, 22, 3),
                   contacts_conf=new(B, T, 22), contacts=new(B, T, 22))
        cam = c2p_R is not None
        if cam:
            out.update(cam_trans=new(B, T, 3), cam_root_orient=new(B, T, 3))
        a = _lib.RolloutPostArgs()
        a.B, a.S = B, S
        for k, v in dict(world=world, trans0=trans0, root0=root0, pose0=pose0, joints0=joints0, c2p_R=c2p_R, c2p_t=c2p_t, **out).items():
            if v is not None:
                setattr(a, k, v.data_ptr())
        lib.call('ha_rollout_post_forward', C.byref(a), _lib.stream_ptr(world))
        ctx.lib, ctx.cam, ctx.dims = lib, cam, (B, S)
        ctx.save_for_backward(world, trans0, root0, c2p_R, c2p_t, out['root_orient'], out['contacts_conf'])
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(out['contacts'])
        return (out['trans'], out['root_orient'], out['pose_body'], out['joints'], out['contacts_conf'], out['contacts'],
                out.get('cam_trans'), out.get('cam_root_orient'))

    @staticmethod
    def backward(ctx, g_trans, g_root, g_pose, g_joints, g_conf, _g_lab, g_ct, g_cr):
        world, trans0, root0, c2p_R, c2p_t, root_orient, conf = ctx.saved_tensors
        B, S = ctx.dims
        T, dev = S + 1, world.device
        new = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        gw, gt0, gr0, gp0, gj0 = new(B, S, 348), new(B, 3), new(B, 3), new(B, 63), new(B, 22, 3)
        gR, gt = (new(B, 3, 3), new(B, 3)) if ctx.cam else (None, None)
        partial = new(B * T, 15)
        a = _lib.RolloutPostArgs()
        a.B, a.S = B, S
        keep = []
        c = lambda x: None if x is None else x.contiguous().float()
        fields = dict(world=world, trans0=trans0, root0=root0, c2p_R=c2p_R, c2p_t=c2p_t, root_orient=root_orient, contacts_conf=conf,
                      g_trans=c(g_trans), g_root_orient=c(g_root), g_pose_body=c(g_pose), g_joints=c(g_joints), g_contacts_conf=c(g_conf),
                      g_cam_trans=c(g_ct), g_cam_root_orient=c(g_cr), g_world=gw, g_trans0=gt0, g_root0=gr0, g_pose0=gp0, g_joints0=gj0,
                      g_c2p_R=gR, g_c2p_t=gt, partial=partial)
        for k, v in fields.items():
            if v is not None:
                keep.append(v)
                setattr(a, k, v.data_ptr())
        ctx.lib.call('ha_rollout_post_backward', C.byref(a), _lib.stream_ptr(world))
        return None, gw, gt0, gr0, gp0, gj0, gR, gt


class RigidImage(torch.autograd.Function):
    """(joints [N,J,3], verts [N,V,3], root [N,3], trans [N,3], root2 [N,3], trans2 [N,3]) -> (joints2, verts2): the SMPL output of
    the same pose and shape under the root pose (root2, trans2), as the rigid image of the evaluation made with (root, trans):
    ha_rigid_image_forward / _backward (include/humor_amd.h), one launch per direction."""

    @staticmethod
    def forward(ctx, lib, joints, verts, root, trans, root2, trans2):
        joints, verts, root, trans, root2, trans2 = (x.detach().contiguous().float() for x in (joints, verts, root, trans, root2, trans2))
        N, J, V = joints.shape[0], joints.shape[1], verts.shape[1]
        joints2, verts2 = torch.empty_like(joints), torch.empty_like(verts)
        a = _lib.RigidImageArgs()
        a.N, a.J, a.V = N, J, V
        for k, v in dict(joints=joints, verts=verts, root=root, trans=trans, root2=root2, trans2=trans2, joints2=joints2, verts2=verts2).items():
            if v.numel():
                setattr(a, k, v.data_ptr())
        lib.call('ha_rigid_image_forward', C.byref(a), _lib.stream_ptr(joints))
        ctx.lib = lib
        ctx.save_for_backward(joints, verts, root, trans, root2, trans2)
        ctx.set_materialize_grads(False)
        return joints2, verts2

    @staticmethod
    def backward(ctx, g_joints2, g_verts2):
        joints, verts, root, trans, root2, trans2 = ctx.saved_tensors
        N, J, V = joints.shape[0], joints.shape[1], verts.shape[1]
        c = lambda x: None if x is None else x.contiguous().float()
        g_joints2, g_verts2 = c(g_joints2), c(g_verts2)
        out = dict(g_joints=torch.empty_like(joints), g_verts=torch.empty_like(verts), g_root=torch.empty_like(root),
                   g_trans=torch.empty_like(trans), g_root2=torch.empty_like(root2), g_trans2=torch.empty_like(trans2))
        a = _lib.RigidImageArgs()
        a.N, a.J, a.V = N, J, V
        for k, v in dict(joints=joints, verts=verts, root=root, trans=trans, root2=root2, trans2=trans2, g_joints2=g_joints2,
                         g_verts2=g_verts2, **out).items():
            if v is not None and v.numel():
                setattr(a, k, v.data_ptr())
        ctx.lib.call('ha_rigid_image_backward', C.byref(a), _lib.stream_ptr(joints))
        return (None, out['g_joints'], out['g_verts'], out['g_root'], out['g_trans'], out['g_root2'], out['g_trans2'])


class GmmNll(torch.autograd.Function):
    """(segments [B, ...] whose per-sequence sizes add up to the mixture's dimension) -> -sum_b log p(x_b) under the Gaussian mixture
    `gmm` (fitting_loss._GMM): ha_gmm_nll (include/humor_amd.h), value and gradient in two launches + one row sum.  A segment that is a
    view with contiguous rows (frame 0 of a [B,T,J,3] tensor) is read in place."""

    @staticmethod
    def forward(ctx, lib, gmm, *segs):
        B, dev = segs[0].shape[0], segs[0].device
        rows = []
        for s in segs:
            s = s.detach()
            w = s[0].numel()
            if s.dtype == torch.float32 and s[0].is_contiguous() and (B == 1 or s.stride(0) >= w):
                rows.append((s, w, s.stride(0) if B > 1 else w))         # row b starts at data_ptr + b * stride(0)
            else:
                s = s.reshape(B, -1).contiguous().float()
                rows.append((s, w, w))
        K, D = gmm.means.shape
        new = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        lp, gpart, nll, g_x = new(B, K), new(B, K, D), new(B), new(B, D)
        a = _lib.GmmArgs()
        a.B, a.K, a.D, a.nseg = B, K, D, len(rows)
        for i, (s, w, st) in enumerate(rows):
            a.seg[i], a.seg_width[i], a.seg_stride[i] = s.data_ptr(), w, st
        t = gmm.device_tables(dev)
        for k, v in dict(means=t['means'], Linv=t['Linv'], LinvT=t['LinvT'], cst=t['const'], lp=lp, gpart=gpart, nll=nll, g_x=g_x).items():
            setattr(a, k, v.data_ptr())
        lib.call('ha_gmm_nll', C.byref(a), _lib.stream_ptr(rows[0][0]))
        ctx.shapes = [tuple(s.shape) for s in segs]
        ctx.save_for_backward(g_x)
        return nll.sum()

    @staticmethod
    def backward(ctx, g):
        g_x, = ctx.saved_tensors
        gx = g_x * g
        out, o = [], 0
        for sh in ctx.shapes:
            w = 1
            for d in sh[1:]:
                w *= d
            out.append(gx[:, o:o + w].reshape(sh))
            o += w
        return (None, None) + tuple(out)


class FitPre(torch.autograd.Function):
    """(floor [B,3], trans0 [B,3], root0 [B,3], pose0 [B,63], jcam [B,22,3], trans_vel [B,3], joints_vel [B,22,3], root_orient_vel [B,3])
    -> (past_in [B,339], trans_p [B,3], root_p [B,3], joints_p [B,22,3], c2p_R [B,3,3], c2p_t [B,3], root_height [B,1]):
    ha_fit_pre_forward / _backward (include/humor_amd.h)."""
    IN = ['floor', 'trans0', 'root0', 'pose0', 'jcam', 'trans_vel', 'joints_vel', 'root_orient_vel']

    @staticmethod
    def forward(ctx, lib, *ins):
        ins = [x.detach().contiguous().float() for x in ins]
        B, dev = ins[0].shape[0], ins[0].device
        new = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        out = dict(past_in=new(B, 339), trans_p=new(B, 3), root_p=new(B, 3), joints_p=new(B, 22, 3), c2p_R=new(B, 3, 3), c2p_t=new(B, 3),
                   root_height=new(B, 1))
        a = _lib.FitPreArgs()
        a.B = B
        for k, v in zip(FitPre.IN, ins):
            setattr(a, k, v.data_ptr())
        for k, v in out.items():
            setattr(a, k, v.data_ptr())
        lib.call('ha_fit_pre_forward', C.byref(a), _lib.stream_ptr(ins[0]))
        ctx.lib = lib
        ctx.save_for_backward(*ins)
        ctx.set_materialize_grads(False)
        return tuple(out.values())

    @staticmethod
    def backward(ctx, *gs):
        ins = ctx.saved_tensors
        a = _lib.FitPreArgs()
        a.B = ins[0].shape[0]
        keep = []
        for k, v in zip(FitPre.IN, ins):
            setattr(a, k, v.data_ptr())
        for k, g in zip(('g_past_in', 'g_trans_p', 'g_root_p', 'g_joints_p', 'g_c2p_R', 'g_c2p_t', 'g_root_height'), gs):
            if g is not None:
                g = g.contiguous().float()
                keep.append(g)
                setattr(a, k, g.data_ptr())
        outs = [torch.empty_like(x) for x in ins]
        for k, v in zip(FitPre.IN, outs):
            setattr(a, 'g_' + k, v.data_ptr())
        ctx.lib.call('ha_fit_pre_backward', C.byref(a), _lib.stream_ptr(ins[0]))
        return (None,) + tuple(outs)
