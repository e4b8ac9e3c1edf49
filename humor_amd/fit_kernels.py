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
               ('prev_tail', 'g_prev_tail'), ('prev_betas', 'g_prev_betas'), ('prev_floor', 'g_prev_floor')]
CONST_INPUTS = ['obs_j2d', 'smpl2op', 'op_mask', 'cam_f', 'cam_c', 'obs_j3d', 'obs_v3d', 'obs_floor', 'overlap']


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
        for (name, _), t in zip(DIFF_INPUTS, tensors):
            if t is None:
                sizes.append(0)
                continue
            t = t.detach()
            if not t.is_contiguous() or t.dtype != torch.float32:
                t = t.contiguous().float()
            keep.append(t)
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
        # one flat buffer: [gradients of every present input | terms NT | loss 1 | per-frame partial sums F*NT]
        ng = sum(sizes)
        flat = torch.empty(ng + NT + 1 + F * NT, dtype=torch.float32, device=dev)
        base, esz, o = flat.data_ptr(), 4, 0
        views = []
        for (_, gname), t, n in zip(DIFF_INPUTS, tensors, sizes):
            if n == 0:
                views.append(None)
                continue
            setattr(a, gname, base + o * esz)
            views.append((o, n, tuple(t.shape)))
            o += n
        a.terms, a.loss, a.partial = base + ng * esz, base + (ng + NT) * esz, base + (ng + NT + 1) * esz
        lib.call('ha_fit_loss', C.byref(a), _lib.stream_ptr(ref))
        ctx.flat, ctx.views, ctx.ng = flat, views, ng
        ctx.set_materialize_grads(False)
        loss, terms = flat[ng + NT], flat[ng:ng + NT]
        ctx.mark_non_differentiable(terms)
        return loss, terms

    @staticmethod
    def backward(ctx, g_loss, _g_terms):
        if g_loss is None:
            return (None, None) + (None,) * len(DIFF_INPUTS)
        scaled = ctx.flat[:ctx.ng] * g_loss           # one launch for all inputs
        out = [None if v is None else scaled[v[0]:v[0] + v[1]].view(v[2]) for v in ctx.views]
        return (None, None) + tuple(out)
