"""The stage-3 objective of MotionOptimizer (humor/fitting/motion_optimizer.py:514-610) as three autograd nodes instead of ten.

Between the library's kernels an objective evaluation used to launch ~36 small ATen kernels (4-5 us each on the GPU, ~10 us of host
time each): the `cat`s in front of the body model, the expanded betas and their summed gradient, slice_backward fills, and above all
one `add` for every tensor that two of the library's autograd Functions read (autograd sums the two gradients with a launch of its
own).  Here the same library calls are grouped so that no tensor with two readers crosses a node boundary:

  Stage3Head   VPoser decode -> frame-0 SMPL (camera frame) -> ha_fit_pre        (latents, root pose, shape, floor, velocities
                                                                                   -> initial roll-out state, key frame, cam2prior)
  (roll-out    HumorModel.roll_out: one node as before, now handing its latent sequence through)
  Stage3Body   ha_rollout_post -> SMPL of the rolled-out poses -> ha_rigid_image (world states -> prior- and camera-frame bodies)
  (loss        fit_kernels.FusedFit, with the init-state GMM term folded in)

Inside a node the gradient that a second reader contributes is passed to the kernel of the first as an ADDEND (ha_smpl_backward_parts,
ha_fit_pre_backward, ha_rigid_image_backward, ha_humor_rollout_backward_ex: `out = adjoint + addend` in the kernel); a tensor that a later
node also reads is handed through the earlier node as an extra output ("thru"), so its two gradients meet inside that node.  The
arithmetic of every kernel is unchanged; only the order in which gradient contributions are summed differs (fp32 rounding).
"""
import ctypes as C

import torch

from . import _lib


def _new(dev, *sh):
    return torch.empty(sh, dtype=torch.float32, device=dev)


def _out(cfg, name, dev, *shape):
    """Buffer for a gradient the node returns: the place the caller reserved for it (cfg['grad_out'], e.g. the rows of the sharded
    closure's gradient arena -- the kernel then writes the variable's gradient where the all-reduce reads it) or a new tensor."""
    t = cfg.get('grad_out', {}).get(name) if cfg.get('grad_out') else None
    if t is not None and tuple(t.shape) == tuple(shape) and t.is_contiguous() and t.dtype == torch.float32 and t.device == dev:
        return t
    return _new(dev, *shape)


def _c(x):
    # (the common case -- a contiguous fp32 tensor -- costs one check: these run ~40 times per evaluation, between L-BFGS's host read and
    # the first launch of the next one)
    if x is None or (x.dtype is torch.float32 and x.is_contiguous()):
        return x
    return x.detach().contiguous().float()


def _p(x):
    return _lib.ptr(x) if x is not None else None


def _sum_opt(a, b):
    """a + b where either may be None (only on paths the fitting configurations do not take: one extra launch)."""
    if a is None:
        return b
    if b is None:
        return a
    return a + b


class Stage3Head(torch.autograd.Function):
    """(latent_pose [B,D], trans [B,3], root_orient [B,3], betas [B,NB], floor [B,3], trans_vel [B,3], joints_vel [B,22,3],
    root_orient_vel [B,3]) -> (pose0 [B,63], past_in [B,339], trans_p, root_p [B,3], joints_p [B,22,3], c2p_R [B,3,3], c2p_t [B,3],
    root_height [B,1], and handed through for the later nodes: floor, trans_vel, joints_vel, root_orient_vel, betas).
    cfg: dict(lib, smpl=BodyModel.parts_config(), vposer=mlp.FusedMLP decoder)."""

    @staticmethod
    def forward(ctx, cfg, latent_pose, trans, root_orient, betas, floor, trans_vel, joints_vel, root_orient_vel):
        lib, sm, vp = cfg['lib'], cfg['smpl'], cfg['vposer']
        ins = [_c(x) for x in (latent_pose, trans, root_orient, betas, floor, trans_vel, joints_vel, root_orient_vel)]
        z, tr, ro, be, fl, tv, jv, rv = ins
        B, dev = z.shape[0], z.device
        st = _lib.stream_ptr(z)
        # VPoser decode (+ 6-D -> R -> axis-angle)
        nws = cfg.setdefault('_vposer_ws', {}).get(B)
        if nws is None:
            n = C.c_int64()
            lib.call('ha_mlp_workspace', vp.ptr, B, C.byref(n))
            nws = cfg['_vposer_ws'][B] = n.value
        vws = _new(dev, nws)
        pose0 = _new(dev, B, vp.out_dim // 2)
        lib.call('ha_mlp_forward', vp.ptr, B, _p(z), 1, _p(pose0), _p(vws), st)
        # frame-0 body in the camera frame: the joints alone (ha_fit_pre reads the first 22; no vertex is evaluated)
        h = sm['handle']
        slot, n_head, tail = sm['slot_all'], 0, None
        jrows = sm['J']
        joints = _new(dev, B, jrows, 3)
        lib.call('ha_smpl_forward_parts', h.ptr, slot, B, sm['n_active'], _p(ro), _p(pose0), _p(be), 1, _p(tr), n_head, _p(joints), _p(tail), st)
        out = dict(past_in=_new(dev, B, 339), trans_p=_new(dev, B, 3), root_p=_new(dev, B, 3), joints_p=_new(dev, B, 22, 3),
                   c2p_R=_new(dev, B, 3, 3), c2p_t=_new(dev, B, 3), root_height=_new(dev, B, 1))
        a = _lib.FitPreArgs()
        a.B = B
        for k, v in dict(floor=fl, trans0=tr, root0=ro, pose0=pose0, jcam=joints, trans_vel=tv, joints_vel=jv, root_orient_vel=rv, **out).items():
            setattr(a, k, v.data_ptr())
        a.jcam_stride = jrows * 3
        lib.call('ha_fit_pre_forward', C.byref(a), st)
        ctx.cfg, ctx.vws, ctx.slot, ctx.n_head, ctx.jrows = cfg, vws, slot, n_head, jrows
        ctx.save_for_backward(z, tr, ro, be, fl, tv, jv, rv, pose0, joints)
        ctx.set_materialize_grads(False)
        return (pose0, out['past_in'], out['trans_p'], out['root_p'], out['joints_p'], out['c2p_R'], out['c2p_t'], out['root_height'],
                floor, trans_vel, joints_vel, root_orient_vel, betas)

    @staticmethod
    def backward(ctx, g_pose0, g_past_in, g_trans_p, g_root_p, g_joints_p, g_c2p_R, g_c2p_t, g_root_height, g_floor_t, g_tv_t, g_jv_t,
                 g_rv_t, g_betas_t):
        cfg = ctx.cfg
        lib, sm, vp = cfg['lib'], cfg['smpl'], cfg['vposer']
        z, tr, ro, be, fl, tv, jv, rv, pose0, joints = ctx.saved_tensors
        B, dev = z.shape[0], z.device
        st = _lib.stream_ptr(z)
        keep = [_c(g) for g in (g_pose0, g_past_in, g_trans_p, g_root_p, g_joints_p, g_c2p_R, g_c2p_t, g_root_height, g_floor_t, g_tv_t,
                                g_jv_t, g_rv_t, g_betas_t)]
        (g_pose0, g_past_in, g_trans_p, g_root_p, g_joints_p, g_c2p_R, g_c2p_t, g_root_height, g_floor_t, g_tv_t, g_jv_t, g_rv_t, g_betas_t) = keep
        # ha_fit_pre adjoint; the gradients of the handed-through floor / velocities and RolloutPost's dL/dpose0 ride along as addends
        o = dict(g_floor=_out(cfg, 'floor', dev, B, 3), g_trans0=_new(dev, B, 3), g_root0=_new(dev, B, 3), g_pose0=_new(dev, B, 63), g_jcam=_new(dev, B, 22, 3),
                 g_trans_vel=_out(cfg, 'trans_vel', dev, B, 3), g_joints_vel=_out(cfg, 'joints_vel', dev, B, 22, 3),
                 g_root_orient_vel=_out(cfg, 'root_orient_vel', dev, B, 3))
        a = _lib.FitPreArgs()
        a.B = B
        fields = dict(floor=fl, trans0=tr, root0=ro, pose0=pose0, jcam=joints, trans_vel=tv, joints_vel=jv, root_orient_vel=rv,
                      g_past_in=g_past_in, g_trans_p=g_trans_p, g_root_p=g_root_p, g_joints_p=g_joints_p, g_c2p_R=g_c2p_R, g_c2p_t=g_c2p_t,
                      g_root_height=g_root_height, add_floor=g_floor_t, add_pose0=g_pose0, add_trans_vel=g_tv_t, add_joints_vel=g_jv_t,
                      add_root_orient_vel=g_rv_t, **o)
        for k, v in fields.items():
            if v is not None:
                setattr(a, k, v.data_ptr())
        a.jcam_stride = ctx.jrows * 3
        lib.call('ha_fit_pre_backward', C.byref(a), st)
        # frame-0 SMPL adjoint: joint gradients = dL/djcam (22 rows), ha_fit_pre's direct gradients of trans / root / pose as addends
        g_root, g_body = _out(cfg, 'root_orient', dev, B, 3), _new(dev, B, 63)
        g_betas, g_transl = _out(cfg, 'betas', dev, B, be.shape[1]), _out(cfg, 'trans', dev, B, 3)
        lib.call('ha_smpl_backward_parts', sm['handle'].ptr, ctx.slot, B, sm['n_active'], _p(ro), _p(pose0), _p(be), 1, ctx.n_head,
                 _p(o['g_jcam']), 22, 22, None, _p(o['g_root0']), _p(o['g_pose0']), _p(g_betas_t), _p(o['g_trans0']),
                 _p(g_root), _p(g_body), _p(g_betas), _p(g_transl), st)
        g_z = _out(cfg, 'latent_pose', dev, B, vp.in_dim)
        lib.call('ha_mlp_backward', vp.ptr, B, _p(g_body), 1, _p(ctx.vws), _p(g_z), st)
        return None, g_z, g_transl, g_root, g_betas, o['g_floor'], o['g_trans_vel'], o['g_joints_vel'], o['g_root_orient_vel']


class Stage3Body(torch.autograd.Function):
    """(world [B,S,348], trans0 [B,3], root0 [B,3], pose0 [B,63], joints0 [B,22,3], c2p_R [B,3,3] | None, c2p_t [B,3] | None, betas [B,NB])
    -> (pri_jtr [B,T,Jx,3], pri_verts [B,T,nv,3], cam_jtr, cam_verts (None without cam2prior), trans, root_orient, pose_body [B,T,.],
    ro_joints [B,T,22,3], contacts_conf, contacts [B,T,22], cam_trans, cam_root_orient [B,T,3] | None, betas handed through).
    cfg: dict(lib, smpl=BodyModel.parts_config())."""

    @staticmethod
    def forward(ctx, cfg, world, trans0, root0, pose0, joints0, c2p_R, c2p_t, betas):
        lib, sm = cfg['lib'], cfg['smpl']
        world, trans0, root0, pose0, joints0, c2p_R, c2p_t, be = (_c(x) for x in (world, trans0, root0, pose0, joints0, c2p_R, c2p_t, betas))
        B, S = world.shape[0], world.shape[1]
        T, dev = S + 1, world.device
        N = B * T
        st = _lib.stream_ptr(world)
        cam = c2p_R is not None
        post = dict(trans=_new(dev, B, T, 3), root_orient=_new(dev, B, T, 3), pose_body=_new(dev, B, T, 63), joints=_new(dev, B, T, 22, 3),
                    contacts_conf=_new(dev, B, T, 22), contacts=_new(dev, B, T, 22))
        if cam:
            post.update(cam_trans=_new(dev, B, T, 3), cam_root_orient=_new(dev, B, T, 3))
        a = _lib.RolloutPostArgs()
        a.B, a.S = B, S
        for k, v in dict(world=world, trans0=trans0, root0=root0, pose0=pose0, joints0=joints0, c2p_R=c2p_R, c2p_t=c2p_t, **post).items():
            if v is not None:
                setattr(a, k, v.data_ptr())
        lib.call('ha_rollout_post_forward', C.byref(a), st)
        # body of the rolled-out poses in the prior frame (one shape row per sequence)
        h, n_sel = sm['handle'], sm['n_sel']
        jx, nv = sm['J'] + n_sel, sm['n_all'] - n_sel
        jtr, verts = _new(dev, B, T, jx, 3), _new(dev, B, T, nv, 3)
        lib.call('ha_smpl_forward_parts', h.ptr, sm['slot_all'], N, sm['n_active'], _p(post['root_orient']), _p(post['pose_body']), _p(be), T,
                 _p(post['trans']), n_sel, _p(jtr), _p(verts) if nv > 0 else None, st)
        cam_jtr = cam_verts = None
        if cam:
            cam_jtr, cam_verts = torch.empty_like(jtr), torch.empty_like(verts)
            r = _lib.RigidImageArgs()
            r.N, r.J, r.V = N, jx, nv
            for k, v in dict(joints=jtr, verts=verts, root=post['root_orient'], trans=post['trans'], root2=post['cam_root_orient'],
                             trans2=post['cam_trans'], joints2=cam_jtr, verts2=cam_verts).items():
                if v.numel():
                    setattr(r, k, v.data_ptr())
            lib.call('ha_rigid_image_forward', C.byref(r), st)
        ctx.cfg, ctx.cam, ctx.dims = cfg, cam, (B, S, jx, nv)
        ctx.save_for_backward(world, trans0, root0, c2p_R, c2p_t, be, post['trans'], post['root_orient'], post['pose_body'], post['contacts_conf'],
                              post.get('cam_trans'), post.get('cam_root_orient'), jtr, verts)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(post['contacts'])
        return (jtr, verts, cam_jtr, cam_verts, post['trans'], post['root_orient'], post['pose_body'], post['joints'], post['contacts_conf'],
                post['contacts'], post.get('cam_trans'), post.get('cam_root_orient'), betas)

    @staticmethod
    def backward(ctx, g_jtr, g_verts, g_cam_jtr, g_cam_verts, g_trans, g_root, g_pose, g_ro_joints, g_conf, _g_lab, g_ct, g_cr, g_betas_t):
        cfg = ctx.cfg
        lib, sm = cfg['lib'], cfg['smpl']
        world, trans0, root0, c2p_R, c2p_t, be, trans, root_orient, pose_body, conf, cam_trans, cam_root, jtr, verts = ctx.saved_tensors
        B, S, jx, nv = ctx.dims
        T, dev = S + 1, world.device
        N = B * T
        st = _lib.stream_ptr(world)
        g_jtr, g_verts, g_cam_jtr, g_cam_verts, g_trans, g_root, g_pose, g_ro_joints, g_conf, g_ct, g_cr, g_betas_t = (
            _c(g) for g in (g_jtr, g_verts, g_cam_jtr, g_cam_verts, g_trans, g_root, g_pose, g_ro_joints, g_conf, g_ct, g_cr, g_betas_t))
        add_root, add_transl, g_t2, g_r2 = g_root, g_trans, g_ct, g_cr
        if ctx.cam and (g_cam_jtr is not None or g_cam_verts is not None):
            # adjoint of the rigid image; the loss's gradients of the prior-frame joints / vertices ride along as addends
            o = dict(g_joints=torch.empty_like(jtr), g_verts=torch.empty_like(verts), g_root=_new(dev, N, 3), g_trans=_new(dev, N, 3),
                     g_root2=_new(dev, N, 3), g_trans2=_new(dev, N, 3))
            r = _lib.RigidImageArgs()
            r.N, r.J, r.V = N, jx, nv
            for k, v in dict(joints=jtr, verts=verts, root=root_orient, trans=trans, root2=cam_root, trans2=cam_trans, g_joints2=g_cam_jtr,
                             g_verts2=g_cam_verts, g_joints_add=g_jtr, g_verts_add=g_verts, **o).items():
                if v is not None and v.numel():
                    setattr(r, k, v.data_ptr())
            lib.call('ha_rigid_image_backward', C.byref(r), st)
            g_jtr, g_verts = o['g_joints'], (o['g_verts'] if nv > 0 else None)
            add_root, add_transl = _sum_opt(g_root, o['g_root']), _sum_opt(g_trans, o['g_trans'])
            g_t2, g_r2 = _sum_opt(g_ct, o['g_trans2']), _sum_opt(g_cr, o['g_root2'])
        # SMPL adjoint over all frames; the rigid image's gradients of the root trajectory as addends
        NB = be.shape[1]
        g_ro, g_pb, g_bf, g_tl = _new(dev, B, T, 3), _new(dev, B, T, 63), _new(dev, N, NB), _new(dev, B, T, 3)
        lib.call('ha_smpl_backward_parts', sm['handle'].ptr, sm['slot_all'], N, sm['n_active'], _p(root_orient), _p(pose_body), _p(be), T,
                 sm['n_sel'], _p(g_jtr), 0, 0, _p(g_verts), _p(add_root), _p(g_pose), None, _p(add_transl),
                 _p(g_ro), _p(g_pb), _p(g_bf), _p(g_tl), st)
        g_betas = _new(dev, B, NB)
        lib.call('ha_seq_sum_add', B, T, NB, _p(g_bf), _p(g_betas_t), None, _p(g_betas), st)
        # adjoint of the post-processing
        gw, gt0, gr0, gp0, gj0 = _new(dev, B, S, 348), _new(dev, B, 3), _new(dev, B, 3), _new(dev, B, 63), _new(dev, B, 22, 3)
        gR, gt = (_new(dev, B, 3, 3), _new(dev, B, 3)) if ctx.cam else (None, None)
        partial = _new(dev, N, 15)
        a = _lib.RolloutPostArgs()
        a.B, a.S = B, S
        fields = dict(world=world, trans0=trans0, root0=root0, c2p_R=c2p_R, c2p_t=c2p_t, root_orient=root_orient, contacts_conf=conf,
                      g_trans=g_tl, g_root_orient=g_ro, g_pose_body=g_pb, g_joints=g_ro_joints, g_contacts_conf=g_conf,
                      g_cam_trans=g_t2, g_cam_root_orient=g_r2, g_world=gw, g_trans0=gt0, g_root0=gr0, g_pose0=gp0, g_joints0=gj0,
                      g_c2p_R=gR, g_c2p_t=gt, partial=partial)
        for k, v in fields.items():
            if v is not None:
                setattr(a, k, v.data_ptr())
        lib.call('ha_rollout_post_backward', C.byref(a), st)
        return None, gw, gt0, gr0, gp0, gj0, gR, gt, g_betas
