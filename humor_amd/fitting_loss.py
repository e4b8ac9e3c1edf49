"""Loss assembly of the fitting closures: same class name, constructor, stage methods (``root_fit`` / ``smpl_fit`` /
``motion_fit``), weight keys and per-term values as the reference's ``FittingLoss``
(humor/fitting/fitting_loss.py:20-517), written for the GPU:

  * masked terms multiply by the visibility mask instead of boolean-mask indexing (no dynamic shapes, no host
    sync; invisible = +-inf observations contribute exactly 0, as in the reference, G6);
  * ``stats_dict`` holds device tensors -- nothing here calls ``.item()``;
  * the overlap-consistency terms take an optional ``halo`` so a rank can evaluate the pair (b-1, b) when b-1 lives
    on the neighbouring rank (humor_amd/distributed.py).

Host-side PyTorch: ~60 small element-wise/reduction launches per closure; the HIP kernels are below it (SMPL, roll-out,
rotation conversions).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fit_kernels as FK
from .tables import OP_NUM_JOINTS, SMPL_PARENTS

CONTACT_HEIGHT_THRESH = 0.08   # fitting_loss.py:18


def gmof(res, sigma):
    """Geman-McClure robust function (fitting_utils.py:250-258)."""
    x2 = res ** 2
    s2 = sigma ** 2
    return (s2 * x2) / (s2 + x2)


def perspective_projection(points, focal_length, camera_center):
    """Pinhole projection with identity extrinsics (fitting_utils.py:647-676 with R = I, t = 0).
    points [N,K,3], focal_length [N,2], camera_center [N,2] -> [N,K,2]."""
    proj = points[:, :, :2] / points[:, :, 2:3]
    return proj * focal_length.unsqueeze(1) + camera_center.unsqueeze(1)


def robust_std(res):
    """Robust standard deviation from the median absolute deviation, per batch row (fitting_utils.py:212-227).  res [B,N] -> [B,1]."""
    B = res.size(0)
    med = torch.median(res, dim=-1)[0].reshape((B, 1))
    mad = torch.median(torch.abs(res - med), dim=-1)[0].reshape((B, 1))
    return mad / 0.67449


def bisquare_robust_weights(res, tune_const=4.6851):
    """Tukey bisquare weights of non-negative residuals (fitting_utils.py:229-249)."""
    norm_res = res / (robust_std(res) * tune_const)
    w = (1.0 - norm_res ** 2) ** 2
    return torch.where(norm_res >= 1.0, torch.zeros_like(w), w)


def apply_robust_weighting(res, robust_loss_type='bisquare', robust_tuning_const=4.6851):
    """Robustly weighted squared residuals; no gradient through the weights (fitting_utils.py:192-210)."""
    det = res.detach()
    w = torch.ones_like(det) if robust_loss_type == 'none' else bisquare_robust_weights(det, tune_const=robust_tuning_const)
    return w * (res ** 2), w


def _masked_sq(obs, pred):
    """0.5 * sum over visible entries of (obs - pred)^2; entries whose observation is +-inf are invisible (a NaN
    observation poisons the loss, as in the reference's get_visible_mask = ~isinf, fitting_loss.py:311-315)."""
    vis = torch.logical_not(torch.isinf(obs))
    diff = torch.where(vis, obs - pred, torch.zeros_like(pred))
    return 0.5 * torch.sum(diff * diff)


class _GMM:
    """Gaussian-mixture log-density with the Cholesky factors inverted once at construction.
    Same value as MixtureSameFamily(Categorical(w), MultivariateNormal(mu, cov)).log_prob (run_fitting.py:248-261,
    fitting_loss.py:83-87, 416-429), but the per-call work is one batched mat-vec: torch.distributions' batched
    triangular solve issues host-side pointer-array copies that cannot be captured into a hipGraph."""

    def __init__(self, weights, means, covs):
        L = torch.linalg.cholesky(covs)
        eye = torch.eye(covs.shape[-1], dtype=covs.dtype, device=covs.device).expand_as(covs)
        self.Linv = torch.linalg.solve_triangular(L, eye, upper=False)                   # [K,D,D]
        self.means = means
        self.const = (torch.log(weights / weights.sum()) - torch.log(torch.diagonal(L, dim1=-2, dim2=-1)).sum(-1)
                      - 0.5 * means.shape[-1] * math.log(2 * math.pi))                    # [K]

    def device_tables(self, device):
        """contiguous fp32 tables for ha_gmm_nll on `device` (the factors and their transposes), built once"""
        key = str(device)
        t = getattr(self, '_tables', None)
        if t is None or t[0] != key:
            f = lambda v: v.detach().to(device=device, dtype=torch.float32).contiguous()
            t = self._tables = (key, dict(means=f(self.means), Linv=f(self.Linv), LinvT=f(self.Linv.transpose(1, 2)), const=f(self.const)))
        return t[1]

    def log_prob(self, x):
        diff = x.unsqueeze(1) - self.means.unsqueeze(0)                                  # [B,K,D]
        y = torch.einsum('kij,bkj->bki', self.Linv, diff)
        return torch.logsumexp(self.const.unsqueeze(0) - 0.5 * (y * y).sum(-1), dim=1)


class FittingLoss(nn.Module):
    '''
    Functions to compute all needed losses for fitting.
    '''

    def __init__(self, loss_weights, init_motion_prior=None, smpl2op_map=None, ignore_op_joints=None, cam_f=None,
                 cam_cent=None, robust_loss='none', robust_tuning_const=4.6851, joints2d_sigma=100, use_chamfer=False,
                 fused=True, _lib_override=None):
        super(FittingLoss, self).__init__()
        # fused=True: every term except the init-state GMM is evaluated (value + gradient) by ONE kernel launch
        # (humor_amd/csrc/fitloss.hip, fit_kernels.FusedFit) whenever the predictions live on the GPU; the term-by-term
        # PyTorch evaluation below remains for the cases the kernel does not cover (cross-batch `prev_batch_overlap_res`).
        self.fused = fused
        self.fold_init_prior = True        # stage 3: the init-state GMM term inside the fused loss launches (see _fused_fit)
        self._lib = _lib_override
        self.all_stage_loss_weights = loss_weights
        self.cur_stage_idx = 0
        self.loss_weights = self.all_stage_loss_weights[self.cur_stage_idx]
        self.smpl2op_map = None if smpl2op_map is None else torch.as_tensor(list(smpl2op_map), dtype=torch.long)
        # ha_fit_loss inverts the map in LDS (one OpenPose joint per SMPL joint, indices below 128): any other map -- legal for the
        # reference's gather formulation -- is evaluated term by term instead
        self._smpl2op_fusable = self.smpl2op_map is None or (
            self.smpl2op_map.unique().numel() == self.smpl2op_map.numel() and int(self.smpl2op_map.min()) >= 0 and int(self.smpl2op_map.max()) < 128)
        self.ignore_op_joints = ignore_op_joints
        self.cam_f, self.cam_cent = cam_f, cam_cent
        self.joints2d_sigma = joints2d_sigma
        self.can_reproj = self.smpl2op_map is not None and cam_f is not None and cam_cent is not None
        if self.can_reproj:
            self.cam_f = self.cam_f.reshape((-1, 1, 2))
            self.cam_cent = self.cam_cent.reshape((-1, 1, 2))
        self.chamfer_dist = None
        if use_chamfer:
            from .chamfer import ChamferDistance
            self.chamfer_dist = ChamferDistance(_lib_override=_lib_override)
        total = {k: sum(w[k] for w in self.all_stage_loss_weights) for k in self.loss_weights}
        self.init_motion_prior = None
        if init_motion_prior is not None and total['init_motion_prior'] > 0.0:
            w, mu, cov = init_motion_prior['gmm']
            self.init_motion_prior = {'gmm': _GMM(w, mu, cov)}
        if robust_loss not in ['none', 'bisquare', 'gm']:
            raise ValueError('Not a valid robust loss: %s' % robust_loss)
        self.robust_loss = robust_loss
        self.robust_tuning_const = robust_tuning_const
        self.cur_optim_step = 0
        self._op_conf_mask = None
        self._dev_cache = {}

    def _idx(self, name, values, device):
        """Index tables as device tensors, created once per device (an H2D copy per closure otherwise)."""
        key = (name, str(device))
        if key not in self._dev_cache:
            self._dev_cache[key] = torch.as_tensor(list(values), dtype=torch.long, device=device)
        return self._dev_cache[key]

    # ------------------------------------------------------------------------------------------------
    # fused evaluation (one kernel for all terms and their gradients)
    # ------------------------------------------------------------------------------------------------
    def _fusable(self, observed_data, ref):
        if not self.fused or 'prev_batch_overlap_res' in observed_data or 'points3d' in observed_data:
            return False
        if 'joints2d' in observed_data and not self._smpl2op_fusable:
            return False
        if ref.is_cuda:
            return True
        return self._lib is not None and self._lib.emulator

    def _const(self, key, make):
        if key not in self._dev_cache:
            self._dev_cache[key] = make()
        return self._dev_cache[key]

    def _fused_fit(self, kind, observed_data, pred_data, cam_pred_data, nsteps, cond_prior=None, init_motion_scale=1.0, halo=None):
        """kind: 'root' | 'smpl' | 'motion' -- the same presence / weight conditions as root_fit / smpl_fit / motion_fit."""
        from . import _lib
        W = self.loss_weights
        cam = cam_pred_data
        lib = self._lib if self._lib is not None else _lib.get_lib()
        w = [0.0] * FK.NT
        spec = {}
        t = dict.fromkeys([n for n, _ in FK.DIFF_INPUTS])
        has = lambda d, k: k in d and d[k] is not None
        # camera-frame joints: the full Jtr tensor when smpl_results provides it, else joints3d (+ extra)
        jtr = cam.get('jtr')
        if jtr is None and 'joints3d' in cam:
            jtr = torch.cat([cam['joints3d'], cam['joints3d_extra']], dim=2) if 'joints3d_extra' in cam else cam['joints3d']
        dev = (jtr if jtr is not None else cam['verts3d']).device
        B, T = (jtr if jtr is not None else cam['verts3d']).shape[:2]
        spec.update(B=B, T=T, nsteps=nsteps)
        c = lambda x: x if x.is_contiguous() else x.contiguous()
        if jtr is not None:
            t['cam_jtr'] = c(jtr)
            spec['nj'] = jtr.shape[2]
            if 'joints3d' in observed_data and W['joints3d'] > 0.0:
                w[FK.J3D] = W['joints3d']
                spec['obs_j3d'] = c(observed_data['joints3d'])
            if 'joints2d' in observed_data and 'joints3d_extra' in cam and W['joints2d'] > 0.0:
                if not self.can_reproj:
                    raise RuntimeError('Must provide camera intrinsics and SMPL to OpenPose joint map to use re-projection loss!')
                w[FK.J2D] = W['joints2d']
                spec['obs_j2d'] = c(observed_data['joints2d'])
                spec['smpl2op'] = self._const(('smpl2op32', str(dev)), lambda: self.smpl2op_map.to(device=dev, dtype=torch.int32).contiguous())

                def mk_mask():
                    m = torch.ones(OP_NUM_JOINTS, device=dev, dtype=torch.float32)
                    if self.ignore_op_joints is not None:
                        m[list(self.ignore_op_joints)] = 0.0
                    return m
                spec['op_mask'] = self._const(('op_mask', str(dev)), mk_mask)
                spec['cam_f'] = self._const(('cam_f', str(dev), B), lambda: self.cam_f.reshape(-1, 2).expand(B, 2).contiguous().float())
                spec['cam_c'] = self._const(('cam_c', str(dev), B), lambda: self.cam_cent.reshape(-1, 2).expand(B, 2).contiguous().float())
                spec['sigma'] = self.joints2d_sigma
            if kind != 'root' and W['joints3d_smooth'] > 0.0:
                w[FK.SMOOTH] = W['joints3d_smooth']
        if 'verts3d' in cam:
            t['cam_verts'] = c(cam['verts3d'])
            spec['nv'] = cam['verts3d'].shape[2]
            if 'verts3d' in observed_data and W['verts3d'] > 0.0:
                w[FK.V3D] = W['verts3d']
                spec['obs_v3d'] = c(observed_data['verts3d'])
        ovw = W['rgb_overlap_consist'] if 'seq_interval' in observed_data else 0.0
        if ovw > 0.0:
            vals = observed_data['seq_interval'].tolist()
            prev_tail = halo.get('prev_tail') if halo is not None else None
            ovs = [min(T, int(vals[b - 1][1]) - int(vals[b][0])) for b in range(1, len(vals))]
            ovs = ([] if prev_tail is not None else [0]) + ovs
            if len(ovs) != B:
                raise ValueError('seq_interval does not match the number of (local) sub-sequences')
            spec['overlap'] = self._const(('overlap', str(dev), tuple(ovs)), lambda: torch.tensor(ovs, dtype=torch.int32, device=dev))
            if 'verts3d' in cam:
                w[FK.OV_VPOS] = w[FK.OV_VVEL] = ovw
                if prev_tail is not None:
                    t['prev_tail'] = c(prev_tail)
        if kind != 'root':
            if 'latent_pose' in cam and W['pose_prior'] > 0.0:
                w[FK.POSE_PRIOR] = W['pose_prior']
                t['latent_pose'] = c(cam['latent_pose'])
                spec['dlp'] = cam['latent_pose'].shape[-1]
            if 'betas' in cam:
                use_b = False
                if W['shape_prior'] > 0.0:
                    w[FK.SHAPE_PRIOR] = W['shape_prior']
                    use_b = True
                if ovw > 0.0:
                    w[FK.OV_BETAS] = ovw
                    use_b = True
                    if halo is not None and halo.get('prev_betas') is not None:
                        t['prev_betas'] = c(halo['prev_betas'])
                if use_b:
                    t['betas'] = c(cam['betas'])
                    spec['nb'] = cam['betas'].shape[-1]
        extra = None
        stats_extra = {}
        if kind == 'motion':
            pred = pred_data
            if 'latent_motion' in pred and W['motion_prior'] > 0.0:
                w[FK.MOTION_PRIOR] = W['motion_prior']
                lm = pred['latent_motion']
                t['latent_motion'] = c(lm)
                spec['S'], spec['dz'] = lm.shape[1], lm.shape[2]
                if cond_prior is not None:
                    t['prior_mu'], t['prior_var'] = c(cond_prior[0]), c(cond_prior[1])
            pj = pred.get('jtr')
            if pj is None and 'joints3d' in pred:
                pj = pred['joints3d']
            want_pj = False
            fold_gmm = False
            if all(k in pred for k in ('joints3d', 'joints_vel', 'trans_vel', 'root_orient_vel')) and W['init_motion_prior'] > 0.0:
                gmm = self.init_motion_prior['gmm']
                # folded into the fused kernel's launches (fit_kernels.FusedFit, spec['gmm']) when it can read frame 0 of the joints in
                # place: no separate autograd node, no accumulation launches for the gradients of the joints and the velocities
                fold_gmm = (self.fold_init_prior and pj is not None and gmm.means.shape[0] <= 64 and gmm.means.shape[1] == 138
                            and pj.dtype == torch.float32 and all(pred[k].dtype == torch.float32 for k in ('joints_vel', 'trans_vel', 'root_orient_vel')))
                if fold_gmm:
                    spec['gmm'] = dict(gmm=gmm, w=W['init_motion_prior'] * init_motion_scale)
                    t['joints_vel'], t['trans_vel'], t['root_orient_vel'] = (c(pred[k]) for k in ('joints_vel', 'trans_vel', 'root_orient_vel'))
                    want_pj = True
                else:
                    cur = self.init_motion_prior_loss(pred['joints3d'][:, 0:1], pred['joints_vel'], pred['trans_vel'], pred['root_orient_vel'])
                    extra = W['init_motion_prior'] * init_motion_scale * cur
                    stats_extra['init_motion_prior'] = cur
            if 'joints3d_rollout' in pred:
                ro = pred['joints3d_rollout']
                use_ro = False
                if pj is not None and W['joint_consistency'] > 0.0:
                    w[FK.JOINT_CONSIST] = W['joint_consistency']
                    use_ro = want_pj = True
                if W['bone_length'] > 0.0:
                    w[FK.BONE_LEN] = W['bone_length']
                    use_ro = True
                if 'joints3d' in observed_data and W['joints3d_rollout'] > 0.0:
                    w[FK.J3D_RO] = W['joints3d_rollout']
                    spec['obs_j3d'] = c(observed_data['joints3d'])
                    use_ro = True
                if use_ro:
                    t['ro_joints'] = c(ro)
            if 'contacts_conf' in pred and pj is not None and (W['contact_vel'] > 0.0 or W['contact_height'] > 0.0):
                w[FK.CONTACT_VEL], w[FK.CONTACT_H] = W['contact_vel'], W['contact_height']
                t['contacts_conf'] = c(pred['contacts_conf'])
                want_pj = True
            if want_pj:
                t['pri_joints'] = c(pj)
                spec['pri_nj'] = pj.shape[2]
            if 'floor_plane' in cam:
                use_f = False
                if 'floor_plane' in observed_data and W['floor_reg'] > 0.0:
                    w[FK.FLOOR_REG] = W['floor_reg']
                    spec['obs_floor'] = c(observed_data['floor_plane'])
                    use_f = True
                if ovw > 0.0:
                    w[FK.OV_FLOOR] = ovw
                    use_f = True
                    if halo is not None and halo.get('prev_floor') is not None:
                        t['prev_floor'] = c(halo['prev_floor'])
                if use_f:
                    t['floor'] = c(cam['floor_plane'])
        spec['w'] = w
        loss, terms, gmm_total = FK.FusedFit.apply(lib, spec, *[t[n] for n, _ in FK.DIFF_INPUTS])
        stats = {FK.TERM_NAMES[k]: terms[k] for k in range(FK.NT) if w[k] != 0.0}
        if 'gmm' in spec:
            stats_extra['init_motion_prior'] = gmm_total
        stats.update(stats_extra)
        if extra is not None:
            loss = loss + extra
        return loss, stats

    def set_stage(self, idx):
        ''' Sets the current stage index. Determines which loss weights are used '''
        self.cur_stage_idx = idx
        self.loss_weights = self.all_stage_loss_weights[idx]

    def forward(self):
        pass

    # ------------------------------------------------------------------------------------------------
    # stage objectives
    # ------------------------------------------------------------------------------------------------
    def root_fit(self, observed_data, pred_data, halo=None):
        '''
        For fitting just global root trans/orientation. Only data terms, no priors (fitting_loss.py:94-181).
        '''
        W = self.loss_weights
        ref = pred_data.get('joints3d', pred_data.get('verts3d'))
        if ref is not None and self._fusable(observed_data, ref):
            return self._fused_fit('root', observed_data, pred_data, pred_data, 1, halo=halo)
        stats = dict()
        loss = 0.0
        if 'joints3d' in observed_data and 'joints3d' in pred_data and W['joints3d'] > 0.0:
            cur = self.joints3d_loss(observed_data['joints3d'], pred_data['joints3d'])
            loss = loss + W['joints3d'] * cur
            stats['joints3d'] = cur
        if 'verts3d' in observed_data and 'verts3d' in pred_data and W['verts3d'] > 0.0:
            cur = self.verts3d_loss(observed_data['verts3d'], pred_data['verts3d'])
            loss = loss + W['verts3d'] * cur
            stats['verts3d'] = cur
        if 'points3d' in observed_data and 'points3d' in pred_data and W['points3d'] > 0.0:
            cur = self.points3d_loss(observed_data['points3d'], pred_data['points3d'])
            loss = loss + W['points3d'] * cur
            stats['points3d'] = cur
        if 'joints2d' in observed_data and 'joints3d' in pred_data and 'joints3d_extra' in pred_data and W['joints2d'] > 0.0:
            if not self.can_reproj:
                raise RuntimeError('Must provide camera intrinsics and SMPL to OpenPose joint map to use re-projection loss!')
            cur = self.joints2d_loss(observed_data['joints2d'], pred_data['joints3d'], pred_data['joints3d_extra'])
            loss = loss + W['joints2d'] * cur
            stats['joints2d'] = cur
        if 'seq_interval' in observed_data and 'verts3d' in pred_data and W['rgb_overlap_consist'] > 0.0:
            pos, vel = self.overlap_verts_loss(observed_data['seq_interval'], pred_data['verts3d'], halo)
            loss = loss + W['rgb_overlap_consist'] * pos + W['rgb_overlap_consist'] * vel
            stats['rgb_overlap_consist_verts3d_pos'] = pos
            stats['rgb_overlap_consist_verts3d_vel'] = vel
            if 'prev_batch_overlap_res' in observed_data and (halo is None or halo.get('first', True)):
                prev = observed_data['prev_batch_overlap_res']
                cur_ov = int(prev['seq_interval'][1]) - int(observed_data['seq_interval'][0, 0])
                T_pred = pred_data['verts3d'].size(1)
                ov_len = min(T_pred, cur_ov)
                prev_pos = prev['verts3d'][-cur_ov:][:ov_len]
                cur_pos = pred_data['verts3d'][0, :ov_len]
                p = self.verts3d_loss(prev_pos, cur_pos)
                v = 0.0
                if cur_ov > 1:
                    v = self.verts3d_loss(prev_pos[1:] - prev_pos[:-1], cur_pos[1:] - cur_pos[:-1])
                loss = loss + W['rgb_overlap_consist'] * p + W['rgb_overlap_consist'] * v
                stats['rgb_overlap_xbatch_verts3d_pos'] = p
                stats['rgb_overlap_xbatch_verts3d_vel'] = v
        return loss, stats

    def overlap_verts_loss(self, seq_interval, verts3d, halo=None):
        """Consistency of the overlapping frames of consecutive sub-sequences (fitting_loss.py:135-157).
        verts3d [Bl,T,43,3] are the local sequences; with `halo`, halo['prev_tail'] [T,43,3] is the full predicted
        verts3d of the sequence before the first local one (None on the first rank) and seq_interval is already the
        local slice extended by that predecessor's interval in row 0."""
        iv = seq_interval
        prev_tail = halo.get('prev_tail') if halo is not None else None
        nb = verts3d.size(0) + (1 if prev_tail is not None else 0)
        vals = iv.tolist()      # one host read of the (CPU) interval table instead of two scalar reads per pair
        ovs = [int(vals[b - 1][1]) - int(vals[b][0]) for b in range(1, nb)]
        pos = verts3d.new_zeros(())
        vel = verts3d.new_zeros(())
        if not ovs:
            return pos, vel
        if all(o == ovs[0] for o in ovs) and ovs[0] > 0:
            # regular splitting (rgb_dataset.py:74-93): every pair overlaps by the same number of frames -> one batched term
            ov = ovs[0]
            allv = verts3d if prev_tail is None else torch.cat([prev_tail.unsqueeze(0), verts3d], dim=0)
            prev_pos, cur_pos = allv[:-1, -ov:], allv[1:, :ov]
            pos = self.verts3d_loss(prev_pos, cur_pos)
            if ov > 1:
                vel = self.verts3d_loss(prev_pos[:, 1:] - prev_pos[:, :-1], cur_pos[:, 1:] - cur_pos[:, :-1])
            return pos, vel
        seqs = ([prev_tail] if prev_tail is not None else []) + [verts3d[b] for b in range(verts3d.size(0))]
        for b in range(1, len(seqs)):
            ov = ovs[b - 1]
            prev_pos = seqs[b - 1][-ov:] if ov > 0 else seqs[b - 1][:0]
            cur_pos = seqs[b][:ov]
            pos = pos + self.verts3d_loss(prev_pos, cur_pos)
            if ov > 1:
                vel = vel + self.verts3d_loss(prev_pos[1:] - prev_pos[:-1], cur_pos[1:] - cur_pos[:-1])
        return pos, vel

    def add_next_side(self, loss, kind, cam_pred_data, halo):
        """Sharded closures, halo option (B): the consistency terms between this rank's LAST sequence and the next rank's first one
        (`halo['next_*']`, detached copies).  The pair's value is counted by the rank that owns the later sequence; here it only
        contributes its gradient w.r.t. this rank's variables (the terms are squared differences, fitting_loss.py:135-157,
        211-215, 296-300), so `g - g.detach()` is added: zero value, exact gradient."""
        W = self.loss_weights['rgb_overlap_consist']
        if halo is None or halo.get('next_head') is None or W <= 0.0:
            return loss
        g = loss.new_zeros(())
        ov = halo['ov_next']
        if ov > 0 and 'verts3d' in cam_pred_data:
            prev_pos, cur_pos = cam_pred_data['verts3d'][-1, -ov:], halo['next_head'][:ov]
            g = g + self.verts3d_loss(prev_pos, cur_pos)
            if ov > 1:
                g = g + self.verts3d_loss(prev_pos[1:] - prev_pos[:-1], cur_pos[1:] - cur_pos[:-1])
        if kind != 'root' and 'betas' in cam_pred_data:
            g = g + self.joints3d_loss(cam_pred_data['betas'][-1], halo['next_betas'])
        if kind == 'motion' and 'floor_plane' in cam_pred_data and halo.get('next_floor') is not None:
            g = g + self.joints3d_loss(cam_pred_data['floor_plane'][-1], halo['next_floor'])
        g = W * g
        return loss + (g - g.detach())

    def smpl_fit(self, observed_data, pred_data, nsteps, halo=None):
        '''
        For fitting full shape and pose of SMPL (fitting_loss.py:183-224).  nsteps scales single-step terms.
        '''
        W = self.loss_weights
        ref = pred_data.get('joints3d', pred_data.get('verts3d'))
        if ref is not None and self._fusable(observed_data, ref):
            return self._fused_fit('smpl', observed_data, pred_data, pred_data, nsteps, halo=halo)
        fused, self.fused = self.fused, False          # term-by-term path: root_fit must not take the fused branch on its own
        try:
            loss, stats = self.root_fit(observed_data, pred_data, halo=halo)
        finally:
            self.fused = fused
        if 'latent_pose' in pred_data and W['pose_prior'] > 0.0:
            cur = torch.sum(pred_data['latent_pose'] ** 2)
            loss = loss + W['pose_prior'] * cur
            stats['pose_prior'] = cur
        if 'betas' in pred_data and W['shape_prior'] > 0.0:
            cur = torch.sum(pred_data['betas'] ** 2)
            loss = loss + W['shape_prior'] * nsteps * cur
            stats['shape_prior'] = cur
        if W['joints3d_smooth'] > 0.0:
            cur = self.joints3d_smooth_loss(pred_data['joints3d'])
            loss = loss + W['joints3d_smooth'] * cur
            stats['joints3d_smooth'] = cur
        if 'seq_interval' in observed_data and 'betas' in pred_data and W['rgb_overlap_consist'] > 0.0:
            betas = pred_data['betas']
            if halo is not None and halo.get('prev_betas') is not None:
                betas = torch.cat([halo['prev_betas'].unsqueeze(0), betas], dim=0)
            cur = self.joints3d_loss(betas[:-1], betas[1:])
            loss = loss + W['rgb_overlap_consist'] * cur
            stats['rgb_overlap_consist_betas'] = cur
            if 'prev_batch_overlap_res' in observed_data and (halo is None or halo.get('first', True)):
                cur = self.joints3d_loss(pred_data['betas'][0], observed_data['prev_batch_overlap_res']['betas'])
                loss = loss + W['rgb_overlap_consist'] * cur
                stats['rgb_overlap_xbatch_betas'] = cur
        return loss, stats

    def motion_fit(self, observed_data, pred_data, cam_pred_data, nsteps, cond_prior=None, init_motion_scale=1.0, halo=None):
        '''
        For fitting full shape and pose of SMPL with the motion prior (fitting_loss.py:226-309).
        pred_data lives in the prior (canonical) frame, cam_pred_data in the camera frame.
        '''
        W = self.loss_weights
        ref = cam_pred_data.get('joints3d', cam_pred_data.get('verts3d'))
        if ref is not None and self._fusable(observed_data, ref):
            return self._fused_fit('motion', observed_data, pred_data, cam_pred_data, nsteps, cond_prior=cond_prior,
                                   init_motion_scale=init_motion_scale, halo=halo)
        fused, self.fused = self.fused, False
        try:
            loss, stats = self.smpl_fit(observed_data, cam_pred_data, nsteps, halo=halo)
        finally:
            self.fused = fused
        if 'latent_motion' in pred_data and W['motion_prior'] > 0.0:
            cur = self.motion_prior_loss(pred_data['latent_motion'], cond_prior=cond_prior)
            loss = loss + W['motion_prior'] * cur
            stats['motion_prior'] = cur
        have_init = all(k in pred_data for k in ('joints3d', 'joints_vel', 'trans_vel', 'root_orient_vel'))
        if have_init and W['init_motion_prior'] > 0.0:
            cur = self.init_motion_prior_loss(pred_data['joints3d'][:, 0:1], pred_data['joints_vel'], pred_data['trans_vel'],
                                              pred_data['root_orient_vel'])
            loss = loss + W['init_motion_prior'] * init_motion_scale * cur
            stats['init_motion_prior'] = cur
        if 'joints3d_rollout' in pred_data and 'joints3d' in pred_data and W['joint_consistency'] > 0.0:
            cur = 0.5 * torch.sum((pred_data['joints3d'] - pred_data['joints3d_rollout']) ** 2)
            loss = loss + W['joint_consistency'] * cur
            stats['joint_consistency'] = cur
        if 'joints3d_rollout' in pred_data and W['bone_length'] > 0.0:
            cur = self.bone_length_loss(pred_data['joints3d_rollout'])
            loss = loss + W['bone_length'] * cur
            stats['bone_length'] = cur
        if 'joints3d' in observed_data and 'joints3d_rollout' in pred_data and W['joints3d_rollout'] > 0.0:
            cur = self.joints3d_loss(observed_data['joints3d'], pred_data['joints3d_rollout'])
            loss = loss + W['joints3d_rollout'] * cur
            stats['joints3d_rollout'] = cur
        if W['contact_vel'] > 0.0 and 'contacts_conf' in pred_data and 'joints3d' in pred_data:
            cur = self.contact_vel_loss(pred_data['contacts_conf'], pred_data['joints3d'])
            loss = loss + W['contact_vel'] * cur
            stats['contact_vel'] = cur
        if W['contact_height'] > 0.0 and 'contacts_conf' in pred_data and 'joints3d' in pred_data:
            cur = self.contact_height_loss(pred_data['contacts_conf'], pred_data['joints3d'])
            loss = loss + W['contact_height'] * cur
            stats['contact_height'] = cur
        if W['floor_reg'] > 0.0 and 'floor_plane' in cam_pred_data and 'floor_plane' in observed_data:
            cur = self.floor_reg_loss(cam_pred_data['floor_plane'], observed_data['floor_plane'])
            loss = loss + W['floor_reg'] * nsteps * cur
            stats['floor_reg'] = cur
        if 'seq_interval' in observed_data and 'floor_plane' in cam_pred_data and W['rgb_overlap_consist'] > 0.0:
            fp = cam_pred_data['floor_plane']
            if halo is not None and halo.get('prev_floor') is not None:
                fp = torch.cat([halo['prev_floor'].unsqueeze(0), fp], dim=0)
            cur = self.joints3d_loss(fp[:-1], fp[1:])
            loss = loss + W['rgb_overlap_consist'] * cur
            stats['rgb_overlap_consist_floor'] = cur
            if 'prev_batch_overlap_res' in observed_data and (halo is None or halo.get('first', True)):
                cur = self.floor_reg_loss(cam_pred_data['floor_plane'][0:1],
                                          observed_data['prev_batch_overlap_res']['floor_plane'].unsqueeze(0))
                loss = loss + W['rgb_overlap_consist'] * cur
                stats['rgb_overlap_xbatch_floor'] = cur
        return loss, stats

    # ------------------------------------------------------------------------------------------------
    # terms
    # ------------------------------------------------------------------------------------------------
    def get_visible_mask(self, obs_data):
        return torch.logical_not(torch.isinf(obs_data))

    def joints2d_loss(self, joints2d_obs, joints3d_pred, joints3d_extra_pred, cam_t=None, cam_R=None, debug_img=None):
        '''GMoF re-projection error weighted by squared detection confidence (fitting_loss.py:317-358).'''
        if cam_t is not None or cam_R is not None:
            raise NotImplementedError('camera extrinsics are the identity on the fitting path')
        B, T = joints2d_obs.size(0), joints2d_obs.size(1)
        full = torch.cat([joints3d_pred, joints3d_extra_pred], dim=2)
        op = full.index_select(2, self._idx('smpl2op', self.smpl2op_map.tolist(), full.device)).reshape(B * T, OP_NUM_JOINTS, 3)
        cam_f = self.cam_f.expand(B, T, 2).reshape(B * T, 2)
        cam_c = self.cam_cent.expand(B, T, 2).reshape(B * T, 2)
        pred2d = perspective_projection(op, cam_f, cam_c).reshape(B, T, OP_NUM_JOINTS, 2)
        conf = joints2d_obs[:, :, :, 2:3]
        if self.ignore_op_joints is not None:
            # the reference zeroes these confidences IN PLACE on the observations (G5); same effect, no mutation
            if self._op_conf_mask is None or self._op_conf_mask.device != conf.device:
                m = torch.ones(OP_NUM_JOINTS, device=conf.device, dtype=conf.dtype)
                m[list(self.ignore_op_joints)] = 0.0
                self._op_conf_mask = m.reshape(1, 1, OP_NUM_JOINTS, 1)
            conf = conf * self._op_conf_mask
        err = (conf ** 2) * gmof(pred2d - joints2d_obs[:, :, :, :2], self.joints2d_sigma)
        return torch.sum(err)

    def joints3d_loss(self, joints3d_obs, joints3d_pred):
        return _masked_sq(joints3d_obs, joints3d_pred)

    def verts3d_loss(self, verts3d_obs, verts3d_pred):
        return _masked_sq(verts3d_obs, verts3d_pred)

    def points3d_loss(self, points3d_obs, points3d_pred):
        """One-way (observation -> body) chamfer term with robust weighting per sequence (fitting_loss.py:378-396)."""
        if self.chamfer_dist is None:
            raise RuntimeError('FittingLoss(use_chamfer=True) is needed for the points3d term')
        B, T, N_obs, _ = points3d_obs.size()
        obs2pred, _ = self.chamfer_dist(points3d_obs.reshape(B * T, -1, 3), points3d_pred.reshape(B * T, -1, 3))
        obs2pred = obs2pred.reshape(B, T * N_obs)
        weighted, _ = apply_robust_weighting(obs2pred.sqrt(), robust_loss_type=self.robust_loss, robust_tuning_const=self.robust_tuning_const)
        return 0.5 * torch.sum(weighted)

    def joints3d_smooth_loss(self, joints3d_pred):
        d = joints3d_pred[:, 1:] - joints3d_pred[:, :-1]
        return 0.5 * torch.sum(d * d)

    def pose_prior_loss(self, latent_pose_pred):
        return torch.sum(latent_pose_pred ** 2)

    def shape_prior_loss(self, betas_pred):
        return torch.sum(betas_pred ** 2)

    def motion_prior_loss(self, latent_motion_pred, cond_prior=None):
        if cond_prior is None:
            return torch.sum(latent_motion_pred ** 2)
        pm, pv = cond_prior
        return -torch.sum(self.log_normal(latent_motion_pred, pm, pv))

    def init_motion_prior_loss(self, joints, joints_vel, trans_vel, root_orient_vel):
        B = joints.size(0)
        gmm = self.init_motion_prior['gmm']
        on_dev = joints.is_cuda or (self._lib is not None and self._lib.emulator)
        if self.fused and on_dev and gmm.means.shape[0] <= 64 and gmm.means.shape[1] <= 256 and joints.dtype == torch.float32:
            # value and gradient in two launches (csrc/gmm.hip); frame 0 of the joints is read in place (a strided row per sequence)
            from . import _lib as _libmod
            from .fit_kernels import GmmNll
            return GmmNll.apply(self._lib if self._lib is not None else _libmod.get_lib(), gmm, joints, joints_vel, trans_vel, root_orient_vel)
        state = torch.cat([joints.reshape(B, -1), joints_vel.reshape(B, -1), trans_vel.reshape(B, -1),
                           root_orient_vel.reshape(B, -1)], dim=-1)
        return -torch.sum(self.init_motion_prior['gmm'].log_prob(state))

    def joint_consistency_loss(self, smpl_joints3d, rollout_joints3d):
        return 0.5 * torch.sum((smpl_joints3d - rollout_joints3d) ** 2)

    def bone_length_loss(self, rollout_joints3d):
        bones = rollout_joints3d[:, :, 1:]
        parents = rollout_joints3d.index_select(2, self._idx('smpl_parents', SMPL_PARENTS[1:], rollout_joints3d.device))
        lengths = torch.norm(bones - parents, dim=-1)
        d = lengths[:, 1:] - lengths[:, :-1]
        return 0.5 * torch.sum(d * d)

    def contact_vel_loss(self, contacts_conf, joints3d):
        delta = (joints3d[:, 1:] - joints3d[:, :-1]) ** 2
        return 0.5 * torch.sum(delta.sum(dim=-1) * contacts_conf[:, 1:])

    def contact_height_loss(self, contacts_conf, joints3d):
        floor_diff = F.relu(torch.abs(joints3d[:, :, :, 2]) - CONTACT_HEIGHT_THRESH)
        return torch.sum(floor_diff * contacts_conf)

    def floor_reg_loss(self, pred_floor_plane, obs_floor_plane):
        obs3 = obs_floor_plane[:, :3] * obs_floor_plane[:, 3:]
        return 0.5 * torch.sum((pred_floor_plane - obs3) ** 2)

    def log_normal(self, x, m, v):
        lp = -torch.log(torch.sqrt(v)) - math.log(math.sqrt(2 * math.pi)) - ((x - m) ** 2 / (2 * v))
        return torch.sum(lp, dim=-1)
