"""ORACLE (test infrastructure only -- never imported by the product path).

Flat-tensor CPU restatement (plain PyTorch, differentiable, fp32 or fp64) of the HuMoR roll-out path
the fitting closure evaluates, for the configuration the fitting pipeline uses
(in_rot_rep='mat', out_rot_rep='aa', steps_in=1, model_data_config='smpl+joints+contacts',
conditional prior, output_delta=True, canonicalize_input=False):

  rot_to_aa                 humor/utils/transforms.py:243-389  (R -> quaternion (4 masked branches) -> atan2 -> aa, NaN->0)
  world2aligned             humor/utils/transforms.py:17-42
  mlp_forward               humor/models/humor_model.py:1206-1241 (Linear -> [GroupNorm(16) -> ReLU -> cat z -> Linear]*)
  prior / decode_compose    humor/models/humor_model.py:407-418, 445-498
  frame changes             humor/models/humor_model.py:696-772
  roll_out                  humor/models/humor_model.py:785-1017 (the autoregressive loop)
  rollout_outputs           humor/fitting/motion_optimizer.py:959-998 (R->aa, concat frame 0, contacts)

Unlike the reference (dicts of [B,1,D] tensors) the state is kept as flat vectors:
  past_in [B,339] = trans 0:3 | trans_vel 3:6 | root R 6:15 | root_vel 15:18 | body R 18:207 | joints 207:273 | joints_vel 273:339
  dec_raw [B,216] = trans 0:3 | trans_vel 3:6 | root aa 6:9 | root_vel 9:12 | body aa 12:75 | joints 75:141 | joints_vel 141:207 | contacts 207:216
          (out_rot_rep '6d' / '9d': 6 / 9 floats per rotation instead of 3 -> 282 / 348 wide, same order)
  state   [B,348] = past_in layout + contacts 339:348

Pinned against the unmodified reference (imported with oracle/ref_loader.py) by tests/test_oracle.py and
by the committed golden vectors tests/golden/rollout_*.npz (generator: oracle/make_golden.py).
The reference's `torch.cross` without `dim` is NOT reproduced: cross is always taken
along the last axis, so never compare against the reference with exactly 3 rows.
"""
import torch
import torch.nn.functional as F

from oracle.lbs_restated import batch_rodrigues

D_IN, D_RAW, D_STATE, LATENT = 339, 216, 348, 48
NJ, NBODY = 22, 21
CONTACT_INDS = [0, 4, 5, 7, 8, 10, 11, 20, 21]   # humor/datasets/amass_utils.py:22-23


# ---------------------------------------------------------------------------------------------
# rotation conversions
# ---------------------------------------------------------------------------------------------
def rot_to_aa(R):
    """[N,3,3] -> [N,3], restating rotation_matrix_to_angle_axis incl. its branch masks and NaN patch."""
    m = R.transpose(1, 2)                       # the reference transposes first ("rmat_t")
    m00, m01, m02 = m[:, 0, 0], m[:, 0, 1], m[:, 0, 2]
    m10, m11, m12 = m[:, 1, 0], m[:, 1, 1], m[:, 1, 2]
    m20, m21, m22 = m[:, 2, 0], m[:, 2, 1], m[:, 2, 2]
    d2 = m22 < 1e-6
    d01 = m00 > m11
    d0n1 = m00 < -m11
    t0 = 1 + m00 - m11 - m22
    q0 = torch.stack([m12 - m21, t0, m01 + m10, m20 + m02], -1)
    t1 = 1 - m00 + m11 - m22
    q1 = torch.stack([m20 - m02, m01 + m10, t1, m12 + m21], -1)
    t2 = 1 - m00 - m11 + m22
    q2 = torch.stack([m01 - m10, m20 + m02, m12 + m21, t2], -1)
    t3 = 1 + m00 + m11 + m22
    q3 = torch.stack([t3, m12 - m21, m20 - m02, m01 - m10], -1)
    c0 = (d2 & d01).to(R.dtype).unsqueeze(1)
    c1 = (d2 & ~d01).to(R.dtype).unsqueeze(1)
    c2 = (~d2 & d0n1).to(R.dtype).unsqueeze(1)
    c3 = (~d2 & ~d0n1).to(R.dtype).unsqueeze(1)
    q = q0 * c0 + q1 * c1 + q2 * c2 + q3 * c3
    q = q / torch.sqrt(t0.unsqueeze(1) * c0 + t1.unsqueeze(1) * c1 + t2.unsqueeze(1) * c2 + t3.unsqueeze(1) * c3)
    q = q * 0.5
    q1_, q2_, q3_ = q[:, 1], q[:, 2], q[:, 3]
    s2 = q1_ * q1_ + q2_ * q2_ + q3_ * q3_
    s = torch.sqrt(s2)
    c = q[:, 0]
    two_theta = 2.0 * torch.where(c < 0.0, torch.atan2(-s, -c), torch.atan2(s, c))
    k = torch.where(s2 > 0.0, two_theta / s, 2.0 * torch.ones_like(s))
    aa = torch.stack([q1_ * k, q2_ * k, q3_ * k], -1)
    return torch.where(torch.isnan(aa), torch.zeros_like(aa), aa)


def world2aligned(R_root):
    """[B,3,3] root orientation -> [B,3,3] heading-alignment rotation (about z)."""
    right = -R_root[:, :, 0]
    xproj = right[:, 0:1] / (torch.norm(right[:, :2], dim=1, keepdim=True) + 1e-6)
    xproj = torch.clamp(xproj, min=-1.0, max=1.0)
    angle = torch.acos(xproj)
    flat = right * torch.tensor([1.0, 1.0, 0.0], dtype=R_root.dtype)
    xaxis = torch.tensor([[1.0, 0.0, 0.0]], dtype=R_root.dtype).expand_as(flat)
    axis = torch.cross(flat, xaxis, dim=1)
    aa = axis / (torch.norm(axis, dim=1, keepdim=True) + 1e-6) * angle
    return batch_rodrigues(aa)


# ---------------------------------------------------------------------------------------------
# MLPs
# ---------------------------------------------------------------------------------------------
def mlp_params(sd, prefix):
    """Pulls (linears, groupnorms) out of a reference state_dict for `prefix` in {decoder, prior_net, encoder}."""
    # module list layout: Linear@0, then (GroupNorm@3k-2, ReLU@3k-1, Linear@3k) for k = 1, 2, ...
    lin = [(sd[f'{prefix}.net.0.weight'], sd[f'{prefix}.net.0.bias'])]
    gn = []
    k = 1
    while f'{prefix}.net.{3 * k}.weight' in sd:
        gn.append((sd[f'{prefix}.net.{3 * k - 2}.weight'], sd[f'{prefix}.net.{3 * k - 2}.bias']))
        lin.append((sd[f'{prefix}.net.{3 * k}.weight'], sd[f'{prefix}.net.{3 * k}.bias']))
        k += 1
    return lin, gn


# ReLU kinks.  The roll-out's gradient is discontinuous wherever a GroupNorm output crosses zero; a unit closer to zero than the rounding
# of an fp32 evaluation can legitimately sit on either side (two CORRECT fp32 implementations then return gradients that differ by
# 1e-3 .. 1e-2 of the largest entry).  KinkProbe lets the tests decide such cases from the REFERENCE side: it records, per ReLU site
# (network, step, layer), the units whose input is within tau[row] of zero, and it can force chosen units on / off -- the one-sided
# derivatives -- so that a gradient can be required to equal the oracle's on ONE of the branches the oracle itself cannot tell apart.
_RELU_HOOK = None


class KinkProbe:
    """hook(y, site) -> relu(y).  tau: float or [B] tensor (None: record nothing); force: {site: int8 [B, C], -1 natural / 0 off / 1 on}."""

    def __init__(self, tau=None, force=None):
        self.tau, self.force, self.near = tau, force or {}, []

    def __call__(self, y, site):
        mask = y > 0
        if self.tau is not None:
            tau = self.tau if not torch.is_tensor(self.tau) else self.tau.to(y.dtype).unsqueeze(1)
            yd = y.detach()
            for r, c in (yd.abs() < tau).nonzero().tolist():
                self.near.append((site, r, c, float(yd[r, c])))
        f = self.force.get(site)
        if f is not None:
            mask = torch.where(f >= 0, f > 0, mask)
        return y * mask.to(y.dtype)

    def __enter__(self):
        global _RELU_HOOK
        self._prev, _RELU_HOOK = _RELU_HOOK, self
        return self

    def __exit__(self, *exc):
        global _RELU_HOOK
        _RELU_HOOK = self._prev


def mlp_forward(x, lin, gn, skip=None, site=None):
    """Linear -> [GroupNorm(16 groups over channels of a 2-D input) -> ReLU -> (cat skip) -> Linear]*.
    site: (network, step) label of this evaluation for an active KinkProbe (the layer index is appended)."""
    h = F.linear(x, lin[0][0], lin[0][1])
    for li, ((w, b), (g, be)) in enumerate(zip(lin[1:], gn)):
        h = F.group_norm(h, 16, g, be, eps=1e-5)
        h = F.relu(h) if _RELU_HOOK is None or site is None else _RELU_HOOK(h, site + (li,))
        if skip is not None:
            h = torch.cat([h, skip], dim=1)
        h = F.linear(h, w, b)
    return h


# ---------------------------------------------------------------------------------------------
# one roll-out step
# ---------------------------------------------------------------------------------------------
def rot6d_to_rotmat(x):
    """[N,6] -> [N,3,3], restating rot6d_to_rotmat (humor/utils/transforms.py:201-220; cross along the last axis, see the header)."""
    x = x.reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = F.normalize(a1)
    b2 = F.normalize(a2 - torch.einsum('bi,bi->b', b1, a2).unsqueeze(-1) * b1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=1)), dim=-1)


def rot9d_to_rotmat(x):
    """[N,9] -> [N,3,3], restating rot9d_to_rotmat (humor/utils/transforms.py:222-241): U diag(1, 1, det(U V^T)) V^T of the SVD."""
    x = x.reshape(-1, 3, 3)
    u, s, v = torch.svd(x)
    v_T = v.transpose(-2, -1)
    s_p = torch.eye(3).to(x).reshape((1, 3, 3)).expand_as(x).clone()
    s_p[:, 2, 2] = torch.det(torch.matmul(u, v_T))
    return torch.matmul(torch.matmul(u, s_p), v_T)


RAW_WIDTH = {216: 3, 282: 6, 348: 9}     # decoder output width -> floats per joint of out_rot_rep ('aa', '6d', '9d')


def decode_compose(past_in, raw, output_delta=True):
    """decoder residual composition (humor_model.py:460-494): vectors add, rotations left-multiply.  The output rotation
    representation follows from the raw width: 216 'aa' (Rodrigues), 282 '6d', 348 '9d' (convert_to_rotmat, transforms.py:60-73).
    output_delta=False: the raw output is the state itself, only its rotations are converted (split_output, humor_model.py:331-347)."""
    B = past_in.shape[0]
    if not output_delta:
        w = RAW_WIDTH[raw.shape[1]]
        conv = {3: batch_rodrigues, 6: rot6d_to_rotmat, 9: rot9d_to_rotmat}[w]
        o_rvel, o_body, o_j = 6 + w, 9 + w, 9 + 22 * w
        return torch.cat([raw[:, 0:6], conv(raw[:, 6:6 + w]).reshape(B, 9), raw[:, o_rvel:o_rvel + 3],
                          conv(raw[:, o_body:o_body + NBODY * w].reshape(-1, w)).reshape(B, NBODY * 9), raw[:, o_j:o_j + 141]], dim=1)
    w = RAW_WIDTH[raw.shape[1]]
    conv = {3: batch_rodrigues, 6: rot6d_to_rotmat, 9: rot9d_to_rotmat}[w]
    o_rvel, o_body, o_j = 6 + w, 9 + w, 9 + 22 * w
    dR = conv(raw[:, 6:6 + w]).reshape(B, 3, 3)
    R_root = torch.matmul(dR, past_in[:, 6:15].reshape(B, 3, 3)).reshape(B, 9)
    dB = conv(raw[:, o_body:o_body + NBODY * w].reshape(-1, w)).reshape(B, NBODY, 3, 3)
    R_body = torch.matmul(dB, past_in[:, 18:207].reshape(B, NBODY, 3, 3)).reshape(B, NBODY * 9)
    return torch.cat([raw[:, 0:3] + past_in[:, 0:3], raw[:, 3:6] + past_in[:, 3:6], R_root,
                      raw[:, o_rvel:o_rvel + 3] + past_in[:, 15:18], R_body, raw[:, o_j:o_j + 66] + past_in[:, 207:273],
                      raw[:, o_j + 66:o_j + 132] + past_in[:, 273:339], raw[:, o_j + 132:o_j + 141]], dim=1)


def _rot_pts(Rm, pts):        # Rm [B,3,3], pts [B,K,3] -> [B,K,3]  (R @ p per point)
    return torch.einsum('bij,bkj->bki', Rm, pts)


def to_local(state, W, wt, t2j):
    """apply_world2local_trans(invert=False) on a [B,348|339] state -> next past_in [B,339]."""
    B = state.shape[0]
    trans = _rot_pts(W, (state[:, 0:3] + wt).unsqueeze(1))[:, 0]
    tvel = _rot_pts(W, state[:, 3:6].unsqueeze(1))[:, 0]
    R_root = torch.matmul(W, state[:, 6:15].reshape(B, 3, 3)).reshape(B, 9)
    rvel = _rot_pts(W, state[:, 15:18].unsqueeze(1))[:, 0]
    joints = (_rot_pts(W, state[:, 207:273].reshape(B, NJ, 3) + wt.unsqueeze(1) + t2j.unsqueeze(1))
              - t2j.unsqueeze(1)).reshape(B, NJ * 3)
    jvel = _rot_pts(W, state[:, 273:339].reshape(B, NJ, 3)).reshape(B, NJ * 3)
    return torch.cat([trans, tvel, R_root, rvel, state[:, 18:207], joints, jvel], dim=1)


def to_world(state, G, gt, t2j):
    """apply_world2local_trans(invert=True) with the accumulated (G, gt) -> world-frame state [B,348]."""
    B = state.shape[0]
    Gt = G.transpose(1, 2)
    trans = _rot_pts(Gt, state[:, 0:3].unsqueeze(1))[:, 0] - gt
    tvel = _rot_pts(Gt, state[:, 3:6].unsqueeze(1))[:, 0]
    R_root = torch.matmul(Gt, state[:, 6:15].reshape(B, 3, 3)).reshape(B, 9)
    rvel = _rot_pts(Gt, state[:, 15:18].unsqueeze(1))[:, 0]
    joints = (_rot_pts(Gt, state[:, 207:273].reshape(B, NJ, 3) + t2j.unsqueeze(1))
              - t2j.unsqueeze(1) - gt.unsqueeze(1)).reshape(B, NJ * 3)
    jvel = _rot_pts(Gt, state[:, 273:339].reshape(B, NJ, 3)).reshape(B, NJ * 3)
    return torch.cat([trans, tvel, R_root, rvel, state[:, 18:207], joints, jvel, state[:, 339:348]], dim=1)


def roll_out(sd, past_in0, z_seq, return_prior=True, eps_seq=None, G0=None, gt0=None, t2j=None, output_delta=True):
    """past_in0 [B,339] (already canonical), z_seq [B,S,48] -> world states [B,S,348], (pm, pv) [B,S,48] each.
    With z_seq=None the latent is sampled per step: z_t = pm_t + eps_seq[:, t] * sqrt(pv_t) (sample_step,
    humor_model.py:1029-1047; eps_seq=0 reproduces use_mean=True).  (G0, gt0, t2j) seed the accumulated world transform
    for canonicalize_input + uncanonicalize_output (humor_model.py:852-858)."""
    dec_lin, dec_gn = mlp_params(sd, 'decoder')
    pri_lin, pri_gn = mlp_params(sd, 'prior_net')
    sampling = z_seq is None
    B, S = (eps_seq if sampling else z_seq).shape[0], (eps_seq if sampling else z_seq).shape[1]
    dt = past_in0.dtype
    G = torch.eye(3, dtype=dt).unsqueeze(0).repeat(B, 1, 1) if G0 is None else G0
    gt = torch.zeros(B, 3, dtype=dt) if gt0 is None else gt0
    zero = torch.zeros(B, 1, dtype=dt)
    if t2j is None:
        t2j = -torch.cat([past_in0[:, 207:209], zero], dim=1)
    return_prior = return_prior or sampling
    past_in = past_in0
    world, pms, pvs = [], [], []
    for t in range(S):
        if return_prior:
            po = mlp_forward(past_in, pri_lin, pri_gn, site=('pri', t))
            pms.append(po[:, :LATENT])
            pvs.append(torch.exp(po[:, LATENT:]))
        z = pms[-1] + eps_seq[:, t] * torch.sqrt(pvs[-1]) if sampling else z_seq[:, t]
        raw = mlp_forward(torch.cat([past_in, z], dim=1), dec_lin, dec_gn, skip=z, site=('dec', t))
        pred = decode_compose(past_in, raw, output_delta)
        W = world2aligned(pred[:, 6:15].reshape(B, 3, 3))
        wt = torch.cat([-pred[:, 0:2], zero], dim=1)
        past_in = to_local(pred, W, wt, t2j)
        wstate = to_world(pred, G, gt, t2j)
        gt = torch.cat([-wstate[:, 0:2], zero], dim=1)
        G = torch.matmul(G, W)
        world.append(wstate)
    world = torch.stack(world, dim=1)
    if return_prior:
        return world, (torch.stack(pms, dim=1), torch.stack(pvs, dim=1))
    return world


def rollout_outputs(world):
    """What MotionOptimizer.rollout_latent_motion derives from the roll-out (motion_optimizer.py:959-998),
    for the predicted steps only (frame 0 is concatenated by the caller):
    trans [B,S,3], root_orient aa [B,S,3], pose_body aa [B,S,63], joints [B,S,22,3], contacts_conf/contacts [B,S,22]."""
    B, S = world.shape[0], world.shape[1]
    root_aa = rot_to_aa(world[:, :, 6:15].reshape(-1, 3, 3)).reshape(B, S, 3)
    body_aa = rot_to_aa(world[:, :, 18:207].reshape(-1, 3, 3)).reshape(B, S, 63)
    conf9 = torch.sigmoid(world[:, :, 339:348])
    lab9 = (conf9 > 0.5).to(world.dtype)
    conf = torch.zeros(B, S, NJ, dtype=world.dtype)
    lab = torch.zeros(B, S, NJ, dtype=world.dtype)
    conf[:, :, CONTACT_INDS] = conf9
    lab[:, :, CONTACT_INDS] = lab9
    return dict(trans=world[:, :, 0:3], root_orient=root_aa, pose_body=body_aa,
                joints=world[:, :, 207:273].reshape(B, S, NJ, 3), trans_vel=world[:, :, 3:6],
                root_orient_vel=world[:, :, 15:18], joints_vel=world[:, :, 273:339].reshape(B, S, NJ, 3),
                contacts_conf=conf, contacts=lab)
