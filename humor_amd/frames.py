"""Coordinate-frame helpers of the fitting path that run once per fit or on [B]-sized tensors (host-side PyTorch on the
GPU; the per-step frame changes of the roll-out itself live in the HIP glue kernels).

  world2aligned_mat / canonicalize_pairs   humor/utils/transforms.py:17-42, humor/models/humor_model.py:1061-1124
  parse_floor_plane / plane_intersection / compute_cam2prior   humor/fitting/fitting_utils.py:61-104, 149-190
  estimate_linear_velocity / estimate_angular_velocity         humor/fitting/motion_optimizer.py:766-800
Cross products are always taken along the last axis (the reference's dim-less torch.cross misbehaves for 3 rows, G1).
"""
import torch


def _rodrigues_torch(aa):
    """Plain-torch Rodrigues with the reference's theta = ||r + 1e-8|| (used off the hot path, tiny batches)."""
    angle = torch.norm(aa + 1e-8, dim=1, keepdim=True)
    n = aa / angle
    c, s = torch.cos(angle).unsqueeze(1), torch.sin(angle).unsqueeze(1)
    z = torch.zeros_like(n[:, :1])
    K = torch.cat([z, -n[:, 2:3], n[:, 1:2], n[:, 2:3], z, -n[:, 0:1], -n[:, 1:2], n[:, 0:1], z], dim=1).view(-1, 3, 3)
    eye = torch.eye(3, dtype=aa.dtype, device=aa.device).unsqueeze(0)
    return eye + s * K + (1 - c) * torch.bmm(K, K)


def world2aligned_mat(R_root):
    """[M,3,3] root orientations -> [M,3,3] rotation about z aligning the body's right vector with +x."""
    right = -R_root[:, :, 0]
    x = right[:, 0:1] / (torch.norm(right[:, :2], dim=1, keepdim=True) + 1e-6)
    angle = torch.acos(torch.clamp(x, min=-1.0, max=1.0))
    flat = right * torch.tensor([1.0, 1.0, 0.0], dtype=R_root.dtype, device=R_root.device)
    xaxis = torch.tensor([[1.0, 0.0, 0.0]], dtype=R_root.dtype, device=R_root.device).expand_as(flat)
    axis = torch.cross(flat, xaxis, dim=1)
    aa = axis / (torch.norm(axis, dim=1, keepdim=True) + 1e-6) * angle
    return _rodrigues_torch(aa)


def canonicalize_pairs(seq, data_names):
    """All consecutive frame pairs (t, t+1) of a global sequence expressed in frame t's canonical system.
    seq: dict of [B,T,D] (rotations as 9-d matrices).  Returns x_past, x_t of shape [B*(T-1), D_total]."""
    B, T = seq['trans'].shape[0], seq['trans'].shape[1]
    M = B * (T - 1)
    cur = {k: v[:, :-1].reshape(M, -1) for k, v in seq.items()}
    nxt = {k: v[:, 1:].reshape(M, -1) for k, v in seq.items()}
    W = world2aligned_mat(cur['root_orient'].reshape(M, 3, 3))
    zero = torch.zeros(M, 1, dtype=W.dtype, device=W.device)
    wt = torch.cat([-cur['trans'][:, :2], zero], dim=1)
    # trans2joint is fixed by the FIRST frame of each sequence (humor_model.py:1088-1091)
    j0 = seq['joints'][:, 0, :2]
    wt0 = -seq['trans'][:, 0, :2]
    t2j_seq = torch.cat([-(j0 + wt0), torch.zeros(B, 1, dtype=W.dtype, device=W.device)], dim=1)
    t2j = t2j_seq.unsqueeze(1).expand(B, T - 1, 3).reshape(M, 3)

    def local(d):
        out = {}
        rot = lambda v: torch.einsum('mij,mj->mi', W, v)
        out['trans'] = rot(d['trans'] + wt)
        out['trans_vel'] = rot(d['trans_vel'])
        out['root_orient'] = torch.matmul(W, d['root_orient'].reshape(M, 3, 3)).reshape(M, 9)
        out['root_orient_vel'] = rot(d['root_orient_vel'])
        out['pose_body'] = d['pose_body']
        J = d['joints'].shape[1] // 3
        pts = d['joints'].reshape(M, J, 3) + wt.unsqueeze(1) + t2j.unsqueeze(1)
        out['joints'] = (torch.einsum('mij,mkj->mki', W, pts) - t2j.unsqueeze(1)).reshape(M, J * 3)
        out['joints_vel'] = torch.einsum('mij,mkj->mki', W, d['joints_vel'].reshape(M, J, 3)).reshape(M, J * 3)
        return out
    lc, ln = local(cur), local(nxt)
    x_past = torch.cat([lc[k] for k in data_names], dim=1)
    x_t = torch.cat([ln[k] for k in data_names], dim=1)
    return x_past, x_t


def parse_floor_plane(floor_plane):
    """[B,3] (normal * offset) -> [B,4] (a,b,c,d) with the normal pointing up in the camera frame (-y)."""
    off = torch.norm(floor_plane, dim=1, keepdim=True)
    normal = floor_plane / off
    neg = normal[:, 1:2] > 0.0
    normal = torch.where(neg.expand_as(normal), -normal, normal)
    off = torch.where(neg, -off, off)
    return torch.cat([normal, off], dim=1)


def plane_intersection(point, direction, plane):
    """Ray/plane intersection; returns (point + s * direction, s).  s < 0 means -direction hits the plane."""
    n, d = plane[:, :3], plane[:, 3]
    s = (d - (n * point).sum(-1)) / (n * direction).sum(-1)
    return point + s.unsqueeze(1) * direction, s


def compute_cam2prior(floor_plane, trans, root_orient_mat, joints):
    """Camera -> canonical (prior) frame from the floor and the key frame's root (fitting_utils.py:149-190).
    floor_plane [B,3|4], trans [B,3], root_orient_mat [B,3,3], joints [B,J,3] -> (R [B,3,3], t [B,3], root_height [B,1])."""
    B = floor_plane.size(0)
    plane = parse_floor_plane(floor_plane) if floor_plane.size(1) == 3 else floor_plane
    normal = plane[:, :3]
    floor_trans, _ = plane_intersection(trans, -normal, plane)
    body_right = -root_orient_mat[:, :, 0]
    floor_right, s = plane_intersection(trans, body_right, plane)
    right = floor_right - floor_trans
    right = torch.where(s.reshape(B, 1) < 0, -right, right)
    right = right / torch.norm(right, dim=1, keepdim=True)
    fwd = torch.cross(normal, right, dim=1)
    fwd = fwd / torch.norm(fwd, dim=1, keepdim=True)
    R = torch.stack([right, fwd, normal], dim=2).transpose(2, 1)
    _, s_root = plane_intersection(joints[:, 0], -normal, plane)
    return R, -trans, s_root.reshape(B, 1)


def estimate_linear_velocity(data_seq, h):
    """[B,T,...] -> [B,T,...]: forward / central / backward differences (motion_optimizer.py:766-783)."""
    init = (data_seq[:, 1:2] - data_seq[:, :1]) / h
    mid = (data_seq[:, 2:] - data_seq[:, 0:-2]) / (2 * h)
    fin = (data_seq[:, -1:] - data_seq[:, -2:-1]) / h
    return torch.cat([init, mid, fin], dim=1)


def estimate_angular_velocity(rot_seq, h):
    """[B,T,3,3] -> [B,T,3] from dR/dt R^T (motion_optimizer.py:785-800)."""
    dRdt = estimate_linear_velocity(rot_seq, h)
    w = torch.matmul(dRdt, rot_seq.transpose(-1, -2))
    wx = (-w[..., 1, 2] + w[..., 2, 1]) / 2.0
    wy = (w[..., 0, 2] - w[..., 2, 0]) / 2.0
    wz = (-w[..., 0, 1] + w[..., 1, 0]) / 2.0
    return torch.stack([wx, wy, wz], dim=-1)


def canonicalize_state(past_in):
    """[B,339] world-frame state -> the same state in its own heading-aligned frame, plus (R0, t0, t2j) to undo it
    (HumorModel.roll_out(canonicalize_input=True), humor_model.py:808-832)."""
    B = past_in.shape[0]
    R0 = world2aligned_mat(past_in[:, 6:15].reshape(B, 3, 3))
    zero = torch.zeros(B, 1, dtype=past_in.dtype, device=past_in.device)
    t0 = torch.cat([-past_in[:, 0:2], zero], dim=1)
    t2j = torch.cat([-(past_in[:, 207:209] + t0[:, :2]), zero], dim=1)
    rot = lambda v: torch.einsum('bij,bj->bi', R0, v)
    joints = torch.einsum('bij,bkj->bki', R0, past_in[:, 207:273].reshape(B, 22, 3) + t0.unsqueeze(1) + t2j.unsqueeze(1)) - t2j.unsqueeze(1)
    jvel = torch.einsum('bij,bkj->bki', R0, past_in[:, 273:339].reshape(B, 22, 3))
    local = torch.cat([rot(past_in[:, 0:3] + t0), rot(past_in[:, 3:6]), torch.matmul(R0, past_in[:, 6:15].reshape(B, 3, 3)).reshape(B, 9),
                       rot(past_in[:, 15:18]), past_in[:, 18:207], joints.reshape(B, 66), jvel.reshape(B, 66)], dim=1)
    return local, (R0, t0, t2j)


def uncanonicalize_world(world, R0, t0, t2j):
    """Maps canonical-frame roll-out outputs [B,S,348] back into the frame of the original input
    (uncanonicalize_output=True): the accumulated transform only gains the constant (R0, t0) factor."""
    B, S = world.shape[0], world.shape[1]
    Rt = R0.transpose(1, 2)
    rot = lambda v: torch.einsum('bij,bsj->bsi', Rt, v)
    joints = (torch.einsum('bij,bskj->bski', Rt, world[:, :, 207:273].reshape(B, S, 22, 3) + t2j.view(B, 1, 1, 3))
              - t2j.view(B, 1, 1, 3) - t0.view(B, 1, 1, 3))
    jvel = torch.einsum('bij,bskj->bski', Rt, world[:, :, 273:339].reshape(B, S, 22, 3))
    Rroot = torch.einsum('bij,bsjk->bsik', Rt, world[:, :, 6:15].reshape(B, S, 3, 3)).reshape(B, S, 9)
    return torch.cat([rot(world[:, :, 0:3]) - t0.unsqueeze(1), rot(world[:, :, 3:6]), Rroot, rot(world[:, :, 15:18]),
                      world[:, :, 18:207], joints.reshape(B, S, 66), jvel.reshape(B, S, 66), world[:, :, 339:348]], dim=2)


def window_to_local(win, W, wt, t2j):
    """The frame change of HumorModel.apply_world2local_trans(invert=False) (humor_model.py:696-772) on a window of states: win = {name:
    [B, S, d]} with rotations as 9-d matrices, W [B,3,3] the new frame's rotation, wt [B,3] its translation, t2j [B,3] the root-to-joint
    offset.  Positions: W (p + wt) (joints about t2j), directions / rotations: W v, W R; the body pose is frame-independent."""
    B = W.shape[0]
    out = {}
    for k, v in win.items():
        S = v.shape[1]
        if k == 'trans':
            out[k] = torch.einsum('bij,bsj->bsi', W, v + wt.unsqueeze(1))
        elif k in ('trans_vel', 'root_orient_vel'):
            out[k] = torch.einsum('bij,bsj->bsi', W, v)
        elif k == 'root_orient':
            out[k] = torch.einsum('bij,bsjk->bsik', W, v.reshape(B, S, 3, 3)).reshape(B, S, 9)
        elif k == 'joints':
            J = v.shape[2] // 3
            p = v.reshape(B, S, J, 3) + wt.view(B, 1, 1, 3) + t2j.view(B, 1, 1, 3)
            out[k] = (torch.einsum('bij,bskj->bski', W, p) - t2j.view(B, 1, 1, 3)).reshape(B, S, J * 3)
        elif k == 'joints_vel':
            J = v.shape[2] // 3
            out[k] = torch.einsum('bij,bskj->bski', W, v.reshape(B, S, J, 3)).reshape(B, S, J * 3)
        else:
            out[k] = v
    return out


def window_to_world(win, G, gt, t2j):
    """The inverse direction (invert=True): G^T p - gt (joints: G^T (p + t2j) - t2j - gt), G^T v, G^T R."""
    B = G.shape[0]
    Gt = G.transpose(1, 2)
    out = {}
    for k, v in win.items():
        S = v.shape[1]
        if k == 'trans':
            out[k] = torch.einsum('bij,bsj->bsi', Gt, v) - gt.unsqueeze(1)
        elif k in ('trans_vel', 'root_orient_vel'):
            out[k] = torch.einsum('bij,bsj->bsi', Gt, v)
        elif k == 'root_orient':
            out[k] = torch.einsum('bij,bsjk->bsik', Gt, v.reshape(B, S, 3, 3)).reshape(B, S, 9)
        elif k == 'joints':
            J = v.shape[2] // 3
            p = torch.einsum('bij,bskj->bski', Gt, v.reshape(B, S, J, 3) + t2j.view(B, 1, 1, 3))
            out[k] = (p - t2j.view(B, 1, 1, 3) - gt.view(B, 1, 1, 3)).reshape(B, S, J * 3)
        elif k == 'joints_vel':
            J = v.shape[2] // 3
            out[k] = torch.einsum('bij,bskj->bski', Gt, v.reshape(B, S, J, 3)).reshape(B, S, J * 3)
        else:
            out[k] = v
    return out
