"""Seeded, shape-faithful synthetic stand-ins for the licence-gated assets of the fitting path.

Nothing here computes the hot path.  The real SMPL+H ``model.npz``, the HuMoR checkpoint, VPoser and
the init-state GMM cannot be redistributed, so tests / bench / smoke run on stand-ins that have the
same array names, shapes, dtypes and sparsity structure as the real files:

* ``make_smplh_npz``   -> dict with the keys ``BodyModel`` reads from ``model.npz``
                          (reference: humor/body_model/body_model.py:37-48): ``v_template [6890,3]``,
                          ``shapedirs [6890,3,16]``, ``posedirs [6890,3,459]``, ``J_regressor [52,6890]``,
                          ``weights [6890,52]`` (exactly 4 non-zeros per row), ``kintree_table [2,52]``
                          (the SMPL+H tree), ``f [13776,3]``.
* ``humor_state_dict`` -> a ``state_dict`` with the reference's key layout
                          ``{encoder,decoder,prior_net}.net.<idx>.{weight,bias}``
                          (reference: humor/models/humor_model.py:178-206, 1206-1241).
* ``SynthVPoser``      -> the three members the optimiser touches: ``latentD``, ``decode``, ``encode``
                          (reference: humor/fitting/motion_optimizer.py:77, 1041-1063).
* ``make_gmm``         -> (weights, means, covs) for the 138-d init-state prior
                          (reference: humor/fitting/run_fitting.py:248-261).
"""
import math

import numpy as np
import torch
import torch.nn as nn

NUM_VERTS = 6890
NUM_JOINTS = 52
NUM_BETAS = 16
NUM_FACES = 13776

# SMPL+H kinematic tree (kintree_table[0]); entry 0 is the root.
SMPLH_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19,
                 20, 22, 23, 20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35,
                 21, 37, 38, 21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50]


def _rest_skeleton():
    """A rough T-pose humanoid (y up, x to the body's left, metres) for the 52 SMPL+H joints."""
    J = np.zeros((NUM_JOINTS, 3), dtype=np.float64)
    J[0] = (0.0, -0.24, 0.03)            # pelvis
    J[1] = (0.07, -0.33, 0.02)           # L hip
    J[2] = (-0.07, -0.33, 0.02)          # R hip
    J[3] = (0.0, -0.12, 0.0)             # spine1
    J[4] = (0.10, -0.71, 0.01)           # L knee
    J[5] = (-0.10, -0.71, 0.01)          # R knee
    J[6] = (0.0, 0.02, 0.02)             # spine2
    J[7] = (0.09, -1.11, -0.03)          # L ankle
    J[8] = (-0.09, -1.11, -0.03)         # R ankle
    J[9] = (0.0, 0.08, 0.04)             # spine3
    J[10] = (0.12, -1.17, 0.09)          # L foot
    J[11] = (-0.12, -1.17, 0.09)         # R foot
    J[12] = (0.0, 0.29, 0.0)             # neck
    J[13] = (0.08, 0.20, 0.0)            # L collar
    J[14] = (-0.08, 0.20, 0.0)           # R collar
    J[15] = (0.0, 0.38, 0.04)            # head
    J[16] = (0.19, 0.23, -0.01)          # L shoulder
    J[17] = (-0.19, 0.23, -0.01)         # R shoulder
    J[18] = (0.45, 0.22, -0.03)          # L elbow
    J[19] = (-0.45, 0.22, -0.03)         # R elbow
    J[20] = (0.70, 0.23, -0.03)          # L wrist
    J[21] = (-0.70, 0.23, -0.03)         # R wrist
    # fingers: 5 chains of 3 per hand
    for hand, (wrist, sgn) in enumerate(((20, 1.0), (21, -1.0))):
        base = 22 + 15 * hand
        for f in range(5):
            zoff = 0.03 * (f - 2)
            for s in range(3):
                J[base + 3 * f + s] = J[wrist] + np.array([sgn * (0.08 + 0.03 * s), -0.005 * f, zoff])
    return J


def make_smplh_npz(seed=0, num_verts=NUM_VERTS, num_betas=NUM_BETAS, dtype=np.float32):
    """Returns a dict shaped exactly like the reference's SMPL+H ``model.npz``."""
    rng = np.random.RandomState(seed)
    parents = np.array(SMPLH_PARENTS, dtype=np.int64)
    J = _rest_skeleton()
    P = (NUM_JOINTS - 1) * 9

    # vertices: points scattered around bones (segment joint->parent), radius by body part
    # vertex ids are grouped by body part, as in the real SMPL topology (neighbouring ids share skinning joints)
    bone_of_vert = np.sort(rng.randint(0, NUM_JOINTS, size=num_verts))
    t = rng.rand(num_verts, 1)
    par = np.where(parents[bone_of_vert] < 0, bone_of_vert, parents[bone_of_vert])
    centre = J[bone_of_vert] * t + J[par] * (1.0 - t)
    radius = np.where(bone_of_vert >= 22, 0.01, np.where(bone_of_vert < 10, 0.09, 0.05))[:, None]
    direction = rng.randn(num_verts, 3)
    direction /= np.linalg.norm(direction, axis=1, keepdims=True)
    v_template = centre + direction * radius * (0.5 + 0.5 * rng.rand(num_verts, 1))

    # skinning weights: the 4 nearest joints, inverse-distance weighted, rows sum to 1
    d = np.linalg.norm(v_template[:, None, :] - J[None, :, :], axis=2)        # [V,52]
    nearest = np.argsort(d, axis=1)[:, :4]
    w4 = 1.0 / (np.take_along_axis(d, nearest, axis=1) + 1e-3) ** 2
    w4 /= w4.sum(axis=1, keepdims=True)
    weights = np.zeros((num_verts, NUM_JOINTS))
    np.put_along_axis(weights, nearest, w4, axis=1)

    # joint regressor: each joint = convex combination of its 24 nearest vertices (+ tiny dense floor)
    J_regressor = np.zeros((NUM_JOINTS, num_verts))
    nearv = np.argsort(d.T, axis=1)[:, :24]
    wj = rng.rand(NUM_JOINTS, 24) + 0.1
    np.put_along_axis(J_regressor, nearv, wj, axis=1)
    J_regressor /= J_regressor.sum(axis=1, keepdims=True)

    shapedirs = rng.randn(num_verts, 3, num_betas) * 0.01
    posedirs = rng.randn(num_verts, 3, P) * 0.001
    faces = rng.randint(0, num_verts, size=(NUM_FACES, 3)).astype(np.int64)
    kintree = np.stack([np.where(parents < 0, 2 ** 32 - 1, parents), np.arange(NUM_JOINTS)]).astype(np.int64)

    return {
        'v_template': v_template.astype(dtype),
        'shapedirs': shapedirs.astype(dtype),
        'posedirs': posedirs.astype(dtype),
        'J_regressor': J_regressor.astype(dtype),
        'weights': weights.astype(dtype),
        'kintree_table': kintree,
        'f': faces,
    }


def write_smplh_npz(path, seed=0, **kw):
    np.savez(path, **make_smplh_npz(seed=seed, **kw))
    return path


# --------------------------------------------------------------------------------------------------
# HuMoR weights (reference key layout; values = PyTorch default init under a fixed seed)
# --------------------------------------------------------------------------------------------------
HUMOR_IN_DIM = 339          # trans3 tvel3 R9 rvel3 body189 joints66 jvel66 ('mat' input rep)
HUMOR_OUT_DIM = 216         # trans3 tvel3 aa3 rvel3 body63 joints66 jvel66 contacts9 ('aa' output rep)
HUMOR_LATENT = 48


def _mlp_keys(prefix, sizes, skip=0):
    """(key, shape) list for reference ``MLP(layers=sizes, use_gn=True, skip_input_idx=...)``.
    net = [Linear, (GroupNorm, ReLU, Linear)*]: Linear idx 0,3,6,...; GroupNorm idx 1,4,7,...
    (reference: humor/models/humor_model.py:1206-1228)."""
    out = [(f'{prefix}.net.0.weight', (sizes[1], sizes[0])), (f'{prefix}.net.0.bias', (sizes[1],))]
    idx = 1
    for li in range(2, len(sizes)):
        c = sizes[li - 1]
        out += [(f'{prefix}.net.{idx}.weight', (c,)), (f'{prefix}.net.{idx}.bias', (c,))]
        out += [(f'{prefix}.net.{idx + 2}.weight', (sizes[li], c + skip)), (f'{prefix}.net.{idx + 2}.bias', (sizes[li],))]
        idx += 3
    return out


def humor_state_dict(seed=0, weight_scale=1.0, latent=HUMOR_LATENT, in_dim=HUMOR_IN_DIM, out_dim=HUMOR_OUT_DIM,
                     randomize_gn=True, out_scale=1.0, yaw_rate=None, past_steps=1):
    """Random weights in the reference checkpoint's key layout.  Linear layers use PyTorch's default
    kaiming-uniform bound 1/sqrt(fan_in); GroupNorm affine is perturbed around (1, 0) when
    ``randomize_gn`` so the affine path is exercised by parity tests.

    ``weight_scale`` scales every Linear weight, ``out_scale`` additionally scales the decoder's output layer (weight and
    bias), ``yaw_rate`` overrides the bias of the root-orientation delta about z (a steady turn, rad / step).  With the
    defaults the autoregressive chain is chaotic (it amplifies fp32 rounding ~1e4x over 59 steps, so no two fp32
    implementations agree at length); ``contractive_state_dict`` picks values for which the chain behaves like a trained
    model at 30 fps -- small per-frame deltas, bounded state -- and fp32 stays within ~1e-5 of fp64 over 119 steps."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    past = past_steps * in_dim      # HumorModel(steps_in=past_steps): the networks see the last `past_steps` states (humor_model.py:175-206)
    specs = (_mlp_keys('encoder', [past + in_dim, 1024, 1024, 1024, 1024, 2 * latent]) +
             _mlp_keys('decoder', [past + latent, 1024, 1024, 512, out_dim], skip=latent) +
             _mlp_keys('prior_net', [past, 1024, 1024, 1024, 1024, 2 * latent]))
    for key, shape in specs:
        if len(shape) == 2:
            bound = weight_scale / math.sqrt(shape[1])
            sd[key] = (torch.rand(shape, generator=g) * 2 - 1) * bound
            last_fan_in = shape[1]
        elif key.endswith('.weight'):     # GroupNorm gamma
            sd[key] = 1.0 + (0.1 * torch.randn(shape, generator=g) if randomize_gn else torch.zeros(shape))
        else:
            # GroupNorm beta or Linear bias (bias follows its weight in the list)
            prev_is_linear = key.replace('.bias', '.weight') in sd and sd[key.replace('.bias', '.weight')].dim() == 2
            if prev_is_linear:
                bound = 1.0 / math.sqrt(last_fan_in)
                sd[key] = (torch.rand(shape, generator=g) * 2 - 1) * bound
            else:
                sd[key] = 0.05 * torch.randn(shape, generator=g) if randomize_gn else torch.zeros(shape)
    if out_scale != 1.0:
        last = max(int(k.split('.')[2]) for k in sd if k.startswith('decoder.net.'))
        sd[f'decoder.net.{last}.weight'] = sd[f'decoder.net.{last}.weight'] * out_scale
        sd[f'decoder.net.{last}.bias'] = sd[f'decoder.net.{last}.bias'] * out_scale
    if yaw_rate is not None:
        last = max(int(k.split('.')[2]) for k in sd if k.startswith('decoder.net.'))
        sd[f'decoder.net.{last}.bias'][8] = yaw_rate        # decoder output layout: trans 0:3 | trans_vel 3:6 | root aa 6:9 | ...
    return sd


CONTRACTIVE = dict(weight_scale=0.5, out_scale=0.05, yaw_rate=0.1)


def contractive_state_dict(seed=0):
    """Well-conditioned synthetic HuMoR weights for parity tests at the BASELINE sequence lengths (59 / 89 / 119 steps).
    Two things make a random-init roll-out ill-conditioned, and neither is a property of the path: (i) O(1) decoder deltas
    per step let the state run away chaotically -> the output layer is scaled down (per-frame deltas of a 30 fps model);
    (ii) with tiny deltas the per-step heading change sits at the singularity of the reference's heading alignment,
    acos(x / (|xy| + 1e-6)) at x -> 1 (humor/utils/transforms.py:17-31), where fp32 resolves 1 - x to ~6 % -> a steady turn
    of 0.1 rad / step keeps it away.  Measured fp32-vs-fp64 drift of the reference roll-out with these weights
    (oracle/make_golden_long.py prints it): < 5e-6 in the states and prior outputs over 119 steps."""
    return humor_state_dict(seed=seed, **CONTRACTIVE)


def rotrep_state_dict(out_rot_rep, seed=0, yaw_rate=0.1):
    """contractive_state_dict for a model with out_rot_rep '6d' or '9d' (decoder output 282 / 348 wide, humor_model.py:100-140):
    the same scaled-down weights, and the bias of every rotation output set to the representation of the identity (body joints) or
    of a steady turn about z (root) -- what a trained residual decoder emits -- so that the per-step deltas stay small."""
    w = {'6d': 6, '9d': 9}[out_rot_rep]
    sd = humor_state_dict(seed=seed, out_dim=150 + 22 * w, weight_scale=CONTRACTIVE['weight_scale'], out_scale=CONTRACTIVE['out_scale'])
    last = max(int(k.split('.')[2]) for k in sd if k.startswith('decoder.net.'))
    bias = sd[f'decoder.net.{last}.bias']
    c, sn = math.cos(yaw_rate), math.sin(yaw_rate)

    def rep(R):        # R row-major 3x3 -> the representation whose conversion returns R
        if w == 9:
            return torch.tensor(R, dtype=torch.float32).reshape(9)
        return torch.tensor([R[0][0], R[0][1], R[1][0], R[1][1], R[2][0], R[2][1]], dtype=torch.float32)   # [3][2]: first two columns
    eye = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]
    bias[6:6 + w] += rep([[c, -sn, 0.0], [sn, c, 0.0], [0.0, 0.0, 1.0]])
    for j in range(21):
        bias[9 + w + j * w:9 + w + (j + 1) * w] += rep(eye)
    return sd


def nodelta_state_dict(seed=0):
    """Well-conditioned synthetic weights for a model with output_delta=False (the decoder emits the next state itself): scaled-down
    weights as in contractive_state_dict, a moderate output layer and a 0.3 rad heading so that the predicted root orientation stays
    away from the singularity of the heading alignment (see contractive_state_dict)."""
    return humor_state_dict(seed=seed, weight_scale=CONTRACTIVE['weight_scale'], out_scale=0.3, yaw_rate=0.3)


# --------------------------------------------------------------------------------------------------
# VPoser stand-in and init-state GMM
# --------------------------------------------------------------------------------------------------
class _Normalish:
    def __init__(self, mean, std):
        self.mean = mean
        self.scale = std


def _rot6d_to_mat(x):
    x = x.reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = torch.nn.functional.normalize(a1, dim=1)
    b2 = torch.nn.functional.normalize(a2 - (b1 * a2).sum(1, keepdim=True) * b1, dim=1)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack((b1, b2, b3), dim=-1)


class SynthVPoser(nn.Module):
    """Shape-faithful VPoser v1.0 stand-in: decoder 32->512->512->126 (6D -> R, returns [N,1,21,9]),
    encoder 63->512->512->(32 mean, 32 std)."""
    def __init__(self, seed=0, latentD=32, hidden=512):
        super().__init__()
        self.latentD = latentD
        torch.manual_seed(seed)
        self.dec = nn.Sequential(nn.Linear(latentD, hidden), nn.LeakyReLU(0.2), nn.Linear(hidden, hidden),
                                 nn.LeakyReLU(0.2), nn.Linear(hidden, 21 * 6))
        self.enc = nn.Sequential(nn.Linear(63, hidden), nn.LeakyReLU(0.2), nn.Linear(hidden, hidden), nn.LeakyReLU(0.2))
        self.enc_mu = nn.Linear(hidden, latentD)
        self.enc_std = nn.Linear(hidden, latentD)
        with torch.no_grad():
            # bias the 6D output towards identity so decoded poses are moderate rotations
            self.dec[-1].weight.mul_(0.3)
            self.dec[-1].bias.copy_(torch.tensor([1., 0., 0., 1., 0., 0.]).repeat(21))
        for p in self.parameters():
            p.requires_grad_(False)

    def decode(self, z, output_type='matrot'):
        assert output_type == 'matrot'
        n = z.shape[0]
        return _rot6d_to_mat(self.dec(z)).reshape(n, 1, 21, 9)

    def encode(self, pose_aa):
        h = self.enc(pose_aa)
        return _Normalish(self.enc_mu(h), torch.nn.functional.softplus(self.enc_std(h)))


def make_gmm(seed=0, ncomp=12, dim=138):
    g = torch.Generator().manual_seed(seed)
    w = torch.rand(ncomp, generator=g) + 0.1
    w = w / w.sum()
    means = 0.1 * torch.randn(ncomp, dim, generator=g)
    a = 0.05 * torch.randn(ncomp, dim, dim, generator=g)
    covs = a @ a.transpose(1, 2) + 0.5 * torch.eye(dim).unsqueeze(0)
    return w, means, covs


def smooth_pose_sequence(B, T, seed=0, amp=0.3):
    """Smooth random axis-angle motion: (root_orient[B,T,3], pose_body[B,T,63], trans[B,T,3])."""
    g = torch.Generator().manual_seed(seed)
    tt = torch.linspace(0, 1, T).view(1, T, 1)

    def smooth(d, a):
        f = torch.rand(B, 1, d, generator=g) * 2.0 + 0.5
        ph = torch.rand(B, 1, d, generator=g) * 6.28
        base = a * torch.randn(B, 1, d, generator=g)
        return base + a * 0.5 * torch.sin(6.28 * f * tt + ph)

    root = smooth(3, amp * 0.5)
    root[:, :, 0] += 0.2
    body = smooth(63, amp)
    trans = smooth(3, 0.3)
    trans[:, :, 2] += 0.9
    return root, body, trans
