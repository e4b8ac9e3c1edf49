"""ORACLE (test infrastructure only).  Seeded inputs for the HumorLoss parity tests (training path, humor/losses/humor_loss.py):
pred / gt dictionaries of one training step in the reference's `B x D` layout, mixed genders, betas.  Shared by the fixture
generator (oracle/make_golden_humor_loss.py) and the tests so both evaluate the same numbers."""
import numpy as np
import torch

NJ, NKV = 22, 43
WEIGHTS = dict(kl_loss=4e-4, kl_loss_anneal_start=0, kl_loss_anneal_end=50, regr_trans_loss=1.0, regr_trans_vel_loss=1.0,
               regr_root_orient_loss=1.0, regr_root_orient_vel_loss=1.0, regr_pose_loss=1.0, regr_pose_vel_loss=1.0,
               regr_joint_loss=1.0, regr_joint_vel_loss=1.0, regr_joint_orient_vel_loss=1.0, regr_vert_loss=1.0, regr_vert_vel_loss=1.0,
               contacts_loss=0.01, contacts_vel_loss=0.01, smpl_joint_loss=1.0, smpl_mesh_loss=1.0, smpl_joint_consistency_loss=1.0,
               smpl_vert_consistency_loss=1.0)
# no mesh term: the body model runs on the 43 key vertices + joints only (subset kernels); cyclic KL annealing
WEIGHTS_NO_MESH = dict(WEIGHTS, smpl_mesh_loss=0.0, kl_loss_anneal_end=0, kl_loss_cycle_len=50)
CASES = (('a', 24, 0, WEIGHTS), ('b', 7, 1, WEIGHTS), ('c', 5, 2, WEIGHTS_NO_MESH))
PRED_GRAD_KEYS = ['trans', 'root_orient', 'pose_body', 'joints', 'verts', 'joints_vel', 'contacts', 'trans_vel']


def _rotmats(n, g, scale):
    """n rotation matrices exp([w]x), |w| ~ scale, from a seeded generator (float64 Rodrigues, stored as float32)."""
    w = scale * torch.randn(n, 3, generator=g, dtype=torch.float64)
    th = w.norm(dim=1, keepdim=True).clamp_min(1e-12)
    k = w / th
    K = torch.zeros(n, 3, 3, dtype=torch.float64)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -k[:, 2], k[:, 1], k[:, 2], -k[:, 0], -k[:, 1], k[:, 0]
    th = th.view(n, 1, 1)
    R = torch.eye(3, dtype=torch.float64) + torch.sin(th) * K + (1 - torch.cos(th)) * (K @ K)
    return R.float()


def make_case(B=24, seed=0):
    g = torch.Generator().manual_seed(1000 + seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    gt = {
        'trans': 0.5 * rn(B, 3), 'trans_vel': 0.3 * rn(B, 3), 'root_orient_vel': 0.3 * rn(B, 3),
        'root_orient': _rotmats(B, g, 1.0).reshape(B, 9), 'pose_body': _rotmats(B * (NJ - 1), g, 0.4).reshape(B, (NJ - 1) * 9),
        'joints': 0.5 * rn(B, NJ * 3), 'joints_vel': 0.3 * rn(B, NJ * 3), 'verts': 0.5 * rn(B, NKV * 3), 'verts_vel': 0.3 * rn(B, NKV * 3),
        'contacts': (torch.rand(B, 9, generator=g) > 0.6).float(),
    }
    pred = {}
    for k, v in gt.items():
        if k == 'root_orient':
            pred[k] = (_rotmats(B, g, 0.15) @ v.reshape(B, 3, 3)).reshape(B, 9)
        elif k == 'pose_body':
            pred[k] = (_rotmats(B * (NJ - 1), g, 0.15) @ v.reshape(-1, 3, 3)).reshape(B, (NJ - 1) * 9)
        elif k == 'contacts':
            pred[k] = 2.0 * rn(B, 9)                         # logits
        else:
            pred[k] = v + 0.1 * rn(*v.shape)
    pred['posterior_distrib'] = (0.3 * rn(B, 48), torch.exp(0.3 * rn(B, 48)))
    pred['prior_distrib'] = (0.3 * rn(B, 48), torch.exp(0.3 * rn(B, 48)))
    gender = np.array([['male'] if i % 3 != 1 else ['female'] for i in range(B)])      # interleaved: exercises the re-ordering
    betas = 0.7 * rn(B, 16)
    return {'pred': pred, 'gt': gt, 'gender': gender, 'betas': betas, 'B': B, 'cur_epoch': 20}


def evaluate(loss_mod, case, device):
    """loss, stats (python floats) and d loss / d pred[k] for PRED_GRAD_KEYS (+ the posterior mean) with `loss_mod` on `device`."""
    pred = {}
    for k, v in case['pred'].items():
        pred[k] = tuple(t.clone().to(device).requires_grad_(True) for t in v) if isinstance(v, tuple) else v.clone().to(device).requires_grad_(True)
    gt = {k: v.clone().to(device) for k, v in case['gt'].items()}
    loss, stats = loss_mod(pred, gt, case['cur_epoch'], gender=case['gender'], betas=case['betas'].to(device))
    wrt = [pred[k] for k in PRED_GRAD_KEYS] + [pred['posterior_distrib'][0]]
    grads = torch.autograd.grad(loss, wrt, allow_unused=True)
    out = {'loss': float(loss.detach())}
    for k, v in stats.items():
        out['stat_' + k] = float(v.detach()) if torch.is_tensor(v) else float(v)
    for k, gk in zip(PRED_GRAD_KEYS + ['posterior_mean'], grads):
        out['grad_' + k] = (torch.zeros_like(wrt[0]) if gk is None else gk).detach().cpu().numpy()
    return out
