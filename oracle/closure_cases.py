"""ORACLE (test infrastructure only).  Seeded synthetic fitting problems shared by the golden generator (which runs the
reference MotionOptimizer on them in the build container) and by the tests (which run humor_amd on the same problems).
Two cases: 'amass' (3D joint observations, no floor: BASELINE config C2 shape) and 'rgb' (2D OpenPose keypoints,
floor optimisation, overlapping sub-sequences: C3/C4 shape)."""
import numpy as np
import torch

from humor_amd.configs import AMASS_WEIGHTS, CAM, LOSS_KEYS as KEYS, RGB_WEIGHTS, camera_matrix, stage_weights as _weights   # noqa: F401 (data only)


# BASELINE-length cases: C2 fit_amass_joints (B=2, T=60), C3 fit_rgb_demo_no_split (one ~90-frame clip, no
# sub-sequence split), a C4 slice (fit_rgb_demo_use_split: 60-frame sub-sequences overlapping by 10; 8 of the 32) and the full C4 batch (all 32)
LONG_CASES = {
    'c2': dict(kind='amass', B=2, T=60, ov=None),
    'c3': dict(kind='rgb', B=1, T=90, ov=None),
    'c4': dict(kind='rgb', B=8, T=60, ov=10),
    # the benchmarked configuration itself (bench.py's workload): all 32 sub-sequences, i.e. one FULL 32-row tile
    'c4_full': dict(kind='rgb', B=32, T=60, ov=10),
}


def det_weights(shape, phase=0.0):
    """Deterministic pseudo-random cotangents cos(0.37 i + 1.3 sin(0.011 i) + phase): recomputed by the tests instead of being
    stored in the fixtures (the same to ~1 ulp on every machine; they only weight the outputs in a scalar test loss)."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    return torch.from_numpy(np.cos(0.37 * i + 1.3 * np.sin(0.011 * i) + phase).astype(np.float32).reshape(shape))


def make_case(kind, B, T, seed=0, ov=3):
    """Observations + a random evaluation point (values of every optimisation variable) for `kind`.
    ov: frames shared by consecutive sub-sequences ('rgb' only; None = a single unsplit clip, no seq_interval)."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: sc * torch.randn(*s, generator=g)
    case = {'kind': kind, 'B': B, 'T': T}
    var = {
        'trans': torch.cat([r(B, T, 2, sc=0.3), 4.0 + r(B, T, 1, sc=0.3)], 2) if kind == 'rgb' else r(B, T, 3, sc=0.3),
        'root_orient': torch.tensor([np.pi, 0.0, 0.0]) + r(B, T, 3, sc=0.2),
        'latent_pose': r(B, T, 32, sc=0.7),
        'betas': r(B, 16, sc=0.5),
        'latent_motion': r(B, T - 1, 48, sc=0.5),
        'trans_vel': r(B, 1, 3, sc=0.3), 'joints_vel': r(B, 1, 22, 3, sc=0.3), 'root_orient_vel': r(B, 1, 3, sc=0.3),
    }
    obs = {}
    if kind == 'amass':
        obs['joints3d'] = r(B, T, 22, 3, sc=0.5)
        obs['joints3d'][0, 2, 4] = float('inf')       # an occluded joint
    else:
        xy = torch.rand(B, T, 25, 2, generator=g) * torch.tensor([1900.0, 1000.0])
        conf = torch.rand(B, T, 25, 1, generator=g)
        conf[:, :, 5] = 0.0
        obs['joints2d'] = torch.cat([xy, conf], 3)
        obs['floor_plane'] = torch.tensor([[0.0, -1.0, 0.0, -0.5]]).expand(B, 4).clone()
        if ov is not None:
            obs['seq_interval'] = torch.tensor([[b * (T - ov), b * (T - ov) + T] for b in range(B)])
        var['floor_plane'] = torch.tensor([[0.02, 0.5, 0.03]]) + r(B, 3, sc=0.02)
    case['obs'], case['var'] = obs, var
    return case


def perturb_obs(obs, seed, eps=1e-6):
    """Every finite fp32 observation moved by `eps` relative, random sign (seeded): the size of an fp32 gradient's rounding error.
    Shared by the reference-side branch generator (oracle/make_golden_branches.py) and the short-run check of the tests, so both
    sides see the SAME perturbed problems."""
    g = torch.Generator().manual_seed(seed)
    out = dict(obs)
    for k in obs:
        if obs[k].dtype == torch.float32:
            sign = (torch.rand(obs[k].shape, generator=g) > 0.5).float() * 2 - 1
            out[k] = torch.where(torch.isfinite(obs[k]), obs[k] * (1.0 + sign * eps), obs[k])
    return out
