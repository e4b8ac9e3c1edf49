"""Per-stage loss weights and camera defaults of the reference's fitting configurations (data only; no computation).

  AMASS_WEIGHTS   configs/fit_amass_joints.cfg:18-34        (3D joint observations, BASELINE config C2)
  RGB_WEIGHTS     configs/fit_rgb_demo_use_split.cfg:18-40  (2D OpenPose keypoints + floor + overlapping sub-sequences, C3/C4)
  CAM             configs/intrinsics_default.json
Each entry of a *_WEIGHTS list is one optimisation stage: {loss name: weight}, every key of LOSS_KEYS present
(humor/fitting/config.py:97-131 builds the same list of dicts from the per-stage 3-tuples of the cfg file)."""
import torch

LOSS_KEYS = ['joints2d', 'joints3d', 'joints3d_rollout', 'verts3d', 'points3d', 'pose_prior', 'shape_prior', 'motion_prior',
             'init_motion_prior', 'joint_consistency', 'bone_length', 'joints3d_smooth', 'contact_vel', 'contact_height', 'floor_reg',
             'rgb_overlap_consist']


def stage_weights(stage_vals):
    return [{k: float(v.get(k, 0.0)) for k in LOSS_KEYS} for v in stage_vals]


AMASS_WEIGHTS = stage_weights([
    {'joints3d': 1.0},
    {'joints3d': 1.0, 'pose_prior': 0.04, 'shape_prior': 0.05, 'joints3d_smooth': 0.1},
    {'joints3d': 1.0, 'shape_prior': 0.05, 'motion_prior': 0.01, 'init_motion_prior': 0.01, 'joint_consistency': 1.0,
     'bone_length': 10.0, 'contact_vel': 1.0, 'contact_height': 1.0}])
RGB_WEIGHTS = stage_weights([
    {'joints2d': 0.001, 'rgb_overlap_consist': 200.0},
    {'joints2d': 0.001, 'pose_prior': 0.04, 'shape_prior': 0.05, 'joints3d_smooth': 100.0, 'rgb_overlap_consist': 200.0},
    {'joints2d': 0.001, 'shape_prior': 0.05, 'motion_prior': 0.075, 'init_motion_prior': 0.075, 'joint_consistency': 100.0,
     'bone_length': 2000.0, 'contact_vel': 100.0, 'contact_height': 10.0, 'floor_reg': 0.167, 'rgb_overlap_consist': 200.0}])

CAM = dict(fx=1060.53, fy=1060.38, cx=951.30, cy=536.77)
# configs/fit_*.cfg: num_iter per stage and the stage-3 schedule (humor/fitting/config.py:108-109, 84-90)
NUM_ITER_RGB = [30, 80, 70]
NUM_ITER_AMASS = [30, 70, 70]
STAGE3_TUNE_INIT_FREEZE = (30, 50)


def camera_matrix(B):
    K = torch.zeros(B, 3, 3)
    K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = CAM['fx'], CAM['fy'], CAM['cx'], CAM['cy'], 1.0
    return K
