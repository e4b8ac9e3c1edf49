"""Result writers in the reference's on-disk layout (humor/fitting/fitting_utils.py:274-396), so its evaluation /
visualisation scripts read our outputs unchanged: per sub-sequence directory with ``stage3_results.npz``
(betas[16], trans[T,3], root_orient[T,3], pose_body[T,63], contacts[T,22], floor_plane[4]), optional
``stage3_results_prior.npz`` (prior-frame trans / root_orient), ``gt_results.npz`` and ``observations.npz``."""
import os

import numpy as np


def _np(t):
    return t.detach().cpu().numpy()


def save_optim_result(cur_res_out_paths, optim_result, per_stage_results, gt_data=None, observed_data=None, data_type='RGB',
                      optim_floor=True, obs_img_paths=None, obs_mask_paths=None):
    betas, trans = _np(optim_result['betas']), _np(optim_result['trans'])
    root_orient, pose_body = _np(optim_result['root_orient']), _np(optim_result['pose_body'])
    contacts = _np(optim_result['contacts']) if 'contacts' in optim_result else None
    floor = _np(optim_result['floor_plane']) if 'floor_plane' in optim_result else None
    for b, out_dir in enumerate(cur_res_out_paths):
        os.makedirs(out_dir, exist_ok=True)
        d = dict(betas=betas[b], trans=trans[b], root_orient=root_orient[b], pose_body=pose_body[b])
        if contacts is not None:
            d['contacts'] = contacts[b]
        if floor is not None:
            d['floor_plane'] = floor[b]
        np.savez(os.path.join(out_dir, 'stage3_results.npz'), **d)
    if per_stage_results is not None and 'stage3' in per_stage_results and optim_floor and 'prior_trans' in per_stage_results['stage3']:
        p_trans = _np(per_stage_results['stage3']['prior_trans'])
        p_root = _np(per_stage_results['stage3']['prior_root_orient'])
        for b, out_dir in enumerate(cur_res_out_paths):
            d = dict(betas=betas[b], trans=p_trans[b], root_orient=p_root[b], pose_body=pose_body[b])
            if contacts is not None:
                d['contacts'] = contacts[b]
            np.savez(os.path.join(out_dir, 'stage3_results_prior.npz'), **d)
    if gt_data is not None and all(k in gt_data for k in ('betas', 'trans', 'root_orient', 'pose_body')):
        g_betas = _np(gt_data['betas'])
        if data_type not in ['PROX-RGB', 'PROX-RGBD'] and g_betas.ndim == 3:
            g_betas = g_betas[:, 0]
        for b, out_dir in enumerate(cur_res_out_paths):
            d = dict(betas=g_betas[b], trans=_np(gt_data['trans'])[b], root_orient=_np(gt_data['root_orient'])[b],
                     pose_body=_np(gt_data['pose_body'])[b])
            if 'contacts' in gt_data:
                d['contacts'] = _np(gt_data['contacts'])[b]
            np.savez(os.path.join(out_dir, 'gt_results.npz'), **d)
    if observed_data is not None:
        for b, out_dir in enumerate(cur_res_out_paths):
            d = {k: _np(v)[b] for k, v in observed_data.items() if hasattr(v, 'detach') and v.shape[0] == len(cur_res_out_paths)
                 and k not in ('prev_batch_overlap_res',)}
            if obs_img_paths is not None:
                d['img_paths'] = np.array(obs_img_paths)[:, b]
            if obs_mask_paths is not None:
                d['mask_paths'] = np.array(obs_mask_paths)[:, b]
            np.savez(os.path.join(out_dir, 'observations.npz'), **d)
