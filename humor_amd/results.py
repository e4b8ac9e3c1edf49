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


def load_res(result_dir, file_name):
    """np.load of a result file as a dict, None if missing (fitting_utils.py:526-535)."""
    path = os.path.join(result_dir, file_name)
    if not os.path.exists(path):
        return None
    res = np.load(path, allow_pickle=True)
    return {k: res[k] for k in res.files}


def apply_cam2prior_seq(trans, root_orient, pose_body, betas, R, t, root_height, body_model):
    """Camera -> prior frame for ONE sequence [T,.] (fitting_utils.py:192-247 as used by the stitcher): rotate the root
    orientation, translate / rotate the root translation, then shift z so that the first frame's root joint sits at
    `root_height` above the floor."""
    import torch

    from . import ops
    lib = getattr(body_model, '_lib', None)
    T = trans.shape[0]
    Rm = ops.batch_rodrigues(root_orient, _lib_override=lib)                        # [T,3,3]
    new_R = torch.matmul(R.reshape(1, 3, 3), Rm)
    root_p = ops.rotation_matrix_to_angle_axis(new_R, _lib_override=lib)
    tr = torch.matmul(R.reshape(1, 3, 3), (trans + t.reshape(1, 3)).unsqueeze(-1)).squeeze(-1)
    body = body_model(pose_body=pose_body, pose_hand=None, betas=betas, root_orient=root_p, trans=tr)
    dh = root_height.reshape(()) - body.Jtr[0, 0, 2]
    tr = tr + torch.stack([torch.zeros_like(dh), torch.zeros_like(dh), dh]).reshape(1, 3)
    return tr, root_p


def save_rgb_stitched_result(seq_intervals, all_res_out_paths, res_out_path, device, body_model_path, num_betas, use_joints2d):
    """Stitches the per-sub-sequence results of a split RGB video into one sequence under <res_out_path>/final_results
    (humor/fitting/fitting_utils.py:398-523): every sub-sequence after the first drops the frames it shares with its
    predecessor; stage3_results.npz (camera frame, floor of the first sub-sequence), stage3_results_prior.npz (the whole
    sequence in the prior frame of frame 0 / that floor), gt_results.npz (cam_mtx), observations.npz (joints2d, img_paths),
    meta.txt.  Returns the final_results directory."""
    import shutil

    import torch

    from . import frames, ops
    from .body_model import BodyModel
    overlaps = [0] + [int(seq_intervals[i][1]) - int(seq_intervals[i + 1][0]) for i in range(len(seq_intervals) - 1)]
    out_dir = os.path.join(res_out_path, 'final_results')
    os.makedirs(out_dir, exist_ok=True)
    cat = {}
    contacts, floors, joints2d, img_paths, cam_mtx = [], [], [], [], None
    for i, res_dir in enumerate(all_res_out_paths):
        if i >= len(overlaps):          # an extra directory from even batching: no interval for it (fitting_utils.py:455-457)
            break
        cur = load_res(res_dir, 'stage3_results.npz')
        ov = overlaps[i]
        T = cur['trans'].shape[0]
        betas = cur['betas'] if cur['betas'].ndim == 2 else np.broadcast_to(cur['betas'][None], (T, cur['betas'].shape[0]))
        for k, v in (('betas', betas), ('trans', cur['trans']), ('root_orient', cur['root_orient']), ('pose_body', cur['pose_body'])):
            cat.setdefault(k, []).append(np.asarray(v)[ov:])
        contacts.append(cur['contacts'][ov:])
        floors.append(cur['floor_plane'].reshape(-1))
        if cam_mtx is None:
            gt = load_res(res_dir, 'gt_results.npz')
            cam_mtx = gt['cam_mtx'] if gt is not None and 'cam_mtx' in gt else None
        obs = load_res(res_dir, 'observations.npz')
        if obs is not None:
            if 'joints2d' in obs:
                joints2d.append(obs['joints2d'][ov:])
            if 'img_paths' in obs:
                img_paths += list(obs['img_paths'][ov:])
    cat = {k: np.concatenate(v, axis=0).astype(np.float32) for k, v in cat.items()}
    contacts = np.concatenate(contacts, axis=0)
    meta = os.path.join(all_res_out_paths[0], 'meta.txt')
    if os.path.exists(meta):
        shutil.copyfile(meta, os.path.join(out_dir, 'meta.txt'))
    if cam_mtx is not None:
        np.savez(os.path.join(out_dir, 'gt_results.npz'), cam_mtx=cam_mtx)
    if joints2d:
        np.savez(os.path.join(out_dir, 'observations.npz'), joints2d=np.concatenate(joints2d, axis=0), img_paths=np.array(img_paths))
    floor0 = floors[0]              # NOTE (reference): the estimate of the first sub-sequence is kept
    np.savez(os.path.join(out_dir, 'stage3_results.npz'), betas=cat['betas'], trans=cat['trans'], root_orient=cat['root_orient'],
             pose_body=cat['pose_body'], floor_plane=floor0, contacts=contacts)
    # the whole camera-frame sequence in the prior frame defined by frame 0 and the first floor
    N = cat['trans'].shape[0]
    bm = BodyModel(body_model_path, num_betas=num_betas, batch_size=N, use_vtx_selector=use_joints2d).to(device)
    d = {k: torch.from_numpy(v).to(device) for k, v in cat.items()}
    with torch.no_grad():
        body = bm(pose_body=d['pose_body'], pose_hand=None, betas=d['betas'], root_orient=d['root_orient'], trans=d['trans'])
        fp = torch.from_numpy(np.asarray(floor0, dtype=np.float32)).to(device).reshape(1, -1)
        R, t, h = frames.compute_cam2prior(fp, d['trans'][:1], ops.batch_rodrigues(d['root_orient'][:1], _lib_override=bm._lib), body.Jtr[:1])
        p_trans, p_root = apply_cam2prior_seq(d['trans'], d['root_orient'], d['pose_body'], d['betas'], R[0], t[0], h[0], bm)
    np.savez(os.path.join(out_dir, 'stage3_results_prior.npz'), betas=cat['betas'], trans=_np(p_trans), root_orient=_np(p_root),
             pose_body=cat['pose_body'], contacts=contacts)
    return out_dir
