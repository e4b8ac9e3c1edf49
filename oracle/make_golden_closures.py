"""ORACLE (test infrastructure only).  Evaluates the REFERENCE MotionOptimizer's stage objectives (its own smpl_results /
rollout_latent_motion / FittingLoss methods composed exactly as its closures do, motion_optimizer.py:241-252, 291-304,
514-607) at seeded points and stores loss + gradients -> tests/golden/closure_<kind>.npz.  Also runs the reference
`run()` for a few iterations and stores the final variables/joints (short-trajectory fixture).
Build container only:  python -m oracle.make_golden_closures"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from humor_amd import synth                       # noqa: E402
from oracle import closure_cases as CC            # noqa: E402
from oracle import ref_loader                     # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def build_reference(R, kind, B, T, npz, state_dict=None):
    dev = torch.device('cpu')
    rgb = kind == 'rgb'
    bm = R.body_model.BodyModel(npz, num_betas=16, batch_size=B * T, use_vtx_selector=rgb).to(dev)
    hm = R.humor_model.HumorModel(in_rot_rep='mat', out_rot_rep='aa', latent_size=48, model_data_config='smpl+joints+contacts', steps_in=1)
    hm.load_state_dict(synth.humor_state_dict(seed=0) if state_dict is None else state_dict)
    hm.eval()
    vp = synth.SynthVPoser(seed=0)
    w, mu, cov = synth.make_gmm(seed=0)
    weights = CC.RGB_WEIGHTS if rgb else CC.AMASS_WEIGHTS
    opt = R.motion_optimizer.MotionOptimizer(dev, bm, 16, B, T, ['joints2d'] if rgb else ['joints3d'], weights, vp, hm,
                                             {'gmm': (w, mu, cov)}, optim_floor=rgb, camera_matrix=CC.camera_matrix(B) if rgb else None,
                                             robust_loss_type='bisquare', joint2d_sigma=100)
    return opt


def eval_stage(R, opt, case, stage):
    """The body of the reference's stage closure at the case's evaluation point; returns loss, dict of grads."""
    var = {k: v.clone().requires_grad_(True) for k, v in case['var'].items()}
    obs = {k: v.clone() for k, v in case['obs'].items()}
    B, T = case['B'], case['T']
    opt.fitting_loss.set_stage(stage)
    if stage < 2:
        opt.trans, opt.root_orient, opt.betas, opt.latent_pose = var['trans'], var['root_orient'], var['betas'], var['latent_pose']
        body_pose = opt.latent2pose(opt.latent_pose)
        pred, _ = opt.smpl_results(opt.trans, opt.root_orient, body_pose, opt.betas)
        if stage == 0:
            loss, _ = opt.fitting_loss.root_fit(obs, pred)
            wrt = ['trans', 'root_orient']
        else:
            pred['latent_pose'], pred['betas'] = opt.latent_pose, opt.betas
            loss, _ = opt.fitting_loss.smpl_fit(obs, pred, T)
            wrt = ['trans', 'root_orient', 'betas', 'latent_pose']
    else:
        # stage 3: variables are the first frame + latent motion + initial velocities (+ floor)
        first = lambda k: var[k][:, :1]
        opt.trans, opt.root_orient, opt.latent_pose = first('trans'), first('root_orient'), first('latent_pose')
        opt.betas, opt.latent_motion = var['betas'], var['latent_motion']
        opt.trans_vel, opt.joints_vel, opt.root_orient_vel = var['trans_vel'], var['joints_vel'], var['root_orient_vel']
        prior_opt_params = [opt.trans_vel, opt.joints_vel, opt.root_orient_vel]
        cur_body_pose = opt.latent2pose(opt.latent_pose)
        if opt.optim_floor:
            opt.floor_plane = var['floor_plane']
            cam_smpl, _ = opt.smpl_results(opt.trans, opt.root_orient, cur_body_pose, opt.betas)
            opt.cam2prior_R, opt.cam2prior_t, opt.cam2prior_root_height = R.fitting_utils.compute_cam2prior(
                opt.floor_plane, opt.trans[:, 0], opt.root_orient[:, 0], cam_smpl['joints3d'][:, 0])
        rr, cam_rr = opt.rollout_latent_motion(opt.trans, opt.root_orient, cur_body_pose, opt.betas, prior_opt_params,
                                               opt.latent_motion, return_prior=opt.cond_prior)
        cur_latent_pose = opt.pose2latent(rr['pose_body'])
        pred, _ = opt.smpl_results(rr['trans'], rr['root_orient'], rr['pose_body'], opt.betas)
        pred.update(latent_pose=cur_latent_pose, betas=opt.betas, latent_motion=opt.latent_motion, joints_vel=opt.joints_vel,
                    trans_vel=opt.trans_vel, root_orient_vel=opt.root_orient_vel, joints3d_rollout=rr['joints'])
        pred['contacts'], pred['contacts_conf'] = rr['contacts'], rr['contacts_conf']
        cam_pred = pred
        if opt.optim_floor:
            cam_pred, _ = opt.smpl_results(cam_rr['trans'], cam_rr['root_orient'], rr['pose_body'], opt.betas)
            cam_pred.update(latent_pose=cur_latent_pose, betas=opt.betas, floor_plane=opt.floor_plane)
        loss, _ = opt.fitting_loss.motion_fit(obs, pred, cam_pred, T, cond_prior=rr['cond_prior'], init_motion_scale=1.0)
        wrt = ['trans', 'root_orient', 'latent_pose', 'betas', 'latent_motion', 'trans_vel', 'joints_vel', 'root_orient_vel']
        if opt.optim_floor:
            wrt.append('floor_plane')
    grads = torch.autograd.grad(loss, [var[k] for k in wrt], allow_unused=True)
    out = {'loss': loss.detach().numpy()}
    for k, gk in zip(wrt, grads):
        out['g_' + k] = np.zeros_like(var[k].detach().numpy()) if gk is None else gk.numpy()
    return out


def main():
    R = ref_loader.load()
    R.motion_optimizer.Logger.log = staticmethod(lambda *a, **k: None)
    R.fitting_loss.Logger.log = staticmethod(lambda *a, **k: None)
    R.motion_optimizer.log_cur_stats = lambda *a, **k: None
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        npz = synth.write_smplh_npz(os.path.join(td, 'model.npz'), seed=0)
        for kind, B, T in (('amass', 2, 8), ('rgb', 4, 8)):
            case = CC.make_case(kind, B, T, seed=1)
            opt = build_reference(R, kind, B, T, npz)
            save = {'B': B, 'T': T, 'seed': 1}
            for stage in range(3):
                res = eval_stage(R, opt, case, stage)
                for k, v in res.items():
                    save[f's{stage}_{k}'] = v
                print(kind, 'stage', stage, 'loss', float(res['loss']))
            # short trajectory: the reference run() itself
            torch.manual_seed(0)
            opt2 = build_reference(R, kind, B, T, npz)
            obs = {k: v.clone() for k, v in CC.make_case(kind, B, T, seed=2)['obs'].items()}
            num_iter = [2, 2, 2]
            trace = []
            for name in ('root_fit', 'smpl_fit', 'motion_fit'):
                orig = getattr(opt2.fitting_loss, name)

                def wrapped(*a, _orig=orig, _name=name, **k):
                    loss, st = _orig(*a, **k)
                    if _name == ('root_fit', 'smpl_fit', 'motion_fit')[opt2.fitting_loss.cur_stage_idx]:
                        trace.append((opt2.fitting_loss.cur_stage_idx, float(loss)))
                    return loss, st
                setattr(opt2.fitting_loss, name, wrapped)
            final, stages = opt2.run(obs, data_fps=30, lr=1.0, num_iter=num_iter, lbfgs_max_iter=5)
            save['run_trace'] = np.array(trace, dtype=np.float64)
            save['run_num_iter'] = np.array(num_iter)
            for k in ('trans', 'root_orient', 'pose_body', 'betas', 'latent_motion'):
                save['run_' + k] = final[k].detach().numpy()
            save['run_stage2_joints3d'] = stages['stage2']['joints3d'].detach().numpy()
            save['run_stage3_joints3d'] = stages['stage3']['joints3d'].detach().numpy()
            np.savez_compressed(os.path.join(OUT, f'closure_{kind}.npz'), **save)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    main()
