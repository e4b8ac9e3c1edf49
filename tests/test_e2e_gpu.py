"""GPU tier: end-to-end 3-stage fits (MotionOptimizer.run) on synthetic problems with a known ground-truth motion --
the whole pipeline the reference drives from run_fitting.py, at reduced iteration counts."""
import os

import numpy as np
import pytest
import torch

import fitting_checks as FC
from humor_amd import results, synth
from humor_amd.body_model import BodyModel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


def _gt_joints(npz, dev, B, T, seed):
    root, body, trans = synth.smooth_pose_sequence(B, T, seed=seed, amp=0.2)
    betas = 0.5 * torch.randn(B, 16, generator=torch.Generator().manual_seed(seed))
    bm = BodyModel(npz, num_betas=16, use_vtx_selector=False, vertex_subset=[0])
    N = B * T
    out = bm(root_orient=root.reshape(N, 3).to(dev), pose_body=body.reshape(N, 63).to(dev), trans=trans.reshape(N, 3).to(dev),
             betas=betas.unsqueeze(1).expand(B, T, 16).reshape(N, 16).to(dev))
    gt = dict(trans=trans, root_orient=root, pose_body=body, betas=betas)
    return out.Jtr[:, :22].reshape(B, T, 22, 3).detach(), gt


@pytest.mark.parametrize('use_graphs', [False, True])
@pytest.mark.parametrize('stage', [0, 2])
def test_non_finite_objective_aborts_the_fit(gpu_lib, dev, smplh_npz, use_graphs, stage):
    """A fit whose objective turns non-finite raises (run_fitting.py:437-439 then skips the batch) instead of returning NaN results, in
    every stage, eager and under hipGraph replay.  Stage 1: a NaN observation.  Stage 3: the latent motion is overwritten with NaN behind the
    optimiser's back after the stage has started (the injection a failed persistent roll-out team would make)."""
    B, T = 2, 12
    joints, _ = _gt_joints(smplh_npz, dev, B, T, seed=5)
    obs = {'joints3d': joints.clone()}
    opt = FC.build(gpu_lib, dev, 'amass', B, T, smplh_npz)
    opt.use_graphs = use_graphs
    if stage == 0:
        obs['joints3d'][1, 3, 5, 0] = float('nan')
        with pytest.raises(RuntimeError, match='non-finite objective in stage 1'):
            opt.run(obs, data_fps=30, lr=1.0, num_iter=[3, 2, 2], lbfgs_max_iter=5)
        return
    health = opt._check_rollout_health
    calls = []

    def poison_after_two():
        health()
        calls.append(1)
        if len(calls) == 2:
            with torch.no_grad():
                opt.latent_motion[1, 4, 7] = float('nan')
    opt._check_rollout_health = poison_after_two
    with pytest.raises(RuntimeError, match='non-finite objective in stage3'):
        opt.run(obs, data_fps=30, lr=1.0, num_iter=[3, 3, 6], lbfgs_max_iter=5)
    assert 2 <= len(calls) <= 4


@pytest.mark.parametrize('use_graphs', [False, True])
def test_amass_style_fit_reduces_joint_error(gpu_lib, dev, smplh_npz, tmp_path, use_graphs):
    """fit_amass_joints-shaped problem (config C2): noisy 3D joints of a known motion; the three stages must reduce the
    joint error substantially and return the reference's result structure; results are written in its npz layout."""
    B, T = 2, 30
    joints, gt = _gt_joints(smplh_npz, dev, B, T, seed=7)
    g = torch.Generator().manual_seed(1)
    obs = {'joints3d': joints + 0.02 * torch.randn(joints.shape, generator=g).to(dev)}
    opt = FC.build(gpu_lib, dev, 'amass', B, T, smplh_npz)
    opt.use_graphs = use_graphs
    opt.loss_trace = []
    out_dirs = [str(tmp_path / f'seq{b}') for b in range(B)]
    for d in out_dirs:
        os.makedirs(d, exist_ok=True)
    final, stages = opt.run(obs, data_fps=30, lr=1.0, num_iter=[6, 10, 8], lbfgs_max_iter=20, stages_res_out=out_dirs)
    err0 = (joints - 0.0).norm(dim=-1).mean().item()
    err2 = (stages['stage2']['joints3d'] - joints).norm(dim=-1).mean().item()
    err3 = (stages['stage3']['joints3d'] - joints).norm(dim=-1).mean().item()
    assert err2 < 0.5 * err0, (err0, err2)
    # stage 3 adds a motion prior with RANDOM weights (no checkpoint is redistributable): it cannot improve the fit, it must
    # stay finite, keep the error bounded and decrease its own objective
    assert np.isfinite(err3) and err3 < err0, (err0, err3)
    tr = np.array(opt.loss_trace)
    for st in range(3):
        ls = tr[tr[:, 0] == st][:, 1]
        assert ls[-1] < ls[0]                       # every stage made progress
    assert final['trans'].shape == (B, T, 3) and final['pose_body'].shape == (B, T, 63)
    assert final['latent_motion'].shape == (B, T - 1, 48) and final['contacts'].shape == (B, T, 22)
    assert set(torch.unique(final['contacts']).tolist()) <= {0.0, 1.0}
    results.save_optim_result(out_dirs, final, stages, gt_data=gt, observed_data=obs, data_type='AMASS', optim_floor=False)
    d = np.load(os.path.join(out_dirs[0], 'stage3_results.npz'))
    assert d['trans'].shape == (T, 3) and d['pose_body'].shape == (T, 63) and d['betas'].shape == (16,) and d['contacts'].shape == (T, 22)
    assert os.path.exists(os.path.join(out_dirs[1], 'gt_results.npz')) and os.path.exists(os.path.join(out_dirs[1], 'observations.npz'))
    for f in ('stage1_results.npz', 'stage2_results.npz', 'stage3_init_results.npz'):       # files eval_fitting_* / viz_fitting_rgb list
        assert os.path.exists(os.path.join(out_dirs[0], f)), f


def test_rgb_style_fit_runs_all_phases(gpu_lib, dev, smplh_npz, tmp_path):
    """fit_rgb_demo_use_split-shaped problem (config C3/C4 shape): 2D keypoints, floor optimisation, overlapping
    sub-sequences; exercises the tune-init / frozen-init / refine phases of stage 3 and the floor outputs."""
    from oracle import closure_cases as CC
    B, T = 4, 20
    opt = FC.build(gpu_lib, dev, 'rgb', B, T, smplh_npz)
    opt.stage3_tune_init_freeze_start, opt.stage3_tune_init_freeze_end = 2, 4
    opt.use_graphs = True
    opt.loss_trace = []
    obs = {k: v.to(dev) for k, v in CC.make_case('rgb', B, T, seed=4)['obs'].items()}
    obs['seq_interval'] = torch.tensor([[b * (T - 5), b * (T - 5) + T] for b in range(B)])
    out_dirs = [str(tmp_path / f'seq{b}') for b in range(B)]
    for d in out_dirs:
        os.makedirs(d, exist_ok=True)
    final, stages = opt.run(obs, data_fps=30, lr=1.0, num_iter=[3, 4, 6], lbfgs_max_iter=10, stages_res_out=out_dirs)
    # every per-stage file the reference writes and its eval / viz scripts list (motion_optimizer.py:260-270, 422-456, 651-674)
    for f in ('stage1_results.npz', 'stage2_results.npz', 'stage3_init_results.npz', 'stage3_init_results_prior.npz',
              'stage2_results_prior.npz', 'stage3_results.npz'):
        d = np.load(os.path.join(out_dirs[1], f))
        assert d['trans'].shape == (T, 3) and d['root_orient'].shape == (T, 3) and d['pose_body'].shape == (T, 63), f
    assert final['floor_plane'].shape == (B, 4)
    assert 'prior_trans' in stages['stage3'] and 'prior_joints3d_rollout' in stages['stage3']
    assert all(torch.isfinite(v).all() for v in final.values())
    tr = np.array(opt.loss_trace)
    assert (tr[:, 0] == 2).sum() > 10 and np.isfinite(tr[:, 1]).all()


def _sharded_run_worker(rank, world, port, npz, out):
    """One rank of a 2-rank sharded 3-stage fit; both ranks share the test box's single GPU (gloo, host-staged collectives)."""
    import sys
    import torch.distributed as dist
    from conftest import ROOT
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle import closure_cases as CC
    from humor_amd import _lib
    from humor_amd.distributed import Shard
    dev = torch.device('cuda:0')
    B, T = 4, 20
    opt = FC.build(_lib.get_lib(), dev, 'rgb', B, T, npz, shard=Shard(B))
    opt.stage3_tune_init_freeze_start, opt.stage3_tune_init_freeze_end = 2, 4
    obs = {k: v.to(dev) for k, v in CC.make_case('rgb', B, T, seed=4)['obs'].items()}
    obs['seq_interval'] = torch.tensor([[b * (T - 5), b * (T - 5) + T] for b in range(B)])
    final, stages = opt.run(obs, data_fps=30, lr=1.0, num_iter=[3, 4, 5], lbfgs_max_iter=8)
    torch.save({k: v.detach().cpu() for k, v in final.items()}, out % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_run_is_identical_on_all_ranks(gpu_lib, dev, smplh_npz, tmp_path):
    """Replicated L-BFGS, sharded closures: after a whole 3-stage MotionOptimizer.run every rank must hold bit-identical
    results (every line-search decision was taken on all-reduced values), finite, with the reference's result structure."""
    import torch.multiprocessing as mp
    out = str(tmp_path / 'final_rank%d.pt')
    port = 31000 + (os.getpid() % 2000)
    os.environ['HUMOR_AMD_ROLLOUT_PERSIST'] = '0'          # two processes on ONE GPU (see humor_amd/_lib.py)
    try:
        mp.spawn(_sharded_run_worker, args=(2, port, smplh_npz, out), nprocs=2, join=True)
    finally:
        del os.environ['HUMOR_AMD_ROLLOUT_PERSIST']
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    assert set(r0) == set(r1) and 'latent_motion' in r0 and r0['floor_plane'].shape == (4, 4)
    for k in r0:
        assert torch.isfinite(r0[k]).all(), k
        assert torch.equal(r0[k], r1[k]), f'{k} differs between ranks'


def test_stitched_rgb_result_round_trip(gpu_lib, dev, smplh_npz, tmp_path):
    """save_rgb_stitched_result (fitting_utils.py:398-523): a known 50-frame sequence cut into three overlapping sub-sequence
    result directories is stitched back frame for frame; the prior-frame copy puts frame 0's root over the origin, upright."""
    from humor_amd import results, synth
    T, ov = 20, 5
    N = 3 * T - 2 * ov
    root, body, trans = synth.smooth_pose_sequence(1, N, seed=3)
    root, body, trans = root[0] + torch.tensor([np.pi, 0.0, 0.0]), body[0], trans[0] + torch.tensor([0.0, 0.0, 3.0])
    g = torch.Generator().manual_seed(0)
    betas = 0.3 * torch.randn(16, generator=g)
    contacts = (torch.rand(N, 22, generator=g) > 0.5).float()
    j2d = torch.rand(N, 25, 3, generator=g)
    ivals, dirs = [], []
    for b in range(3):
        s = b * (T - ov)
        ivals.append((s, s + T))
        d = tmp_path / f'seq{b}'
        d.mkdir()
        dirs.append(str(d))
        np.savez(str(d / 'stage3_results.npz'), betas=betas.numpy(), trans=trans[s:s + T].numpy(), root_orient=root[s:s + T].numpy(),
                 pose_body=body[s:s + T].numpy(), contacts=contacts[s:s + T].numpy(), floor_plane=np.array([0.0, -1.0, 0.0, -0.9 - 0.01 * b], dtype=np.float32))
        np.savez(str(d / 'gt_results.npz'), cam_mtx=np.eye(3))
        np.savez(str(d / 'observations.npz'), joints2d=j2d[s:s + T].numpy(), img_paths=np.array([f'f{i:04d}.png' for i in range(s, s + T)]))
        (d / 'meta.txt').write_text('gender neutral')
    out = results.save_rgb_stitched_result(ivals, dirs, str(tmp_path), dev, smplh_npz, 16, True)
    r = np.load(os.path.join(out, 'stage3_results.npz'))
    assert r['trans'].shape == (N, 3) and r['betas'].shape == (N, 16) and r['contacts'].shape == (N, 22)
    assert np.array_equal(r['trans'], trans.numpy()) and np.array_equal(r['pose_body'], body.numpy()) and np.array_equal(r['contacts'], contacts.numpy())
    assert np.allclose(r['floor_plane'], [0.0, -1.0, 0.0, -0.9])
    o = np.load(os.path.join(out, 'observations.npz'))
    assert np.array_equal(o['joints2d'], j2d.numpy()) and list(o['img_paths']) == [f'f{i:04d}.png' for i in range(N)]
    assert os.path.exists(os.path.join(out, 'meta.txt')) and os.path.exists(os.path.join(out, 'gt_results.npz'))
    p = np.load(os.path.join(out, 'stage3_results_prior.npz'))
    assert p['trans'].shape == (N, 3) and np.abs(p['trans'][0, :2]).max() < 1e-5 and np.isfinite(p['root_orient']).all()
    # rigid map: pairwise root distances are preserved
    d_cam = np.linalg.norm(trans.numpy()[1:] - trans.numpy()[:-1], axis=1)
    d_pri = np.linalg.norm(p['trans'][1:] - p['trans'][:-1], axis=1)
    assert np.abs(d_cam - d_pri).max() < 1e-4


def test_bench_starts_n_ranks_itself(gpu_lib):
    """`python bench.py --gpus 2` with no launcher in front of it must start two ranks (it used to run one and print
    n_gpus 1).  On the one-GPU test box both ranks share cuda:0 and the collectives go through gloo (host staging): the numbers mean
    nothing, the line's shape does -- n_gpus = 2, a global batch of 64 sub-sequences, the strong-scaling companion, the collectives."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, HUMOR_AMD_BENCH_BACKEND='gloo', HUMOR_AMD_BENCH_ONE_GPU='1', HUMOR_AMD_ROLLOUT_PERSIST='0')
    env.pop('WORLD_SIZE', None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['config']['global_batch'] == 64 and res['scaling'] == 'weak'
    assert res['strong']['global_batch'] == 32 and res['strong']['sequences_per_gpu'] == [16, 16]
    assert res['rccl']['allreduce_us'] > 0 and res['value'] > 0
