"""GPU tier: the fitting objectives (stage 1/2/3 closures) and short MotionOptimizer runs against fixtures produced by
the reference MotionOptimizer."""
import os
import sys

import pytest
import torch

import fitting_checks as FC
from conftest import ROOT
from humor_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


@pytest.mark.parametrize('kind', ['amass', 'rgb'])
def test_stage_objectives_match_reference(gpu_lib, dev, smplh_npz, kind):
    FC.check_objectives(gpu_lib, dev, smplh_npz, kind)


@pytest.mark.parametrize('kind', ['amass', 'rgb'])
def test_short_run_matches_reference(gpu_lib, dev, smplh_npz, kind):
    FC.check_short_run(gpu_lib, dev, smplh_npz, kind)


@pytest.mark.parametrize('name', ['c2', 'c3', 'c4', 'c4_full'])
def test_stage_objectives_at_baseline_sizes(gpu_lib, dev, smplh_npz, name):
    """C2 (2x60, joints3d), C3 (1x90, joints2d + floor), a C4 slice (8x60, overlap 10) and the FULL C4 batch (32x60 = bench.py's
    workload: one full 32-row tile, split-K policy) against the reference MotionOptimizer."""
    print(name, FC.check_objectives_long(gpu_lib, dev, smplh_npz, name))


@pytest.mark.parametrize('name,kind', [('c2', 'amass'), ('c3', 'rgb'), ('c4', 'rgb')])
def test_short_run_at_baseline_sizes(gpu_lib, dev, smplh_npz, name, kind):
    """C2 (2 x 60, joints3d), C3 (1 x 90, joints2d + floor: fit_rgb_demo_no_split) and the C4 slice (8 x 60, overlap 10)."""
    FC.check_short_run(gpu_lib, dev, smplh_npz, kind, long_name=name)


@pytest.mark.parametrize('name,kind', [('c2', 'amass'), ('c3', 'rgb'), ('c4', 'rgb')])
def test_lbfgs_trajectory_equals_torch_lbfgs_on_the_same_closure(gpu_lib, dev, smplh_npz, name, kind):
    """humor_amd.lbfgs.LBFGS against torch.optim.LBFGS, both driving THIS implementation's closures on the fixture problems: the identical
    evaluation sequence, stage by stage (the optimiser's deterministic pin; the closures are pinned by the objective tests)."""
    FC.check_lbfgs_trajectory(gpu_lib, dev, smplh_npz, kind, long_name=name)


@pytest.mark.parametrize('n,h,k', [(94752, 100, 100), (73472, 100, 71), (301, 128, 128), (1000, 100, 3)])
def test_lbfgs_kernels_against_two_loop_recursion(gpu_lib, dev, n, h, k):
    """ha_lbfgs_gram / ha_lbfgs_pair_coeffs / ha_lbfgs_scalars at the stage-3 and stage-2 vector lengths of the C4 fit with a full
    history, against the float64 two-loop recursion of torch.optim.LBFGS."""
    import lbfgs_checks as LC
    print('direction rel. error', LC.check_direction(gpu_lib, dev, n=n, h=h, k=k, seed=k))


@pytest.mark.parametrize('B,T,reps', [(4, 8, 3), (32, 60, 8), (40, 16, 4)])
def test_graphed_closure_equals_eager(gpu_lib, dev, smplh_npz, B, T, reps):
    """hipGraph replay of the stage-3 closure returns the eager loss and gradients, call after call, while every variable changes in
    place between the calls; at the metric's batch the persistent roll-out kernels are inside the graph and must not report a failed
    launch (round 4: with hipMemsetAsync in front of them they started on the previous replay's team counters -- error word 0x100 --
    as soon as the values changed; the exchange space is now cleared by a kernel, csrc/common.h zero_async).  40 x 16: the pipelined
    kernels of rollout_pipe.inc (more than 32 sequences) inside the graph."""
    from oracle import closure_cases as CC
    case = CC.make_case('rgb', B, T, seed=1)
    res = {}
    for graphs in (False, True):
        opt = FC.build(gpu_lib, dev, 'rgb', B, T, smplh_npz)
        opt.use_graphs = graphs
        var = {k: v.clone().to(dev) for k, v in case['var'].items()}
        obs = {k: v.clone().to(dev) for k, v in case['obs'].items()}
        opt.fitting_loss.set_stage(2)
        opt.trans, opt.root_orient, opt.latent_pose = (var[k][:, :1].clone().requires_grad_(True) for k in ('trans', 'root_orient', 'latent_pose'))
        opt.betas = var['betas'].requires_grad_(True)
        opt.latent_motion = var['latent_motion'].requires_grad_(True)
        opt.trans_vel, opt.joints_vel, opt.root_orient_vel = (var[k].requires_grad_(True) for k in ('trans_vel', 'joints_vel', 'root_orient_vel'))
        opt.floor_plane = var['floor_plane'].requires_grad_(True)
        prior = [opt.trans_vel, opt.joints_vel, opt.root_orient_vel]
        params = [opt.trans, opt.root_orient, opt.latent_pose, opt.betas, opt.latent_motion] + prior + [opt.floor_plane]
        ol = opt._local_obs(obs)
        closure = opt.make_closure(lambda: opt._stage3_objective(ol, None, prior, False, 15, 1.0, 200.0, True, 'neutral'), params, None)
        out = []
        for it in range(reps):
            loss = closure()
            out.append((loss.item(), [p.grad.clone() for p in params]))
            with torch.no_grad():
                opt.latent_motion.add_(0.01)          # the optimiser updates variables in place between evaluations
                for p in params:
                    p.mul_(1.0 + 2.0 ** -20)
            torch.cuda.synchronize()
        res[graphs] = out
        available, err, _ = opt.motion_prior.persistent_rollout_status(dev)
        assert err == 0 and available == 1, (graphs, available, hex(err))
    for (l0, g0), (l1, g1) in zip(res[False], res[True]):
        assert abs(l0 - l1) <= 1e-5 * abs(l0)
        for a, b in zip(g0, g1):
            assert (a - b).abs().max().item() <= 2e-4 * max(1.0, a.abs().max().item())


def _sharded_gpu_worker(rank, world, port, npz, out):
    """Two ranks on the ONE GPU of the test box (gloo: device tensors are staged through the host), sharded stage-3 closure."""
    import torch.distributed as dist
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle import closure_cases as CC
    from humor_amd import _lib
    from humor_amd.distributed import Shard, allreduce_loss_and_grads
    dev = torch.device('cuda:0')
    lib = _lib.get_lib()
    B, T = 6, 10
    case = CC.make_case('rgb', B, T, seed=2)
    opt = FC.build(lib, dev, 'rgb', B, T, npz, shard=Shard(B))
    res = FC.eval_stage(opt, case, 2, dev)
    keys = [k for k in res if k != 'loss']
    params = [torch.zeros_like(res[k]).requires_grad_(True) for k in keys]
    for p, k in zip(params, keys):
        p.grad = res[k].clone()
    loss = allreduce_loss_and_grads(res['loss'], params)
    if rank == 0:
        torch.save({'loss': loss.item(), **{k: p.grad.cpu() for k, p in zip(keys, params)}}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_stage3_closure_on_gpu(gpu_lib, dev, smplh_npz, tmp_path):
    """The multi-GPU closure path (bench.py --gpus N, MotionOptimizer(shard=...)) for stage 3 with the real kernels: two
    ranks, three sub-sequences each, loss and every gradient equal the single-process closure."""
    import torch.multiprocessing as mp
    from oracle import closure_cases as CC
    out = str(tmp_path / 'sharded3.pt')
    port = 29500 + (os.getpid() % 2000)
    os.environ['HUMOR_AMD_ROLLOUT_PERSIST'] = '0'          # two processes on ONE GPU (see humor_amd/_lib.py)
    try:
        mp.spawn(_sharded_gpu_worker, args=(2, port, smplh_npz, out), nprocs=2, join=True)
    finally:
        del os.environ['HUMOR_AMD_ROLLOUT_PERSIST']
    sharded = torch.load(out)
    B, T = 6, 10
    case = CC.make_case('rgb', B, T, seed=2)
    opt = FC.build(gpu_lib, dev, 'rgb', B, T, smplh_npz)
    gpu_lib.call('ha_tune_set', b'rollout_persist', 0)      # the same roll-out path as the two workers: this test is about the sharding
    try:
        res = FC.eval_stage(opt, case, 2, dev)
    finally:
        gpu_lib.call('ha_tune_set', b'rollout_persist', 1)
    assert abs(res['loss'].item() - sharded['loss']) <= 1e-5 * abs(res['loss'].item())
    for k, v in res.items():
        if k != 'loss':
            v = v.detach().cpu()
            assert (v - sharded[k]).abs().max().item() <= 2e-4 * max(1.0, v.abs().max().item()), k


def test_fused_fit_loss_equals_term_by_term(gpu_lib, dev):
    """ha_fit_loss against the term-by-term PyTorch evaluation (itself pinned bit-for-bit to the reference FittingLoss in the CPU
    tier): loss, every term, every gradient; C4-like sizes, with and without the multi-GPU halo."""
    import fitloss_checks as FL
    for B, T, seed in ((3, 7, 0), (8, 60, 1), (1, 90, 2)):
        print('fused fit loss', B, T, 'worst rel grad diff', FL.check_fused_vs_terms(gpu_lib, dev, B=B, T=T, seed=seed))


def test_points3d_chamfer_term_through_the_optimizer(gpu_lib, dev, smplh_npz):
    """Point-cloud fitting (fit_proxd-style): MotionOptimizer with the 'points3d' modality evaluates the dense 6890-vertex SMPL, the
    chamfer kernels and the robust weighting.  The loss equals a re-evaluation with the ORACLE's nearest-neighbour indices; the
    gradient w.r.t. the root translation matches central finite differences of the (locally smooth) objective."""
    from humor_amd import synth
    from humor_amd.body_model import BodyModel
    from humor_amd.configs import stage_weights
    from humor_amd.fitting_loss import apply_robust_weighting
    from humor_amd.humor_model import HumorModel
    from humor_amd.motion_optimizer import MotionOptimizer
    from oracle import chamfer_restated as CR
    B, T, NOBS = 2, 3, 700
    bm = BodyModel(smplh_npz, num_betas=16, batch_size=B * T, use_vtx_selector=False)
    hm = HumorModel(in_rot_rep='mat', out_rot_rep='aa', latent_size=48, model_data_config='smpl+joints+contacts', steps_in=1)
    hm.load_state_dict(synth.humor_state_dict(seed=0))
    hm = hm.to(dev).eval()
    w, mu, cov = synth.make_gmm(seed=0)
    weights = stage_weights([{'points3d': 1.0}] * 3)
    opt = MotionOptimizer(dev, bm, 16, B, T, ['points3d'], weights, synth.SynthVPoser(seed=0).to(dev).eval(), hm,
                          {'gmm': (w.to(dev), mu.to(dev), cov.to(dev))}, robust_loss_type='bisquare', use_chamfer=True)
    g = torch.Generator().manual_seed(0)
    opt.trans = (0.1 * torch.randn(B, T, 3, generator=g)).to(dev).requires_grad_(True)
    opt.root_orient = (torch.tensor([0.3, 0.1, -0.2]) + 0.1 * torch.randn(B, T, 3, generator=g)).to(dev).requires_grad_(True)
    with torch.no_grad():
        pred, _ = opt.smpl_results(opt.trans, opt.root_orient, opt.latent2pose(opt.latent_pose), opt.betas)
    assert pred['points3d'].shape == (B, T, 6890, 3) and pred['verts3d'].shape == (B, T, 43, 3)
    pick = torch.randint(0, 6890, (B, T, NOBS), generator=g).to(dev)
    cloud = torch.gather(pred['points3d'], 2, pick.unsqueeze(-1).expand(B, T, NOBS, 3)) + 0.02 * torch.randn(B, T, NOBS, 3, generator=g).to(dev)
    obs = {'points3d': cloud}
    opt.fitting_loss.set_stage(0)
    loss, stats = opt._stage1_objective(opt._local_obs(obs), False)
    # re-evaluation with the oracle's indices
    i1 = CR.nnsearch(cloud.reshape(B * T, NOBS, 3).cpu().numpy(), pred['points3d'].reshape(B * T, 6890, 3).cpu().numpy())[1]
    near = torch.gather(pred['points3d'].reshape(B * T, 6890, 3), 1, torch.from_numpy(i1).long().to(dev).unsqueeze(-1).expand(B * T, NOBS, 3))
    d = ((cloud.reshape(B * T, NOBS, 3) - near) ** 2).sum(-1).reshape(B, T * NOBS)
    ref = 0.5 * apply_robust_weighting(d.sqrt(), 'bisquare', 4.6851)[0].sum()
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item()), (loss.item(), ref.item())
    # gradient: with the robust weights switched off the objective is a smooth function of the variables (the bisquare weights are
    # deliberately detached, fitting_utils.py:202, so with them the "gradient" is not the derivative of the loss value)
    opt.fitting_loss.robust_loss = 'none'
    loss, stats = opt._stage1_objective(opt._local_obs(obs), False)
    gt, = torch.autograd.grad(loss, [opt.trans])
    eps = 1e-3
    for (b, t, c) in ((0, 0, 0), (1, 2, 1), (0, 1, 2)):
        vals = []
        for sgn in (1.0, -1.0):
            with torch.no_grad():
                opt.trans[b, t, c] += sgn * eps
                vals.append(opt._stage1_objective(opt._local_obs(obs), False)[0].item())
                opt.trans[b, t, c] -= sgn * eps
        fd = (vals[0] - vals[1]) / (2 * eps)
        assert abs(fd - gt[b, t, c].item()) <= 0.05 * max(1.0, abs(fd)), (b, t, c, fd, gt[b, t, c].item())


def test_rollout_post_kernel_equals_op_chain(gpu_lib, dev, smplh_npz):
    """Fused roll-out post-processing against the op chain it replaces, stand-alone (C4-like sizes) and through the stage-3 objective."""
    import fitloss_checks as FL
    from oracle import closure_cases as CC
    for B, S, cam in ((3, 6, True), (8, 59, True), (2, 89, False)):
        print('rollout post', B, S, cam, FL.check_rollout_post(gpu_lib, dev, B=B, S=S, seed=B, cam=cam))
    B, T = 4, 12
    case = CC.make_case('rgb', B, T, seed=5)
    res = []
    for fused in (True, False):
        opt = FC.build(gpu_lib, dev, 'rgb', B, T, smplh_npz)
        opt.fused_post = fused
        res.append(FC.eval_stage(opt, case, 2, dev))
    assert abs(res[0]['loss'].item() - res[1]['loss'].item()) <= 1e-6 * abs(res[1]['loss'].item())
    for k in res[0]:
        if k != 'loss':
            e = (res[0][k] - res[1][k]).abs().max().item() / max(1.0, res[1][k].abs().max().item())
            assert e < 1e-4, (k, e)


def test_rigid_image_equals_second_smpl_evaluation(gpu_lib, dev, smplh_npz):
    """The camera-frame body as the rigid image of the prior-frame one (ha_rigid_image_*) against a second SMPL evaluation: stand-alone
    (key-vertex subset and every vertex) and through the stage-3 objective."""
    import fitloss_checks as FL
    from oracle import closure_cases as CC
    for N, dense in ((7, False), (1920, False), (64, True)):
        print('rigid image', N, dense, FL.check_rigid_image(gpu_lib, dev, smplh_npz, N=N, seed=N, dense=dense))
    B, T = 4, 12
    case = CC.make_case('rgb', B, T, seed=7)
    res = []
    for rigid in (True, False):
        opt = FC.build(gpu_lib, dev, 'rgb', B, T, smplh_npz)
        opt.rigid_cam_body = rigid
        res.append(FC.eval_stage(opt, case, 2, dev))
    assert abs(res[0]['loss'].item() - res[1]['loss'].item()) <= 1e-5 * abs(res[1]['loss'].item())
    for k in res[0]:
        if k != 'loss':
            e = (res[0][k] - res[1][k]).abs().max().item() / max(1.0, res[1][k].abs().max().item())
            assert e < 2e-4, (k, e)


def test_gmm_nll_kernel(gpu_lib, dev):
    """ha_gmm_nll against the op-by-op mixture log-density and its autograd (32 sequences as in the closure, strided frame-0 rows, B = 1)."""
    import fitloss_checks as FL
    print('gmm nll: worst relative gradient difference', FL.check_gmm_nll(gpu_lib, dev, B=32, seed=2))


def test_fit_pre_kernel_equals_op_chain(gpu_lib, dev, smplh_npz):
    """Fused stage-3 set-up against the op chain it replaces, stand-alone and through the stage-3 objective."""
    import fitloss_checks as FL
    from oracle import closure_cases as CC
    for B, seed in ((3, 0), (32, 1)):
        print('fit pre', B, FL.check_fit_pre(gpu_lib, dev, smplh_npz, B=B, seed=seed))
    B, T = 4, 12
    case = CC.make_case('rgb', B, T, seed=6)
    res = []
    for fused in (True, False):
        opt = FC.build(gpu_lib, dev, 'rgb', B, T, smplh_npz)
        opt.fused_pre = fused
        res.append(FC.eval_stage(opt, case, 2, dev))
    assert abs(res[0]['loss'].item() - res[1]['loss'].item()) <= 1e-5 * abs(res[1]['loss'].item())
    for k in res[0]:
        if k != 'loss':
            e = (res[0][k] - res[1][k]).abs().max().item() / max(1.0, res[1][k].abs().max().item())
            assert e < 2e-4, (k, e)


@pytest.mark.parametrize('B,T', [(4, 12), (32, 60)])
def test_stage3_nodes_equal_separate_functions(gpu_lib, dev, smplh_npz, B, T):
    """The stage-3 objective as three composite autograd nodes with in-kernel gradient addends (humor_amd/stage3.py, the init-state GMM
    term folded into the fused loss) against the same objective built from the separate Functions + autograd's accumulation launches:
    same kernels, same arithmetic, only the order in which gradient contributions are summed differs (fp32 rounding)."""
    from oracle import closure_cases as CC
    case = CC.make_case('rgb', B, T, seed=13)
    res = []
    for nodes in (True, False):
        opt = FC.build(gpu_lib, dev, 'rgb', B, T, smplh_npz)
        opt.fused_stage3 = nodes
        opt.fitting_loss.fold_init_prior = nodes
        assert (opt._stage3_nodes_config(torch.zeros(1, device=dev)) is not None) == nodes
        res.append(FC.eval_stage(opt, case, 2, dev))
    assert abs(res[0]['loss'].item() - res[1]['loss'].item()) <= 1e-6 * abs(res[1]['loss'].item())
    for k in res[0]:
        if k != 'loss':
            e = (res[0][k] - res[1][k]).abs().max().item() / max(1.0, res[1][k].abs().max().item())
            assert e < 2e-5, (k, e)


def test_stage3_nodes_with_frozen_initial_state(gpu_lib, dev, smplh_npz):
    """The frozen-init phase of stage 3 (motion_optimizer.py:451-458: the initial state does not require gradients, init_motion_scale 4):
    composite nodes against the separate Functions, gradients of the variables that phase optimises."""
    from oracle import closure_cases as CC
    B, T = 8, 20
    case = CC.make_case('rgb', B, T, seed=3)
    out = []
    for nodes in (True, False):
        opt = FC.build(gpu_lib, dev, 'rgb', B, T, smplh_npz)
        opt.fused_stage3 = nodes
        opt.fitting_loss.fold_init_prior = nodes
        var = {k: v.clone().to(dev) for k, v in case['var'].items()}
        obs = {k: v.clone().to(dev) for k, v in case['obs'].items()}
        opt.fitting_loss.set_stage(2)
        opt.trans, opt.root_orient, opt.latent_pose = var['trans'][:, :1].clone(), var['root_orient'][:, :1].clone(), var['latent_pose'][:, :1].clone()
        opt.betas, opt.latent_motion = var['betas'].requires_grad_(True), var['latent_motion'].requires_grad_(True)
        opt.trans_vel, opt.joints_vel, opt.root_orient_vel = var['trans_vel'], var['joints_vel'], var['root_orient_vel']
        opt.floor_plane = var['floor_plane'].requires_grad_(True)
        loss, _ = opt._stage3_objective(opt._local_obs(obs), None, [opt.trans_vel, opt.joints_vel, opt.root_orient_vel], False, 15, 4.0,
                                        opt.fitting_loss.loss_weights['rgb_overlap_consist'], True, 'neutral')
        out.append((loss.item(), torch.autograd.grad(loss, [opt.latent_motion, opt.betas, opt.floor_plane])))
    assert abs(out[0][0] - out[1][0]) <= 1e-6 * abs(out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item())


def test_backward_addends(gpu_lib, dev):
    """In-kernel gradient addends of ha_rigid_image_backward / ha_fit_pre_backward and the strided jcam read (ABI 2)."""
    import fitloss_checks as FL
    print('addends: worst relative difference', FL.check_backward_addends(gpu_lib, dev, B=32, seed=2))


def test_fit_after_an_earlier_persistent_failure_completes_on_the_launch_chain(gpu_lib, dev, smplh_npz):
    """the persistent roll-out's error word is sticky for the lifetime of the network handle, and run_fitting.py makes a
    new MotionOptimizer per batch around ONE HumorModel.  A fit that starts after a failure has already been reported (its evaluations run
    on the launch chain: valid) must complete -- only a failure DURING a fit aborts that fit."""
    from humor_amd import _lib
    from oracle import closure_cases as CC
    B, T = 4, 8
    opt = FC.build(gpu_lib, dev, 'rgb', B, T, smplh_npz)
    hm = opt.motion_prior
    g = torch.Generator().manual_seed(5)
    import rollout_checks as RC
    past, z = RC.canonical_state(B, g).to(dev), torch.randn(B, 3, 48, generator=g).to(dev)
    with torch.no_grad():
        hm.roll_out(past, None, 3, z_seq=z)
        if hm.persistent_rollout_status(dev)[0] != 1:
            pytest.skip('the persistent roll-out is not available on this device')
        gpu_lib.call('ha_tune_set', b'rollout_persist_inject', 1)
        try:
            hm.roll_out(past, None, 3, z_seq=z)
            torch.cuda.synchronize()
        finally:
            gpu_lib.call('ha_tune_set', b'rollout_persist_inject', 0)
        with pytest.raises(_lib.HumorAmdError):
            hm.roll_out(past, None, 3, z_seq=z)                 # the failure is reported once
    av, err, _ = hm.persistent_rollout_status(dev)
    assert av == 0 and err != 0
    obs = {k: v.clone().to(dev) for k, v in CC.make_case('rgb', B, T, seed=2)['obs'].items()}
    for _ in range(2):                                           # two "batches" around the same HumorModel
        opt2 = FC.build(gpu_lib, dev, 'rgb', B, T, smplh_npz, hm=hm)
        final, _stages = opt2.run(obs, data_fps=30, lr=1.0, num_iter=[1, 1, 2], lbfgs_max_iter=3)
        assert all(torch.isfinite(v).all() for v in final.values())


def test_failure_inside_a_replayed_graph_aborts_that_fit_only(gpu_lib, dev, smplh_npz):
    """a persistent launch that fails inside a REPLAYED hipGraph never passes an entry point; MotionOptimizer finds the
    error word after the outer iteration (_check_rollout_health), aborts THAT fit and acknowledges the failure (ha_humor_persist_ack), so
    the next fit around the same HumorModel is not handed the failure again by its first entry point: it completes on the launch chain.
    Injection: the failure knob is set only while the stage-3 closure is being captured, so the captured persistent forward drops a CU in
    every replay while every eager evaluation is healthy."""
    from oracle import closure_cases as CC
    B, T = 4, 8
    opt = FC.build(gpu_lib, dev, 'rgb', B, T, smplh_npz)
    hm = opt.motion_prior
    obs = {k: v.clone().to(dev) for k, v in CC.make_case('rgb', B, T, seed=2)['obs'].items()}
    opt.use_graphs = True
    inner = opt._stage3_objective

    def objective(*a, **k):
        capturing = torch.cuda.is_current_stream_capturing()
        gpu_lib.call('ha_tune_set', b'rollout_persist_inject', 1 if capturing else 0)
        try:
            return inner(*a, **k)
        finally:
            gpu_lib.call('ha_tune_set', b'rollout_persist_inject', 0)
    opt._stage3_objective = objective
    try:
        with pytest.raises(RuntimeError, match='persistent roll-out reported an incomplete launch|non-finite objective'):
            opt.run(obs, data_fps=30, lr=1.0, num_iter=[1, 1, 3], lbfgs_max_iter=3)
    finally:
        gpu_lib.call('ha_tune_set', b'rollout_persist_inject', 0)
    if getattr(opt, 'graph_failures', 0):
        pytest.skip('hipGraph capture of the closure is not available on this box')
    av, err, _ = hm.persistent_rollout_status(dev)
    assert av == 0 and err != 0, (av, hex(err))
    for _ in range(2):                                           # the next "batches" around the same HumorModel complete (launch chain)
        opt2 = FC.build(gpu_lib, dev, 'rgb', B, T, smplh_npz, hm=hm)
        final, _stages = opt2.run(obs, data_fps=30, lr=1.0, num_iter=[1, 1, 2], lbfgs_max_iter=3)
        assert all(torch.isfinite(v).all() for v in final.values())


def test_stage3_closure_beyond_32_sequences_pipelined_equals_launch_chain(gpu_lib, dev, smplh_npz):
    """A stage-3 objective of 40 overlapping sub-sequences (roll-out on the pipelined persistent kernels, rollout_pipe.inc) against the same
    objective with the roll-out on the launch chain (ha_tune_set "rollout_persist" 0): loss and every gradient."""
    from oracle import closure_cases as CC
    B, T = 40, 14
    case = CC.make_case('rgb', B, T, seed=21)
    res = {}
    try:
        for knob in (0, 1):
            gpu_lib.call('ha_tune_set', b'rollout_persist', knob)
            opt = FC.build(gpu_lib, dev, 'rgb', B, T, smplh_npz, state_dict=synth.contractive_state_dict(0))
            res[knob] = {k: v.detach().clone() for k, v in FC.eval_stage(opt, case, 2, dev).items()}
            if knob == 1:
                av, err, n = opt.motion_prior.persistent_rollout_status(dev)
                assert av == 1 and err == 0 and (n & 0xffffffff) >= 1 and (n >> 32) >= 1, (av, hex(err), n)
    finally:
        gpu_lib.call('ha_tune_set', b'rollout_persist', 1)
    assert abs(res[0]['loss'].item() - res[1]['loss'].item()) <= 1e-5 * abs(res[0]['loss'].item())
    for k in res[0]:
        if k != 'loss':
            e = (res[0][k] - res[1][k]).abs().reshape(res[0][k].shape[0], -1).amax(1) / max(1.0, res[0][k].abs().max().item())
            # (per sub-sequence; a ReLU kink within fp32 rounding moves one sequence's gradient by a per cent between two correct evaluations)
            assert (e > 1e-3).sum().item() <= 2 and e.max().item() <= 5e-2, (k, e.max().item(), int((e > 1e-3).sum()))
