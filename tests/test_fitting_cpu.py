"""CPU tier: host-side fitting logic.
  * FittingLoss against the reference's FittingLoss (build container only);
  * stage-1/2 objectives of MotionOptimizer on the SIMT-emulator build vs reference-generated fixtures;
  * the sharded (2-rank, gloo) closure == the single-process closure (loss and gradient), incl. the overlap halo."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fitting_checks as FC
from conftest import ROOT, golden
from humor_amd import synth, tables
from humor_amd.fitting_loss import FittingLoss
from oracle import closure_cases as CC
from oracle import ref_loader

CPU = torch.device('cpu')


@pytest.mark.skipif(not ref_loader.available(), reason='reference tree only exists in the build container')
def test_fitting_loss_matches_reference():
    R = ref_loader.load()
    R.fitting_loss.Logger.log = staticmethod(lambda *a, **k: None)
    torch.manual_seed(0)
    B, T = 4, 12
    stages = CC._weights([
        {'joints2d': 0.001, 'joints3d': 1.0, 'verts3d': 0.7, 'rgb_overlap_consist': 200.0},
        {'joints2d': 0.001, 'joints3d': 1.0, 'verts3d': 0.5, 'pose_prior': 0.04, 'shape_prior': 0.05, 'joints3d_smooth': 100., 'rgb_overlap_consist': 200.0},
        {'joints2d': 0.001, 'joints3d': 0.3, 'joints3d_rollout': 0.2, 'shape_prior': 0.05, 'motion_prior': 0.075, 'init_motion_prior': 0.075,
         'joint_consistency': 100., 'bone_length': 2000., 'contact_vel': 100., 'contact_height': 10., 'floor_reg': 0.167,
         'rgb_overlap_consist': 200.0, 'pose_prior': 0.04}])
    args = dict(init_motion_prior={'gmm': synth.make_gmm()}, smpl2op_map=tables.SMPLH_TO_OPENPOSE25,
                ignore_op_joints=tables.OP_IGNORE_JOINTS, cam_f=torch.tensor([[1060., 1060.]]).expand(B, 2),
                cam_cent=torch.tensor([[951., 536.]]).expand(B, 2), robust_loss='bisquare', joints2d_sigma=100)
    ref, our = R.fitting_loss.FittingLoss(stages, **args), FittingLoss(stages, **args)
    obs = {'joints3d': torch.randn(B, T, 22, 3), 'verts3d': torch.randn(B, T, 43, 3),
           'joints2d': torch.cat([torch.rand(B, T, 25, 2) * 1000, torch.rand(B, T, 25, 1)], 3),
           'floor_plane': torch.tensor([[0., -1., 0., -0.5]]).expand(B, 4).clone(),
           'seq_interval': torch.tensor([[0, 12], [9, 21], [18, 30], [27, 39]])}
    obs['joints3d'][0, 3, 5] = float('inf')
    obs['verts3d'][1, :4] = float('inf')
    obs['prev_batch_overlap_res'] = {'seq_interval': torch.tensor([-9, 3]), 'verts3d': torch.randn(T, 43, 3), 'betas': torch.randn(16),
                                     'floor_plane': torch.randn(4)}
    off = torch.tensor([0, 0, 5.])
    rg = lambda *s: torch.randn(*s).requires_grad_(True)
    pred = {'joints3d': (torch.randn(B, T, 22, 3) + off).requires_grad_(True), 'joints3d_extra': (torch.randn(B, T, 51, 3) + off).requires_grad_(True),
            'verts3d': rg(B, T, 43, 3), 'latent_pose': rg(B, T, 32), 'betas': rg(B, 16), 'latent_motion': rg(B, T - 1, 48),
            'joints_vel': rg(B, 1, 22, 3), 'trans_vel': rg(B, 1, 3), 'root_orient_vel': rg(B, 1, 3), 'joints3d_rollout': rg(B, T, 22, 3),
            'contacts_conf': torch.rand(B, T, 22).requires_grad_(True), 'floor_plane': rg(B, 3)}
    cond = (rg(B, T - 1, 48), (torch.rand(B, T - 1, 48) + 0.5).requires_grad_(True))
    cp = lambda d: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in d.items()}
    for st in range(3):
        ref.set_stage(st)
        our.set_stage(st)
        if st == 0:
            l1, s1 = ref.root_fit(cp(obs), pred)
            l2, s2 = our.root_fit(cp(obs), pred)
        elif st == 1:
            l1, s1 = ref.smpl_fit(cp(obs), pred, T)
            l2, s2 = our.smpl_fit(cp(obs), pred, T)
        else:
            l1, s1 = ref.motion_fit(cp(obs), pred, pred, T, cond_prior=cond, init_motion_scale=4.0)
            l2, s2 = our.motion_fit(cp(obs), pred, pred, T, cond_prior=cond, init_motion_scale=4.0)
        assert sorted(s1.keys()) == sorted(s2.keys())
        assert abs(l1.item() - l2.item()) <= 1e-6 * abs(l1.item())
        ps = list(pred.values()) + list(cond)
        g1 = torch.autograd.grad(l1, ps, allow_unused=True)
        g2 = torch.autograd.grad(l2, ps, allow_unused=True)
        for a, b in zip(g1, g2):
            assert (a is None) == (b is None)
            if a is not None:
                assert (a - b).abs().max() <= 1e-5 * max(1.0, a.abs().max().item())


@pytest.mark.parametrize('kind', ['amass', 'rgb'])
def test_emu_stage12_objectives(emu_lib, smplh_npz, kind):
    gd = golden(f'closure_{kind}.npz')
    B, T = int(gd['B']), int(gd['T'])
    case = CC.make_case(kind, B, T, seed=int(gd['seed']))
    opt = FC.build(emu_lib, CPU, kind, B, T, smplh_npz)
    for stage in (0, 1):
        res = FC.eval_stage(opt, case, stage, CPU)
        ref_loss = float(gd[f's{stage}_loss'])
        assert abs(res['loss'].item() - ref_loss) <= 1e-5 * abs(ref_loss)
        for k, v in res.items():
            if k != 'loss':
                ref = gd[f's{stage}_{k}']
                assert np.abs(v.detach().numpy() - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), (kind, stage, k)


def test_emu_non_finite_objective_aborts_the_fit(emu_lib, smplh_npz):
    """MotionOptimizer.run raises when the objective L-BFGS has just read on the host is non-finite (here: a NaN observation, stage 1) instead of
    carrying NaN into its results; LBFGS.step returns after the ONE evaluation that showed it (the GPU tier repeats this in stage 3 and under
    hipGraph replay: tests/test_e2e_gpu.py::test_non_finite_objective_aborts_the_fit)."""
    B, T = 2, 4
    case = CC.make_case('amass', B, T, seed=3)
    opt = FC.build(emu_lib, CPU, 'amass', B, T, smplh_npz)
    obs = {k: v.clone() for k, v in case['obs'].items()}
    obs['joints3d'][1, 2, 5, 0] = float('nan')
    with pytest.raises(RuntimeError, match='non-finite objective in stage 1'):
        opt.run(obs, data_fps=30, lr=1.0, num_iter=[2, 1, 1], lbfgs_max_iter=3)
    assert opt.closure_evals == 1


def _sharded_worker(rank, world, port, npz, emu_path, out, B=4):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from humor_amd import _lib
    from humor_amd.distributed import Shard, allreduce_loss_and_grads
    lib = _lib.load(emu_path, emulator=True)
    T = 8
    case = CC.make_case('rgb', B, T, seed=1)
    opt = FC.build(lib, CPU, 'rgb', B, T, npz, shard=Shard(B))
    results = {}
    for stage in (0, 1):
        res = FC.eval_stage(opt, case, stage, CPU)
        keys = [k for k in res if k != 'loss']
        params = [torch.zeros_like(res[k]).requires_grad_(True) for k in keys]
        for p, k in zip(params, keys):
            p.grad = res[k].clone()
        loss = allreduce_loss_and_grads(res['loss'], params)
        results[stage] = {'loss': loss.item(), **{k: p.grad.clone() for k, p in zip(keys, params)}}
    if rank == 0:
        torch.save(results, out)
    dist.barrier()
    dist.destroy_process_group()


# (the default CPU tier keeps the 2-rank case; three ranks / ragged shards run with HUMOR_AMD_SLOW=1 -- about a minute each on the emulator)
@pytest.mark.parametrize('world,B', [(2, 4), pytest.param(3, 4, marks=pytest.mark.slow), pytest.param(2, 6, marks=pytest.mark.slow)])
def test_sharded_closure_equals_single_process(emu_lib, smplh_npz, tmp_path, world, B):
    """world_size-2/3 gloo: replicated variables, closure on the local slice, packed all-reduce, forward-only halo (option B: both
    neighbours evaluate the boundary pair; with 3 ranks over 4 sequences the middle ranks own ONE sequence that is both the
    first and the last of its shard; 6 sequences over 2 ranks = exactly THREE per rank, the batch size the reference's dim-less
    torch.cross mishandles, SURVEY G1 -- nothing on this path may depend on the local batch not being 3)."""
    out = str(tmp_path / 'sharded.pt')
    port = 29500 + (os.getpid() % 2000) + world + 7 * B
    mp.spawn(_sharded_worker, args=(world, port, smplh_npz, emu_lib.path, out, B), nprocs=world, join=True)
    sharded = torch.load(out)
    T = 8
    case = CC.make_case('rgb', B, T, seed=1)
    opt = FC.build(emu_lib, CPU, 'rgb', B, T, smplh_npz)
    for stage in (0, 1):
        res = FC.eval_stage(opt, case, stage, CPU)
        assert abs(res['loss'].item() - sharded[stage]['loss']) <= 1e-5 * abs(res['loss'].item())
        for k, v in res.items():
            if k != 'loss':
                assert (v - sharded[stage][k]).abs().max().item() <= 1e-4 * max(1.0, v.abs().max().item()), (stage, k)


def test_emu_fused_fit_loss_equals_term_by_term(emu_lib):
    """ha_fit_loss (one kernel: all terms + gradients) on the SIMT-emulator build against the term-by-term PyTorch FittingLoss that
    test_fitting_loss_matches_reference pins to the reference: root / smpl / motion objectives, with and without the halo."""
    import fitloss_checks as FL
    worst = FL.check_fused_vs_terms(emu_lib, CPU, B=3, T=5)
    print('fused fit loss: worst relative gradient difference', worst)


@pytest.mark.skipif(not ref_loader.available(), reason='reference tree only exists in the build container')
def test_robust_weighting_matches_reference():
    """apply_robust_weighting / bisquare weights / MAD std of the points3d term against fitting_utils.py:192-249, bit for bit."""
    from humor_amd import fitting_loss as FLM
    R = ref_loader.load()
    g = torch.Generator().manual_seed(0)
    res = torch.rand(3, 500, generator=g) * 0.2
    res[1, :7] = 5.0                     # outliers
    for kind in ('bisquare', 'none'):
        a, wa = FLM.apply_robust_weighting(res.clone().requires_grad_(True), kind, 4.6851)
        b, wb = R.fitting_utils.apply_robust_weighting(res.clone().requires_grad_(True), kind, 4.6851)
        assert torch.equal(a, b) and torch.equal(wa, wb)
    assert torch.equal(FLM.robust_std(res), R.fitting_utils.robust_std(res))


def test_emu_rollout_post_kernel_equals_op_chain(emu_lib):
    """ha_rollout_post_forward / _backward on the SIMT emulator against the op-by-op chain of rollout_latent_motion."""
    import fitloss_checks as FL
    for cam in (True, False):
        print('rollout post (cam=%s): worst relative gradient difference' % cam, FL.check_rollout_post(emu_lib, CPU, B=2, S=4, seed=3, cam=cam))


def test_emu_fit_pre_kernel_equals_op_chain(emu_lib, smplh_npz):
    """ha_fit_pre on the SIMT emulator against compute_cam2prior + apply_cam2prior + the initial-state assembly with three SMPL calls."""
    import fitloss_checks as FL
    print('fit pre: worst relative gradient difference', FL.check_fit_pre(emu_lib, CPU, smplh_npz, B=2, seed=1))


def test_emu_rigid_image_equals_second_smpl_evaluation(emu_lib, smplh_npz):
    """ha_rigid_image on the SIMT emulator against a second body-model evaluation under the second root pose."""
    import fitloss_checks as FL
    print('rigid image: worst relative gradient difference', FL.check_rigid_image(emu_lib, CPU, smplh_npz, N=3, seed=2))


@pytest.mark.slow          # (two stage-3 evaluations on the emulator: ~15 minutes; the GPU tier runs the same check at 4 x 12 and 32 x 60)
def test_emu_stage3_nodes_equal_separate_functions(emu_lib, smplh_npz):
    """The stage-3 objective as composite autograd nodes (humor_amd/stage3.py) against the separate Functions on the emulator, 1 x 3: same
    kernels in the same order, so at this size even the gradient sums agree to the last bit."""
    case = CC.make_case('rgb', 1, 3, seed=3)
    res = []
    for nodes in (True, False):
        opt = FC.build(emu_lib, CPU, 'rgb', 1, 3, smplh_npz)
        opt.fused_stage3 = nodes
        opt.fitting_loss.fold_init_prior = nodes
        res.append(FC.eval_stage(opt, case, 2, CPU))
    for k in res[0]:
        assert (res[0][k] - res[1][k]).abs().max().item() <= 1e-6 * max(1.0, res[1][k].abs().max().item()), k


def test_emu_backward_addends(emu_lib):
    """In-kernel gradient addends of ha_rigid_image_backward / ha_fit_pre_backward and the strided jcam read (ABI 2; the stage-3 composite
    nodes) on the emulator."""
    import fitloss_checks as FL
    print('addends: worst relative difference', FL.check_backward_addends(emu_lib, CPU, B=2, seed=1))


def test_emu_gmm_nll_kernel(emu_lib):
    """ha_gmm_nll on the SIMT emulator against the op-by-op mixture log-density (strided frame-0 rows, B = 1)."""
    import fitloss_checks as FL
    print('gmm nll: worst relative gradient difference', FL.check_gmm_nll(emu_lib, CPU, B=3, seed=1))


def test_emu_lbfgs_kernels_long_history(emu_lib):
    """Gram pass, pair installation + coefficient kernel and the scalars kernel on the SIMT emulator with 90 stored pairs in rotated
    slots (lanes own two rows of the recurrences) against the float64 two-loop recursion."""
    import lbfgs_checks as LC
    print('direction rel. error', LC.check_direction(emu_lib, CPU, n=300, h=100, k=90))
    print('direction rel. error (k = 128)', LC.check_direction(emu_lib, CPU, n=200, h=128, k=128, seed=1))


# (speculation is opt-in and off by default: its two tests run with HUMOR_AMD_SLOW=1)
@pytest.mark.parametrize('speculate', [pytest.param(True, marks=pytest.mark.slow), False])
def test_emu_fused_lbfgs_follows_torch_lbfgs(emu_lib, speculate):
    """humor_amd.lbfgs.LBFGS (flat buffer, one Gram pass + coefficient-form two-loop recursion + one GEMV per direction, every scalar
    of an evaluation in one read; the next inner iteration issued speculatively before that read, or -- speculate=False -- one iteration
    at a time) against torch.optim.LBFGS on a smooth non-quadratic problem with a short history (pairs get evicted): the same number of
    closure evaluations in every step() call (state carried over; speculative evaluations that were rolled back are reported through
    the closure's `discard_last` and do not count), loss trace equal to fp32 rounding.  The problem is sized so that three steps stay
    clear of the fp32 noise floor -- at the floor the stopping tests flip on the last bit and the evaluation counts of ANY two
    implementations drift apart."""
    from humor_amd.lbfgs import LBFGS
    torch.manual_seed(0)
    n = 120
    Q, _ = torch.linalg.qr(torch.randn(n, n))
    A = Q @ torch.diag(torch.logspace(0, 1.7, n)) @ Q.t()
    b = torch.randn(n)

    def f(ps):
        x = torch.cat(ps)
        return 0.5 * x @ A @ x - b @ x + 0.1 * torch.sum(torch.cos(3 * x)) + 0.05 * (x ** 4).sum()
    res = {}
    for name in ('torch', 'ours'):
        ps = [torch.zeros(70, requires_grad=True), torch.zeros(50, requires_grad=True)]
        kw = dict(max_iter=8, lr=1.0, line_search_fn='strong_wolfe', history_size=7)
        opt = torch.optim.LBFGS(ps, **kw) if name == 'torch' else LBFGS(ps, _lib_override=emu_lib, **kw)
        if name == 'ours':
            opt.speculate = speculate
        trace, counts = [], []

        def closure():
            for p in ps:
                p.grad = None
            l = f(ps)
            l.backward()
            trace.append(l.item())
            return l
        closure.discard_last = lambda: trace.pop()
        for _ in range(3):
            opt.step(closure)
            counts.append(len(trace))
        res[name] = (trace, counts, torch.cat([p.detach() for p in ps]))
        if name == 'ours':
            print('speculative iterations issued / rolled back:', opt.spec_stats)
            assert (opt.spec_stats['issued'] > 0) == speculate
    (t0, c0, x0), (t1, c1, x1) = res['torch'], res['ours']
    assert c0 == c1, (c0, c1)
    assert max(abs(a - c) / max(1.0, abs(a)) for a, c in zip(t0, t1)) < 1e-5
    assert (x0 - x1).abs().max().item() < 1e-3 and t1[-1] < t1[0] - 1.0


@pytest.mark.slow
def test_emu_speculative_lbfgs_is_bit_identical_to_sequential(emu_lib):
    """The speculative issue of the next inner iteration changes WHEN kernels are issued, never what they compute: on a problem whose line
    search regularly needs more than one trial (so speculative iterations ARE rolled back: pair dropped, retired pair restored, gradient
    row reset, x rewritten) iterates, evaluation counts and the final history order equal the one-iteration-at-a-time run bit for bit."""
    from humor_amd.lbfgs import LBFGS
    torch.manual_seed(3)
    n = 90
    A = torch.randn(n, n)
    A = A @ A.t() / n + 0.05 * torch.eye(n)
    b = torch.randn(n)

    def f(x):
        return 0.5 * x @ A @ x - b @ x + 2.0 * torch.sum(torch.sin(2.5 * x) ** 2) + 0.01 * (x ** 4).sum()
    out = {}
    for spec in (False, True):
        p = torch.full((n,), 0.3, requires_grad=True)
        opt = LBFGS([p], max_iter=12, lr=1.0, line_search_fn='strong_wolfe', history_size=4, _lib_override=emu_lib)
        opt.speculate = spec
        trace = []

        def closure():
            p.grad = None
            l = f(p)
            l.backward()
            trace.append(l.item())
            return l
        closure.discard_last = lambda: trace.pop()
        for _ in range(4):
            opt.step(closure)
        out[spec] = (list(trace), p.detach().clone(), list(opt._hist['order']), dict(opt.spec_stats))
    assert out[True][3]['issued'] > 0 and out[True][3]['rolled_back'] > 0, out[True][3]
    assert out[False][0] == out[True][0]
    assert torch.equal(out[False][1], out[True][1])
    assert out[False][2] == out[True][2]


@pytest.mark.slow          # (a minute on the emulator; the speculative issue is off by default)
def test_speculative_lbfgs_survives_rollback_of_a_non_finite_iteration(emu_lib):
    """Advisor, round 5: repeated step() calls at a converged point start with a vanishing pair (y = 0, H = y.s / y.y = 0 / 0); the iteration
    built on it and the speculative one behind it are non-finite, the host rejects them (curvature test) and rolls the speculation back.
    The dead slot's Gram rows / projections must not leak NaN into later directions: the speculative run stays finite and equal to the
    sequential one."""
    from humor_amd.lbfgs import LBFGS
    torch.manual_seed(0)
    n = 30
    A = torch.randn(n, n)
    A = A @ A.t() / n + 0.1 * torch.eye(n)
    b = torch.randn(n)

    def f(x):
        return 0.5 * x @ A @ x - b @ x + 0.01 * (x ** 4).sum()
    out = {}
    for spec in (False, True):
        p = torch.zeros(n, requires_grad=True)
        opt = LBFGS([p], max_iter=20, lr=1.0, line_search_fn='strong_wolfe', history_size=3, _lib_override=emu_lib)
        opt.speculate = spec
        trace = []

        def closure():
            p.grad = None
            l = f(p)
            l.backward()
            trace.append(l.item())
            return l
        closure.discard_last = lambda: trace.pop()
        for _ in range(4):
            opt.step(closure)
        H = opt._hist
        out[spec] = (list(trace), p.detach().clone(), bool(torch.isfinite(H['G']).all() and torch.isfinite(H['coef']).all()))
    assert all(np.isfinite(out[True][0])) and out[True][2], out[True]
    assert torch.isfinite(out[True][1]).all()
    assert out[False][0] == out[True][0]
    assert torch.equal(out[False][1], out[True][1])


def test_lbfgs_failed_curvature_test_keeps_the_oldest_pair(emu_lib):
    """With a full history a new pair goes to a spare slot: dropping it (curvature test failed, or a speculative iteration rolled back)
    puts the retired oldest pair back, untouched, as torch.optim.LBFGS keeps it; two allocations can be open at a time and are undone in
    reverse order; a retired pair's slot is handed out again only when its allocation can no longer be undone."""
    from humor_amd.lbfgs import LBFGS
    p = torch.zeros(40, requires_grad=True)
    opt = LBFGS([p], history_size=3, line_search_fn='strong_wolfe', _lib_override=emu_lib)
    opt._init_history(40, CPU)
    H = opt._hist
    slots = [opt._alloc_slot() for _ in range(3)]
    assert H['order'] == slots and len(set(slots)) == 3 and all(e is None for _, e in H['evicted'])
    H['M'][slots[0]].fill_(7.0)                       # the oldest pair's s row
    H['M'][slots[1]].fill_(8.0)
    s4 = opt._alloc_slot()
    assert s4 not in slots and H['order'] == slots[1:] + [s4] and H['evicted'][-1] == (s4, slots[0])
    opt._pop_pair(s4)                                 # curvature test failed
    assert H['order'] == slots and bool((H['M'][slots[0]] == 7.0).all())
    s5 = opt._alloc_slot()                            # the current iteration's pair ...
    s6 = opt._alloc_slot()                            # ... and the speculatively issued next iteration's: both retire a pair, both rows intact
    assert len({s5, s6} | set(slots)) == 5 and H['order'] == [slots[2], s5, s6]
    assert bool((H['M'][slots[0]] == 7.0).all()) and bool((H['M'][slots[1]] == 8.0).all())
    opt._pop_pair(s6)                                 # roll the speculative one back, then the current one (curvature test failed)
    opt._pop_pair(s5)
    assert H['order'] == slots and H['evicted'] in ([], [(slots[2], None)])
    s7 = opt._alloc_slot()                            # accepted this time
    s8 = opt._alloc_slot()
    s9 = opt._alloc_slot()                            # the allocation of s7 can no longer be undone: the pair it retired may be overwritten
    assert H['order'] == [s7, s8, s9] and s9 == slots[0]


def _arena_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from humor_amd.distributed import GradArena, Shard, allreduce_loss_and_grads
    torch.manual_seed(0)
    B = 5
    a = torch.randn(B, 3, requires_grad=True)
    b = torch.randn(B, 1, 4, requires_grad=True)
    frozen = torch.randn(B, 2)                     # a parameter that does not require gradients this phase
    w = torch.arange(1, B + 1, dtype=torch.float32).reshape(B, 1)
    sh = Shard(B)
    params = [a, frozen, b]

    def objective(get):
        ra, rb = get(a), get(b)
        rb2 = get(b)                               # a second reader of the same variable
        return ((ra * sh.sl(w)) ** 2).sum() + (rb.reshape(rb.shape[0], -1) * sh.sl(w)).sum() * 3.0 + (rb2 ** 3).sum() + (sh.sl(frozen) ** 2).sum()

    results = []
    for use_arena in (False, True, True):          # (twice with the arena: it is reused across evaluations)
        for p in params:
            p.grad = None
        if use_arena:
            if len(results) == 1:
                arena = GradArena(params, sh)
            assert arena.matches(params)
            arena.begin()
            loss = objective(lambda p: arena.rows(p))
            loss.backward()
            total = arena.allreduce(loss, None)
        else:
            loss = objective(sh.sl)
            loss.backward()
            total = allreduce_loss_and_grads(loss, params, None)
        results.append((total.item(), a.grad.clone(), b.grad.clone()))
    if rank == 0:
        torch.save(results, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_grad_arena_equals_packed_allreduce(tmp_path, world):
    """distributed.GradArena (the sharded closure's persistent packed gradient buffer: rows of a variable handed out once per evaluation,
    gradients dropped into the arena by the backward of that node, all-reduce in place, p.grad left as views) against the slice +
    cat + all-reduce + copy-back form it replaces: same loss and gradients, also on the arena's second evaluation, with a variable read
    twice and a frozen parameter in the list."""
    out = str(tmp_path / 'arena.pt')
    port = 29500 + (os.getpid() % 2000) + 31 * world
    mp.spawn(_arena_worker, args=(world, port, out), nprocs=world, join=True)
    res = torch.load(out)
    for r in res[1:]:
        assert abs(r[0] - res[0][0]) <= 1e-6 * abs(res[0][0])
        assert torch.allclose(r[1], res[0][1], rtol=1e-6, atol=1e-6) and torch.allclose(r[2], res[0][2], rtol=1e-6, atol=1e-6)
