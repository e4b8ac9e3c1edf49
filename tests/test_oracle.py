"""CPU tier: pins the oracle (oracle/*.py) -- against the real reference when /root/reference is present (build
container), against the committed golden vectors everywhere, and through analytic identities / gradcheck."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import humor_restated as H
from oracle import lbs_restated as L
from oracle import ref_loader
from humor_amd import synth

needs_ref = pytest.mark.skipif(not ref_loader.available(), reason='reference tree only exists in the build container')


def _layer(ds, N, selector=True, dtype=torch.float32):
    return L.SMPLHLayer(data_struct=ds, num_betas=16, batch_size=N, vertex_ids=L.VERTEX_IDS_SMPLH if selector else None,
                        dtype=dtype)


def test_lbs_zero_pose_is_template(smplh_struct):
    layer = _layer(smplh_struct, 2, selector=False)
    tr = torch.tensor([[0.1, -0.2, 0.3], [1.0, 2.0, 3.0]])
    out = layer(betas=torch.zeros(2, 16), global_orient=torch.zeros(2, 3), body_pose=torch.zeros(2, 63), transl=tr)
    vt = torch.tensor(smplh_struct.v_template)
    assert torch.allclose(out.vertices, vt[None] + tr[:, None], atol=2e-6)
    J0 = torch.tensor(smplh_struct.J_regressor) @ vt
    assert torch.allclose(out.joints, J0[None] + tr[:, None], atol=2e-6)


def test_lbs_root_rotation_is_rigid(smplh_struct):
    torch.manual_seed(0)
    layer = _layer(smplh_struct, 1, selector=False, dtype=torch.float64)
    body = 0.3 * torch.randn(1, 63, dtype=torch.float64)
    betas = torch.randn(1, 16, dtype=torch.float64)
    z = torch.zeros(1, 3, dtype=torch.float64)
    base = layer(betas=betas, global_orient=z, body_pose=body, transl=z)
    aa = torch.tensor([[0.3, -0.7, 0.5]], dtype=torch.float64)
    rot = layer(betas=betas, global_orient=aa, body_pose=body, transl=z)
    Rm = L.batch_rodrigues(aa)[0]
    root = base.joints[0, 0]
    expect = (base.vertices[0] - root) @ Rm.T + root
    assert torch.allclose(rot.vertices[0], expect, atol=1e-6)   # weight rows sum to 1 only to fp32 rounding


def test_lbs_fp32_vs_fp64(smplh_struct):
    g = torch.Generator().manual_seed(3)
    N = 3
    args = dict(betas=torch.randn(N, 16, generator=g), global_orient=0.5 * torch.randn(N, 3, generator=g),
                body_pose=0.4 * torch.randn(N, 63, generator=g), transl=torch.randn(N, 3, generator=g))
    o32 = _layer(smplh_struct, N)(**args)
    o64 = _layer(smplh_struct, N, dtype=torch.float64)(**{k: v.double() for k, v in args.items()})
    assert (o32.vertices.double() - o64.vertices).abs().max() < 1e-5
    assert o32.joints.shape == (N, 73, 3)


def test_lbs_gradcheck(smplh_struct):
    # tiny sub-model (first 40 vertices) keeps gradcheck fast
    ds = smplh_struct
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))
    V = 40
    posedirs = t(np.reshape(ds.posedirs[:V], [-1, ds.posedirs.shape[-1]]).T)
    parents = torch.tensor(ds.kintree_table[0].astype(np.int64))
    parents[0] = -1
    consts = (t(ds.v_template[:V]), t(ds.shapedirs[:V]), posedirs, t(ds.J_regressor[:, :V]), parents, t(ds.weights[:V]))
    torch.manual_seed(0)
    betas = torch.randn(1, 16, dtype=torch.float64, requires_grad=True)
    pose = (0.3 * torch.randn(1, 156, dtype=torch.float64)).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda b, p: L.lbs(b, p, *consts)[0], (betas, pose), eps=1e-6, atol=1e-5)


def test_selector_order():
    idx = L.selector_indices(L.VERTEX_IDS_SMPLH)
    assert idx.tolist() == [332, 6260, 2800, 4071, 583, 3216, 3226, 3387, 6617, 6624, 6787,
                            2746, 2319, 2445, 2556, 2673, 6191, 5782, 5905, 6016, 6133]
    from humor_amd.body_model import SMPLH_SELECTOR_VERTS
    assert idx.tolist() == SMPLH_SELECTOR_VERTS


def test_golden_smpl_matches_oracle(smplh_struct):
    gd = golden('smpl_bodymodel.npz')
    N = gd['betas'].shape[0]
    t = lambda k: torch.tensor(gd[k])
    out = _layer(smplh_struct, N)(betas=t('betas'), global_orient=t('root_orient'), body_pose=t('pose_body'), transl=t('trans'))
    assert np.abs(out.joints.detach().numpy() - gd['Jtr']).max() < 1e-6
    assert np.abs(out.vertices[:, gd['keep_verts']].detach().numpy() - gd['v_keep']).max() < 1e-6


def test_golden_rollout_matches_oracle():
    gd = golden('rollout.npz')
    sd = synth.humor_state_dict(seed=int(gd['weight_seed']))
    past0 = torch.tensor(gd['past0'], requires_grad=True)
    z = torch.tensor(gd['z'], requires_grad=True)
    world, (pm, pv) = H.roll_out(sd, past0, z)
    assert np.abs(world.detach().numpy() - gd['world']).max() < 1e-5
    assert np.abs(pm.detach().numpy() - gd['prior_mu']).max() < 1e-5
    assert np.abs(pv.detach().numpy() - gd['prior_var']).max() < 1e-5
    loss = (world * torch.tensor(gd['gw'])).sum() + (pm * torch.tensor(gd['gm'])).sum() + (pv * torch.tensor(gd['gv'])).sum()
    g0, gz = torch.autograd.grad(loss, [past0, z])
    assert np.abs(g0.numpy() - gd['g_past0']).max() < 2e-3 * max(1.0, np.abs(gd['g_past0']).max())
    assert np.abs(gz.numpy() - gd['g_z']).max() < 2e-3 * max(1.0, np.abs(gd['g_z']).max())
    outs = H.rollout_outputs(world.detach())
    assert np.abs(outs['root_orient'].numpy() - gd['aa_root']).max() < 1e-5
    assert np.abs(outs['pose_body'].numpy() - gd['aa_body']).max() < 1e-5


def test_golden_rotations_match_oracle():
    gd = golden('rotations.npz')
    aa = torch.tensor(gd['aa'], requires_grad=True)
    Rm = L.batch_rodrigues(aa)
    assert np.abs(Rm.detach().numpy() - gd['R']).max() == 0.0
    g_aa = torch.autograd.grad((Rm * torch.tensor(gd['gR'])).sum(), aa)[0]
    assert np.allclose(g_aa.numpy(), gd['g_aa'], rtol=1e-4, atol=1e-4, equal_nan=True)
    back = H.rot_to_aa(torch.tensor(gd['R']))
    assert np.abs(back.numpy() - gd['aa_back']).max() == 0.0


def test_golden_sampling_matches_oracle():
    """The restatement's sampling / canonicalising forms against the reference-generated vectors (no reference needed)."""
    from humor_amd import frames
    gd = golden('rollout_sampling.npz')
    sd = synth.humor_state_dict(seed=int(gd['weight_seed']))
    past, eps = torch.from_numpy(gd['past']), torch.from_numpy(gd['eps'])
    w, (pm, pv) = H.roll_out(sd, past, None, eps_seq=eps)
    assert np.abs(w.numpy() - gd['world_sampled']).max() < 1e-5
    assert np.abs((pm + eps * torch.sqrt(pv)).numpy() - gd['z_sampled']).max() < 1e-5
    n = gd['world_mean'].shape[1]
    wm, _ = H.roll_out(sd, past, None, eps_seq=torch.zeros(past.shape[0], n, 48))
    assert np.abs(wm.numpy() - gd['world_mean']).max() < 1e-5
    moved, z = torch.from_numpy(gd['moved']), torch.from_numpy(gd['z_canon'])
    local, (R0, t0, t2j) = frames.canonicalize_state(moved)
    wu, _ = H.roll_out(sd, local, z, G0=R0, gt0=t0, t2j=t2j)
    wc, _ = H.roll_out(sd, local, z, t2j=t2j)
    assert np.abs(wu.numpy() - gd['world_canon_uncanon']).max() < 1e-5
    assert np.abs(wc.numpy() - gd['world_canon']).max() < 1e-5


@needs_ref
def test_reference_rollout_live():
    """The restatement against the unmodified reference HumorModel.roll_out, forward and gradients."""
    R = ref_loader.load()
    sd = synth.humor_state_dict(seed=1, weight_scale=1.0)
    hm = R.humor_model.HumorModel(in_rot_rep='mat', out_rot_rep='aa', latent_size=48,
                                  model_data_config='smpl+joints+contacts', steps_in=1)
    hm.load_state_dict(sd)
    hm.eval()
    from oracle.make_golden import canonical_state
    g = torch.Generator().manual_seed(9)
    B, S = 5, 7
    past0 = canonical_state(B, g).requires_grad_(True)
    z = torch.randn(B, S, 48, generator=g).requires_grad_(True)
    names = ['trans', 'trans_vel', 'root_orient', 'root_orient_vel', 'pose_body', 'joints', 'joints_vel']
    dims = [3, 3, 9, 3, 189, 66, 66]
    d, o = {}, 0
    for k, n in zip(names, dims):
        d[k] = past0[:, o:o + n].unsqueeze(1)
        o += n
    pred, (pm, pv) = hm.roll_out(None, d, S, z_seq=z, return_prior=True)
    ref_world = torch.cat([pred[k] for k in names + ['contacts']], 2)
    world, (pm2, pv2) = H.roll_out(sd, past0, z)
    assert (world - ref_world).abs().max() < 1e-5
    assert (pm - pm2).abs().max() < 1e-5 and (pv - pv2).abs().max() < 1e-5
    gw = torch.randn(world.shape, generator=g)
    g1 = torch.autograd.grad((ref_world * gw).sum() + pm.sum(), [past0, z], retain_graph=True)
    g2 = torch.autograd.grad((world * gw).sum() + pm2.sum(), [past0, z])
    for a, b in zip(g1, g2):
        assert (a - b).abs().max() < 1e-3 * max(1.0, a.abs().max().item())


@needs_ref
def test_reference_sampling_and_canonicalize_live():
    """BASELINE config C1's path (test_humor sampling: z drawn from the conditional prior at every step) and the
    canonicalize_input / uncanonicalize_output options, restatement vs the unmodified reference (same noise)."""
    from humor_amd import frames
    from oracle import lbs_restated as L
    from oracle.make_golden import canonical_state
    R = ref_loader.load()
    sd = synth.humor_state_dict(seed=0)
    hm = R.humor_model.HumorModel(in_rot_rep='mat', out_rot_rep='aa', latent_size=48,
                                  model_data_config='smpl+joints+contacts', steps_in=1)
    hm.load_state_dict(sd)
    hm.eval()
    g = torch.Generator().manual_seed(3)
    B, S = 2, 8
    past = canonical_state(B, g)
    eps = torch.randn(B, S, 48, generator=g)
    names = ['trans', 'trans_vel', 'root_orient', 'root_orient_vel', 'pose_body', 'joints', 'joints_vel']
    dims = [3, 3, 9, 3, 189, 66, 66]

    def as_dict(p):
        d, o = {}, 0
        for k, n in zip(names, dims):
            d[k] = p[:, o:o + n].unsqueeze(1)
            o += n
        return d

    # sampling with the reference's torch.randn_like replaced by our noise sequence
    it = iter([eps[:, t] for t in range(S)])
    orig = torch.randn_like
    torch.randn_like = lambda x: next(it)
    try:
        with torch.no_grad():
            pred = hm.roll_out(None, as_dict(past), S)
    finally:
        torch.randn_like = orig
    ref_w = torch.cat([pred[k] for k in names + ['contacts']], 2)
    w, _ = H.roll_out(sd, past, None, eps_seq=eps)
    assert (w - ref_w).abs().max() < 1e-5
    # a state moved rigidly in the world, canonicalised by the model
    ang = torch.tensor([0.7, -1.2])
    Rz = L.batch_rodrigues(torch.stack([torch.zeros(2), torch.zeros(2), ang], 1))
    shift = torch.tensor([[1.5, -0.7, 0.0], [-2.0, 0.3, 0.0]])
    rot = lambda v: torch.einsum('bij,bj->bi', Rz, v)
    j = torch.einsum('bij,bkj->bki', Rz, past[:, 207:273].reshape(B, 22, 3)) + shift.unsqueeze(1)
    jv = torch.einsum('bij,bkj->bki', Rz, past[:, 273:339].reshape(B, 22, 3))
    moved = torch.cat([rot(past[:, 0:3]) + shift, rot(past[:, 3:6]), torch.matmul(Rz, past[:, 6:15].reshape(B, 3, 3)).reshape(B, 9),
                       rot(past[:, 15:18]), past[:, 18:207], j.reshape(B, 66), jv.reshape(B, 66)], 1)
    z = torch.randn(B, 5, 48, generator=g)
    local, (R0, t0, t2j) = frames.canonicalize_state(moved)
    for unc in (True, False):
        with torch.no_grad():
            pred = hm.roll_out(None, as_dict(moved.clone()), 5, z_seq=z, canonicalize_input=True, uncanonicalize_output=unc)
        ref_w = torch.cat([pred[k] for k in names + ['contacts']], 2)
        w, _ = H.roll_out(sd, local, z, G0=R0 if unc else None, gt0=t0 if unc else None, t2j=t2j)
        assert (w - ref_w).abs().max() < 1e-5
        if unc:
            wc, _ = H.roll_out(sd, local, z, t2j=t2j)
            assert (frames.uncanonicalize_world(wc, R0, t0, t2j) - w).abs().max() < 1e-5


@needs_ref
def test_reference_bodymodel_live(smplh_npz, smplh_struct):
    """The reference's BodyModel wrapper (with the smplx shim) slices/returns what our restated layer computes."""
    R = ref_loader.load()
    N = 3
    bm = R.body_model.BodyModel(smplh_npz, num_betas=16, batch_size=N, use_vtx_selector=False)
    g = torch.Generator().manual_seed(2)
    args = dict(root_orient=0.4 * torch.randn(N, 3, generator=g), pose_body=0.4 * torch.randn(N, 63, generator=g),
                betas=torch.randn(N, 16, generator=g), trans=torch.randn(N, 3, generator=g))
    out = bm(**args)
    assert out.Jtr.shape == (N, 52, 3) and out.v.shape == (N, 6890, 3) and out.f.shape == (13776, 3)
    assert out.full_pose.shape == (N, 156) and out.pose_hand.shape == (N, 90)
    assert R.transforms.batch_rodrigues(args['root_orient']).equal(L.batch_rodrigues(args['root_orient']))


@pytest.mark.parametrize('kind', ['amass', 'rgb'])
def test_restated_closure_matches_reference_fixture(smplh_struct, kind):
    """oracle/closure_restated.py (the bench's CPU baseline) reproduces the reference MotionOptimizer's stage-3 objective."""
    from oracle import closure_cases as CC
    from oracle.closure_restated import RestatedFit
    gd = golden(f'closure_{kind}.npz')
    B, T = int(gd['B']), int(gd['T'])
    case = CC.make_case(kind, B, T, seed=int(gd['seed']))
    rgb = kind == 'rgb'
    fit = RestatedFit(smplh_struct, synth.humor_state_dict(seed=0), synth.SynthVPoser(seed=0), synth.make_gmm(seed=0),
                      CC.RGB_WEIGHTS if rgb else CC.AMASS_WEIGHTS, B, T, rgb, CC.camera_matrix(B) if rgb else None)
    var = {k: v.clone() for k, v in case['var'].items()}
    for k in ('trans', 'root_orient', 'latent_pose'):
        var[k] = var[k][:, :1]
    var = {k: v.requires_grad_(True) for k, v in var.items()}
    loss = fit.objective(var, case['obs'])
    ref = float(gd['s2_loss'])
    assert abs(loss.item() - ref) <= 1e-4 * abs(ref), (loss.item(), ref)
    g = torch.autograd.grad(loss, [var['latent_motion'], var['betas']])
    assert np.abs(g[0].numpy() - gd['s2_g_latent_motion']).max() <= 2e-3 * max(1.0, np.abs(gd['s2_g_latent_motion']).max())
    assert np.abs(g[1].numpy() - gd['s2_g_betas']).max() <= 2e-3 * max(1.0, np.abs(gd['s2_g_betas']).max())


def test_chamfer_restatement_matches_compiled_reference():
    """oracle/chamfer_restated.py against the reference's own chamfer_distance.cpp compiled into oracle/_ref (oracle/build_ref.py):
    indices and distances bit-exact (ties included), gradients to rounding."""
    from oracle import build_ref, chamfer_restated as CR
    cd = build_ref.load()
    if cd is None:
        pytest.skip('neither /root/reference nor a prebuilt oracle/_ref is available')
    g = torch.Generator().manual_seed(0)
    b, n, m = 3, 257, 700
    x1, x2 = torch.randn(b, n, 3, generator=g), torch.randn(b, m, 3, generator=g)
    x2[:, 10] = x2[:, 5]
    x1[:, 3] = x2[:, 5]
    d1, d2 = torch.zeros(b, n), torch.zeros(b, m)
    i1, i2 = torch.zeros(b, n, dtype=torch.int), torch.zeros(b, m, dtype=torch.int)
    cd.forward(x1, x2, d1, d2, i1, i2)
    r = CR.forward(x1.numpy(), x2.numpy())
    assert np.array_equal(r[1], i1.numpy()) and np.array_equal(r[3], i2.numpy())
    assert np.array_equal(r[0], d1.numpy()) and np.array_equal(r[2], d2.numpy())
    assert i1[0, 3].item() == 5          # the tie goes to the lower index
    g1, g2 = torch.randn(b, n, generator=g), torch.randn(b, m, generator=g)
    gx1, gx2 = torch.zeros(b, n, 3), torch.zeros(b, m, 3)
    cd.backward(x1, x2, gx1, gx2, g1, g2, i1, i2)
    rg = CR.backward(x1.numpy(), x2.numpy(), g1.numpy(), i1.numpy(), g2.numpy(), i2.numpy())
    assert np.abs(rg[0] - gx1.numpy()).max() < 1e-5 and np.abs(rg[1] - gx2.numpy()).max() < 1e-5


@pytest.mark.skipif(not ref_loader.available(), reason='reference tree only exists in the build container')
def test_rot6d_formula_matches_reference_live():
    """The formula tests/rollout_checks.check_rot6d holds the kernel to is the reference's rot6d_to_rotmat (transforms.py:201-220), bit for bit."""
    R = ref_loader.load()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(64, 6, generator=g)          # (not 3 rows: the reference's dim-less torch.cross)
    v = x.view(-1, 3, 2)
    a1, a2 = v[:, :, 0], v[:, :, 1]
    b1 = torch.nn.functional.normalize(a1)
    b2 = torch.nn.functional.normalize(a2 - torch.einsum('bi,bi->b', b1, a2).unsqueeze(-1) * b1)
    ours = torch.stack((b1, b2, torch.cross(b1, b2, dim=1)), dim=-1)
    assert torch.equal(ours, R.transforms.rot6d_to_rotmat(x))


@pytest.mark.parametrize('rep', ['6d', '9d', 'nd'])
def test_oracle_rollout_rotation_representations_match_the_golden_fixture(rep):
    """The restated roll-out with the 6-D / 9-D residual rotations against the reference-generated rollout_rotrep.npz."""
    from conftest import golden
    from humor_amd import synth
    from oracle import humor_restated as H
    gd = golden('rollout_rotrep.npz')
    p = 'r' + rep + '_'
    delta = rep != 'nd'          # 'nd': axis-angle outputs, output_delta=False
    sd = synth.rotrep_state_dict(rep, seed=int(gd['weight_seed'])) if delta else synth.nodelta_state_dict(seed=int(gd['weight_seed']))
    past = torch.tensor(gd[p + 'past0']).requires_grad_(True)
    z = torch.tensor(gd[p + 'z']).requires_grad_(True)
    world, (pm, pv) = H.roll_out(sd, past, z, output_delta=delta)
    assert np.abs(world.detach().numpy() - gd[p + 'world']).max() < 2e-5
    assert np.abs(pm.detach().numpy() - gd[p + 'prior_mu']).max() < 2e-5
    t = lambda k: torch.tensor(gd[p + k])
    g0, gz = torch.autograd.grad((world * t('gw')).sum() + (pm * t('gm')).sum() + (pv * t('gv')).sum(), [past, z])
    assert np.abs(g0.numpy() - gd[p + 'g_past0']).max() < 3e-4 * max(1.0, np.abs(gd[p + 'g_past0']).max())
    assert np.abs(gz.numpy() - gd[p + 'g_z']).max() < 3e-4 * max(1.0, np.abs(gd[p + 'g_z']).max())


@pytest.mark.skipif(not ref_loader.available(), reason='reference tree only exists in the build container')
def test_rot9d_formula_matches_reference_live():
    """The formula tests/rollout_checks.check_rot9d holds the kernel to is the reference's rot9d_to_rotmat (transforms.py:222-241), bit for bit."""
    import rollout_checks as RC
    R = ref_loader.load()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(64, 9, generator=g)
    assert torch.equal(RC.rot9d_reference_formula(x), R.transforms.rot9d_to_rotmat(x))


@pytest.mark.skipif(not ref_loader.available(), reason='reference tree only exists in the build container')
def test_infer_global_seq_and_velocities_match_reference_live():
    """SURVEY 8(a) a15, directly: HumorModel.infer_global_seq (all T-1 frame pairs canonicalised and pushed through prior + posterior
    in one batch) against the reference's per-frame Python loop (humor_model.py:1061-1203), and the finite-difference velocity
    estimators against motion_optimizer.py:766-800."""
    from humor_amd import frames
    from humor_amd.humor_model import HumorModel
    R = ref_loader.load()
    sd = synth.humor_state_dict(seed=0)
    ours = HumorModel(in_rot_rep='mat', out_rot_rep='aa', latent_size=48, model_data_config='smpl+joints+contacts', steps_in=1)
    ref = R.humor_model.HumorModel(in_rot_rep='mat', out_rot_rep='aa', latent_size=48, model_data_config='smpl+joints+contacts', steps_in=1)
    ours.load_state_dict(sd)
    ref.load_state_dict(sd)
    ours.eval()
    ref.eval()
    B, T = 2, 9
    g = torch.Generator().manual_seed(3)
    r = lambda *s, sc=1.0: sc * torch.randn(*s, generator=g)
    seq = {'trans': r(B, T, 3, sc=0.5), 'trans_vel': r(B, T, 3, sc=0.3),
           'root_orient': L.batch_rodrigues(r(B * T, 3, sc=0.7)).reshape(B, T, 9), 'root_orient_vel': r(B, T, 3, sc=0.3),
           'pose_body': L.batch_rodrigues(r(B * T * 21, 3, sc=0.4)).reshape(B, T, 189), 'joints': r(B, T, 66, sc=0.4), 'joints_vel': r(B, T, 66, sc=0.3)}
    with torch.no_grad():
        (pm, pv), (qm, qv) = ours.infer_global_seq({k: v.clone() for k, v in seq.items()})
        (pm_r, pv_r), (qm_r, qv_r) = ref.infer_global_seq({k: v.clone() for k, v in seq.items()})
    for a, b_, name in ((pm, pm_r, 'prior mean'), (pv, pv_r, 'prior var'), (qm, qm_r, 'posterior mean'), (qv, qv_r, 'posterior var')):
        assert a.shape == b_.shape == (B, T - 1, 48), name
        assert (a - b_).abs().max().item() < 2e-5 * max(1.0, b_.abs().max().item()), (name, (a - b_).abs().max().item())
    # velocity estimators (the reference methods do not touch `self`)
    h = 1.0 / 30
    x = r(B, T, 22, 3)
    assert torch.equal(frames.estimate_linear_velocity(x, h), R.motion_optimizer.MotionOptimizer.estimate_linear_velocity(None, x, h))
    rot = L.batch_rodrigues(r(B * T, 3, sc=0.8)).reshape(B, T, 3, 3)

    class _S:
        estimate_linear_velocity = staticmethod(lambda d, hh: R.motion_optimizer.MotionOptimizer.estimate_linear_velocity(None, d, hh))
    w_ref = R.motion_optimizer.MotionOptimizer.estimate_angular_velocity(_S, rot, h)
    assert (frames.estimate_angular_velocity(rot, h) - w_ref).abs().max().item() < 1e-6


@pytest.mark.skipif(not ref_loader.available(), reason='reference tree only exists in the build container')
@pytest.mark.parametrize('rep', ['aa', '6d'])
def test_full_forward_pass_matches_reference_live(rep):
    """HumorModel.forward / single_step and infer_global_seq(full_forward_pass=True) (humor_model.py:352-404, 1131-1160) against the
    reference with the posterior's sample replaced by its mean on both sides (the reference draws per step, ours once for all pairs):
    every prediction key and both distributions.  (Not 3 rows: the reference's dim-less torch.cross in rot6d_to_rotmat)"""
    from humor_amd.humor_model import HumorModel
    R = ref_loader.load()
    sd = synth.humor_state_dict(seed=0) if rep == 'aa' else synth.rotrep_state_dict(rep, seed=0)
    kw = dict(in_rot_rep='mat', out_rot_rep=rep, latent_size=48, model_data_config='smpl+joints+contacts', steps_in=1)
    ours, ref = HumorModel(**kw), R.humor_model.HumorModel(**kw)
    for m in (ours, ref):
        m.load_state_dict(sd)
        m.eval()
        m.rsample = lambda mu, var: mu
    B, T = 2, 5
    g = torch.Generator().manual_seed(8)
    r = lambda *s, sc=1.0: sc * torch.randn(*s, generator=g)
    seq = {'trans': r(B, T, 3, sc=0.5), 'trans_vel': r(B, T, 3, sc=0.3),
           'root_orient': L.batch_rodrigues(r(B * T, 3, sc=0.7)).reshape(B, T, 9), 'root_orient_vel': r(B, T, 3, sc=0.3),
           'pose_body': L.batch_rodrigues(r(B * T * 21, 3, sc=0.4)).reshape(B, T, 189), 'joints': r(B, T, 66, sc=0.4), 'joints_vel': r(B, T, 66, sc=0.3)}
    with torch.no_grad():
        a = ours.infer_global_seq({k: v.clone() for k, v in seq.items()}, full_forward_pass=True)
        b = ref.infer_global_seq({k: v.clone() for k, v in seq.items()}, full_forward_pass=True)
    assert set(a.keys()) == set(b.keys()), (sorted(a.keys()), sorted(b.keys()))
    for k in b:
        pa, pb = (a[k], b[k]) if isinstance(b[k], tuple) else ((a[k],), (b[k],))
        for x, y in zip(pa, pb):
            assert x.shape == y.shape, (k, x.shape, y.shape)
            assert (x - y).abs().max().item() < 3e-5 * max(1.0, y.abs().max().item()), (k, (x - y).abs().max().item())


def test_lbs_restatement_matches_independent_paper_derivation(smplh_npz, smplh_struct):
    """The smplx restatement (written from the smplx op sequence) against oracle/smpl_paper_numpy.py, an fp64 numpy derivation
    written from the SMPL paper's equations with a different structure: breaks common-mode error between the restatement and the
    kernels while the real package is unavailable ("parity unpinned")."""
    from oracle import smpl_paper_numpy as SP
    data = np.load(smplh_npz)
    N = 3
    g = torch.Generator().manual_seed(4)
    r = lambda *s, sc=1.0: sc * torch.randn(*s, generator=g)
    inp = dict(global_orient=r(N, 3, sc=0.6).double(), body_pose=r(N, 63, sc=0.5).double(), left_hand_pose=r(N, 45, sc=0.3).double(),
               right_hand_pose=r(N, 45, sc=0.3).double(), betas=r(N, 16).double(), transl=r(N, 3).double())
    layer = L.SMPLHLayer(data_struct=smplh_struct, num_betas=16, batch_size=N, vertex_ids=L.VERTEX_IDS_SMPLH, dtype=torch.float64)
    out = layer(**inp)
    kin = np.asarray(data['kintree_table'])[0].astype(np.int64)
    model = dict(v_template=data['v_template'], shapedirs=data['shapedirs'], posedirs=data['posedirs'], J_regressor=data['J_regressor'],
                 weights=data['weights'], parents=[-1] + [int(p) for p in kin[1:]])
    for i in range(N):
        pose = np.concatenate([inp[k][i].numpy() for k in ('global_orient', 'body_pose', 'left_hand_pose', 'right_hand_pose')])
        v, j = SP.smpl_frame(model, pose, inp['betas'][i].numpy(), inp['transl'][i].numpy())
        assert np.abs(out.vertices[i].numpy() - v).max() < 1e-7
        assert np.abs(out.joints[i, :52].numpy() - j).max() < 1e-7


@needs_ref
def test_index_tables_equal_the_reference_live():
    """every fixed index table of humor_amd/tables.py equals the reference's own table, entry for entry
    (body_model/utils.py:5-19, 53-56; datasets/amass_utils.py:21-23; fitting/fitting_utils.py:678-680)."""
    from humor_amd import tables as T
    R = ref_loader.load()
    assert T.SMPL_JOINTS == R.bm_utils.SMPL_JOINTS
    assert list(T.SMPL_PARENTS) == list(R.bm_utils.SMPL_PARENTS)
    assert list(T.KEYPT_VERTS) == list(R.bm_utils.KEYPT_VERTS)
    assert list(T.CONTACT_ORDERING) == list(R.amass_utils.CONTACT_ORDERING)
    assert list(T.CONTACT_INDS) == list(R.amass_utils.CONTACT_INDS)
    op = R.bm_utils.smpl_to_openpose('smplh', use_hands=False, use_face=False, use_face_contour=False, openpose_format='coco25')
    assert [int(i) for i in op] == list(T.SMPLH_TO_OPENPOSE25)
    assert T.OP_NUM_JOINTS == R.fitting_utils.OP_NUM_JOINTS
    assert list(T.OP_IGNORE_JOINTS) == list(R.fitting_utils.OP_IGNORE_JOINTS)
    assert [list(e) for e in T.OP_EDGE_LIST] == [list(e) for e in R.fitting_utils.OP_EDGE_LIST]
    assert T.NUM_BODY_JOINTS == len(R.bm_utils.SMPL_JOINTS) - 1
