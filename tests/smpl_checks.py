"""Shared SMPL parity checks: the same assertions run against the host SIMT-emulator build (CPU tier, tiny sizes)
and against the gfx950 build on a real MI355X (GPU tier, through the C ABI)."""
import numpy as np
import torch

from conftest import golden
from humor_amd.body_model import BodyModel
from oracle import lbs_restated as L

from humor_amd.tables import KEYPT_VERTS   # noqa: E402  (body_model/utils.py:17-19)

FWD_TOL = 1e-4      # north_star: fp32 vertices within 1e-4
GRAD_RTOL = 2e-4    # gradients: relative to the largest reference gradient entry


def make_inputs(N, seed, device, hands=False):
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(device).requires_grad_(True)
    d = dict(root_orient=mk(N, 3, sc=0.5), pose_body=mk(N, 63, sc=0.4), betas=mk(N, 16), trans=mk(N, 3))
    if hands:
        d['pose_hand'] = mk(N, 90, sc=0.3)
    return d


def oracle_forward(ds, inputs, selector):
    N = inputs['betas'].shape[0]
    layer = L.SMPLHLayer(data_struct=ds, num_betas=16, batch_size=N, vertex_ids=L.VERTEX_IDS_SMPLH if selector else None)
    cpu = {k: v.detach().cpu().requires_grad_(True) for k, v in inputs.items()}
    kw = dict(betas=cpu['betas'], global_orient=cpu['root_orient'], body_pose=cpu['pose_body'], transl=cpu['trans'])
    if 'pose_hand' in cpu:
        kw.update(left_hand_pose=cpu['pose_hand'][:, :45], right_hand_pose=cpu['pose_hand'][:, 45:])
    return layer(**kw), cpu


def check_forward_backward(lib, npz, ds, N, device, seed=0, hands=False, selector=True, subset=None, algo=0, dense_grad=False):
    inputs = make_inputs(N, seed, device, hands)
    ref, cpu = oracle_forward(ds, inputs, selector)
    bm = BodyModel(npz, num_betas=16, batch_size=N, use_vtx_selector=selector, vertex_subset=subset, algo=algo,
                   _lib_override=lib)
    out = bm(**inputs)
    ref_v = ref.vertices if subset is None else ref.vertices[:, subset]
    assert out.v.shape == ref_v.shape and out.Jtr.shape == ref.joints.shape
    dv = (out.v.detach().cpu() - ref_v).abs().max().item()
    dj = (out.Jtr.detach().cpu() - ref.joints).abs().max().item()
    assert dv < FWD_TOL and dj < FWD_TOL, (dv, dj)
    # backward: joints always, vertices when a subset is used or a dense gradient is requested
    g = torch.Generator().manual_seed(seed + 100)
    gJ = torch.randn(ref.joints.shape, generator=g)
    loss_ref = (ref.joints * gJ).sum()
    loss = (out.Jtr * gJ.to(device)).sum()
    if subset is not None or dense_grad:
        gV = torch.randn(ref_v.shape, generator=g)
        loss_ref = loss_ref + (ref_v * gV).sum()
        loss = loss + (out.v * gV.to(device)).sum()
    keys = list(inputs.keys())
    g_ref = torch.autograd.grad(loss_ref, [cpu[k] for k in keys])
    g_our = torch.autograd.grad(loss, [inputs[k] for k in keys])
    for k, a, b in zip(keys, g_ref, g_our):
        scale = max(1.0, a.abs().max().item())
        err = (a - b.cpu()).abs().max().item()
        assert err < GRAD_RTOL * scale, (k, err, scale)
    return dv, dj


def check_golden(lib, npz, device):
    gd = golden('smpl_bodymodel.npz')
    t = lambda k: torch.tensor(gd[k]).to(device).requires_grad_(True)
    inputs = dict(root_orient=t('root_orient'), pose_body=t('pose_body'), betas=t('betas'), trans=t('trans'))
    N = gd['betas'].shape[0]
    bm = BodyModel(npz, num_betas=16, batch_size=N, use_vtx_selector=True, _lib_override=lib)
    out = bm(**inputs)
    assert np.abs(out.Jtr.detach().cpu().numpy() - gd['Jtr']).max() < FWD_TOL
    assert np.abs(out.v[:, gd['keep_verts']].detach().cpu().numpy() - gd['v_keep']).max() < FWD_TOL
    # the fitting-style subset evaluation and its gradients
    bm2 = BodyModel(npz, num_betas=16, use_vtx_selector=True, vertex_subset=gd['keypt_verts'].tolist(), _lib_override=lib)
    o2 = bm2(**inputs)
    assert np.abs(o2.v.detach().cpu().numpy() - gd['v_keypt']).max() < FWD_TOL
    loss = (o2.Jtr * torch.tensor(gd['gJ']).to(device)).sum() + (o2.v * torch.tensor(gd['gV']).to(device)).sum()
    grads = torch.autograd.grad(loss, [inputs[k] for k in ('root_orient', 'pose_body', 'betas', 'trans')])
    for k, gr in zip(('g_root', 'g_body', 'g_betas', 'g_trans'), grads):
        scale = max(1.0, np.abs(gd[k]).max())
        assert np.abs(gr.cpu().numpy() - gd[k]).max() < GRAD_RTOL * scale, k


def check_skin_variants(lib, npz, ds, device, N=3, variants=(5, 4, 6, 0, 1, 2), seed=0):
    """ha_lbs_skin on its own (the streaming LBS kernel of the roofline): every launch geometry selectable through
    ha_tune_set('skin_variant') must give the same bits, and those must match the fp64 linear-blend formula
    verts = (sum_q w_q A_q) [v; 1] + transl  (smplx lbs.py:223-233) to fp32 rounding."""
    from humor_amd import _lib
    V, J = ds.v_template.shape[0], ds.weights.shape[1]
    g = torch.Generator().manual_seed(seed)
    vp = torch.randn(N * V * 3 + 4, generator=g)
    A = torch.randn(N, J, 12, generator=g)
    tr = torch.randn(N, 3, generator=g)
    W = torch.from_numpy(np.asarray(ds.weights)).double()                      # [V,J]
    # kernel layout of A: [R00 R01 R02 R10 | R11 R12 R20 R21 | R22 t0 t1 t2]
    R = A[:, :, :9].reshape(N, J, 3, 3).double()
    t = A[:, :, 9:12].double()
    Tr = torch.einsum('vj,njab->nvab', W, R)
    Tt = torch.einsum('vj,nja->nva', W, t)
    v = vp[:N * V * 3].reshape(N, V, 3).double()
    ref = torch.einsum('nvab,nvb->nva', Tr, v) + Tt + tr.double().unsqueeze(1)
    h = BodyModel(npz, num_betas=16, _lib_override=lib)._handle_for(device)
    vp_d, A_d, tr_d = vp.to(device), A.to(device), tr.to(device)
    outs = {}
    try:
        for var in variants:
            lib.call('ha_tune_set', b'skin_variant', var)
            out = torch.full((N, V, 3), float('nan'), device=device)
            lib.call('ha_lbs_skin', h.ptr, N, _lib.ptr(vp_d), _lib.ptr(A_d), _lib.ptr(tr_d), _lib.ptr(out), _lib.stream_ptr(out))
            outs[var] = out.cpu()
    finally:
        lib.call('ha_tune_set', b'skin_variant', -1)
    first = outs[variants[0]]
    scale = ref.abs().max().item()
    assert (first.double() - ref).abs().max().item() < 2e-6 * scale
    for var, o in outs.items():
        assert torch.equal(o, first), f'skin variant {var} differs from variant {variants[0]}'


def check_fused_forward(lib, npz, ds, N, device, seed=0, hands=False):
    """Forward-only dense call (no input requires a gradient): BodyModel takes ha_smpl_forward algo 3 -- the pose-blend GEMM with the
    skinning in its epilogue, v_posed never materialised.  Same arithmetic as blend + lbs_skin (algo 2): the vertices must be
    bit-identical to the two-kernel path, and both within the forward tolerance of the oracle."""
    inputs = {k: v.detach() for k, v in make_inputs(N, seed, device, hands).items()}
    ref, _ = oracle_forward(ds, inputs, True)
    bm = BodyModel(npz, num_betas=16, batch_size=N, use_vtx_selector=True, _lib_override=lib)
    bm.fused_min_frames = 0                                               # (the default policy takes this path from 4096 frames on)
    with torch.no_grad():
        fused = bm(**inputs)                                              # algo 0 (auto) + nothing requires grad -> algo 3
    two = BodyModel(npz, num_betas=16, batch_size=N, use_vtx_selector=True, algo=2, _lib_override=lib)(**inputs)
    assert torch.equal(fused.v, two.v), (fused.v - two.v).abs().max().item()
    assert torch.equal(fused.Jtr, two.Jtr)
    dv = (fused.v.cpu() - ref.vertices.detach()).abs().max().item()
    assert dv < FWD_TOL, dv
    return dv


def check_parts_api(lib, npz, device, B=2, T=3, seed=0):
    """ha_smpl_forward_parts / ha_smpl_backward_parts / ha_seq_sum_add (split pose pointers, one shape row per sequence, a reduced number of
    joint-gradient rows, addends on every output gradient) against BodyModel's split path on the expanded inputs + the additions autograd
    would launch: same kernels, so forward bit-identical and gradients to rounding of the addend sums."""
    import ctypes as C
    from humor_amd import _lib
    N = B * T
    bm = BodyModel(npz, num_betas=16, batch_size=N, use_vtx_selector=True, vertex_subset=KEYPT_VERTS, _lib_override=lib)
    sm = bm.parts_config(device)
    assert sm is not None
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(device)
    root, body, betas, trans = rnd(N, 3, sc=0.5), rnd(N, 63, sc=0.4), rnd(B, 16), rnd(N, 3)
    betas_x = betas.reshape(B, 1, 16).expand(B, T, 16).reshape(N, 16).clone()
    leaf = [t.clone().requires_grad_(True) for t in (root, body, betas_x, trans)]
    out = bm(root_orient=leaf[0], pose_body=leaf[1], betas=leaf[2], trans=leaf[3])
    h, st = sm['handle'], C.c_void_p(torch.cuda.current_stream(device).cuda_stream if device.type == 'cuda' else 0)
    jx, nv = sm['J'] + sm['n_sel'], sm['n_all'] - sm['n_sel']
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)
    joints, verts = new(N, jx, 3), new(N, nv, 3)
    p = _lib.ptr
    lib.call('ha_smpl_forward_parts', h.ptr, sm['slot_all'], N, sm['n_active'], p(root), p(body), p(betas), T, p(trans), sm['n_sel'], p(joints), p(verts), st)
    assert torch.equal(joints, out.Jtr.detach()) and torch.equal(verts, out.v.detach())
    # backward: only the first 22 joint rows carry a gradient (a [N,22,3] tensor), every output gradient has an addend
    gj22, gv = rnd(N, 22, 3), rnd(N, nv, 3)
    add_root, add_body, add_betas, add_tr = rnd(N, 3), rnd(N, 63), rnd(N, 16), rnd(N, 3)
    loss = (out.Jtr[:, :22] * gj22).sum() + (out.v * gv).sum()
    ref = torch.autograd.grad(loss, leaf)
    g_root, g_body, g_bf, g_tr = new(N, 3), new(N, 63), new(N, 16), new(N, 3)
    lib.call('ha_smpl_backward_parts', h.ptr, sm['slot_all'], N, sm['n_active'], p(root), p(body), p(betas), T, sm['n_sel'], p(gj22), 22, 22, p(gv),
             p(add_root), p(add_body), p(add_betas), p(add_tr), p(g_root), p(g_body), p(g_bf), p(g_tr), st)
    worst = 0.0
    for got, want in ((g_root, ref[0] + add_root), (g_body, ref[1] + add_body), (g_bf, ref[2] + add_betas), (g_tr, ref[3] + add_tr)):
        e = (got - want).abs().max().item() / max(1.0, want.abs().max().item())
        worst = max(worst, e)
        assert e < 1e-5, e
    # joints only (n_head = 0, no vertex tensor): the first J rows of the full evaluation; its adjoint = the full adjoint with zero vertex gradients
    jo = new(N, sm['J'], 3)
    lib.call('ha_smpl_forward_parts', h.ptr, sm['slot_all'], N, sm['n_active'], p(root), p(body), p(betas), T, p(trans), 0, p(jo), None, st)
    assert torch.equal(jo, joints[:, :sm['J']].contiguous())
    out2 = bm(root_orient=leaf[0], pose_body=leaf[1], betas=leaf[2], trans=leaf[3])
    ref_j = torch.autograd.grad((out2.Jtr[:, :22] * gj22).sum(), leaf)
    r2, b2, f2, t2 = new(N, 3), new(N, 63), new(N, 16), new(N, 3)
    lib.call('ha_smpl_backward_parts', h.ptr, sm['slot_all'], N, sm['n_active'], p(root), p(body), p(betas), T, 0, p(gj22), 22, 22, None,
             None, None, None, None, p(r2), p(b2), p(f2), p(t2), st)
    for got, want in ((r2, ref_j[0]), (b2, ref_j[1]), (f2, ref_j[2]), (t2, ref_j[3])):
        assert (got - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    # per-frame shape gradients back to one row per sequence, with two addends
    a1, a2 = rnd(B, 16), rnd(B, 16)
    g_betas = new(B, 16)
    lib.call('ha_seq_sum_add', B, T, 16, p(g_bf), p(a1), p(a2), p(g_betas), st)
    want = g_bf.reshape(B, T, 16).sum(1) + a1 + a2
    assert (g_betas - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    return worst
