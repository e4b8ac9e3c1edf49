"""GPU tier (-m gpu): SMPL parity on a real MI355X, through the C ABI of the gfx950 build."""
import numpy as np
import pytest
import torch

import smpl_checks as SC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


@pytest.mark.parametrize('N,hands', [(1, False), (33, False), (64, True), (130, False), (1920, False)])
def test_fused_blend_skin_forward(gpu_lib, smplh_npz, smplh_struct, dev, N, hands):
    """Forward-only dense calls take the fused blend + skin kernel (ha_smpl_forward algo 3): bit-identical to the two-kernel path."""
    print('fused dense forward vs oracle', SC.check_fused_forward(gpu_lib, smplh_npz, smplh_struct, N=N, device=dev, seed=N, hands=hands))


@pytest.mark.parametrize('N', [1, 2, 7, 64, 130])
@pytest.mark.parametrize('algo', [1, 2])
def test_dense_forward_backward(gpu_lib, smplh_npz, smplh_struct, dev, N, algo):
    SC.check_forward_backward(gpu_lib, smplh_npz, smplh_struct, N=N, device=dev, seed=N, algo=algo)


@pytest.mark.parametrize('N', [1, 5, 60, 257])
def test_subset_forward_backward(gpu_lib, smplh_npz, smplh_struct, dev, N):
    SC.check_forward_backward(gpu_lib, smplh_npz, smplh_struct, N=N, device=dev, seed=N + 1, subset=SC.KEYPT_VERTS)


def test_hands_and_dense_gradient(gpu_lib, smplh_npz, smplh_struct, dev):
    SC.check_forward_backward(gpu_lib, smplh_npz, smplh_struct, N=9, device=dev, hands=True, selector=False, algo=1, dense_grad=True)
    SC.check_forward_backward(gpu_lib, smplh_npz, smplh_struct, N=9, device=dev, hands=True, selector=True, algo=2, dense_grad=True)


@pytest.mark.parametrize('N,hands,dA', [(70, False, 2), (130, True, 2), (70, True, 1), (33, False, 0)])
def test_dense_backward_kernels(gpu_lib, smplh_npz, smplh_struct, dev, N, hands, dA):
    """Every vertex carries a gradient (point-cloud / chamfer term): ha_smpl_backward_dense (streaming dL/dv_posed, MFMA dL/dA and
    dL/dcoeff with K split, chain adjoint) against the oracle's autograd; frame counts that leave a ragged 64-frame row pair.  dA: the
    dL/dA variant (ha_tune_set "dense_gA_sparse": 2 = the default chunk-compressed MFMA product, 1 = joint lists, 0 = dense product)."""
    gpu_lib.call('ha_tune_set', b'dense_gA_sparse', dA)
    try:
        SC.check_forward_backward(gpu_lib, smplh_npz, smplh_struct, N=N, device=dev, seed=N, hands=hands, selector=not hands, algo=2, dense_grad=True)
    finally:
        gpu_lib.call('ha_tune_set', b'dense_gA_sparse', 2)


@pytest.mark.parametrize('num_verts', [1101, 1100])
def test_dense_backward_small_models_odd_vertex_count(gpu_lib, dev, tmp_path, num_verts):
    """The chunk-compressed dL/dA kernel stages a chunk's 192 + 192 floats with 8-byte loads when the frame stride V * 3 floats is 8-byte
    aligned (V even) and with scalar loads otherwise; both against the oracle on models with a ragged last 64-vertex chunk."""
    import numpy as np
    from humor_amd import synth
    npz = synth.write_smplh_npz(str(tmp_path / f'model_{num_verts}.npz'), seed=1, num_verts=num_verts)
    data = np.load(npz)

    class DS:
        pass
    ds = DS()
    for k in data.files:
        setattr(ds, k, data[k])
    SC.check_forward_backward(gpu_lib, npz, ds, N=5, device=dev, hands=True, selector=False, algo=2, dense_grad=True)


def test_golden_vectors(gpu_lib, smplh_npz, dev):
    SC.check_golden(gpu_lib, smplh_npz, dev)


def test_full_size_properties(gpu_lib, smplh_npz, smplh_struct, dev):
    """BASELINE config C4 size (32 x 60 frames): size-independent properties instead of an oracle run."""
    from humor_amd.body_model import BodyModel
    N = 32 * 60
    inp = {k: v.detach() for k, v in SC.make_inputs(N, 3, dev).items()}
    bm1 = BodyModel(smplh_npz, num_betas=16, use_vtx_selector=True, algo=1)
    bm2 = BodyModel(smplh_npz, num_betas=16, use_vtx_selector=True, algo=2)
    o1, o2 = bm1(**inp), bm2(**inp)
    assert torch.isfinite(o2.v).all()
    # the two independent kernels (VALU wave-per-frame vs MFMA + streaming skinning) agree
    assert (o1.v - o2.v).abs().max().item() < 1e-4 and (o1.Jtr - o2.Jtr).abs().max().item() < 1e-5
    # subset evaluation == gather of the dense result
    bm3 = BodyModel(smplh_npz, num_betas=16, use_vtx_selector=True, vertex_subset=SC.KEYPT_VERTS)
    o3 = bm3(**inp)
    assert (o3.v - o2.v[:, SC.KEYPT_VERTS]).abs().max().item() < 1e-5
    # translation equivariance
    shift = torch.tensor([[0.5, -1.0, 2.0]], device=dev)
    o4 = bm2(**{**inp, 'trans': inp['trans'] + shift})
    assert (o4.v - (o2.v + shift[:, None])).abs().max().item() < 1e-5
    # rigid root rotation: |v - root joint| is invariant to root_orient
    o5 = bm2(**{**inp, 'root_orient': inp['root_orient'] + 0.3})
    d2 = (o2.v - o2.Jtr[:, :1]).norm(dim=2)
    d5 = (o5.v - o5.Jtr[:, :1]).norm(dim=2)
    assert (d2 - d5).abs().max().item() < 1e-4
    # zero pose/shape -> template + trans
    z = lambda d: torch.zeros(8, d, device=dev)
    o6 = bm2(root_orient=z(3), pose_body=z(63), betas=z(16), trans=inp['trans'][:8])
    vt = torch.tensor(smplh_struct.v_template, device=dev)
    assert (o6.v - (vt[None] + inp['trans'][:8, None])).abs().max().item() < 1e-5
    # spot-check 4 frames of the big batch against the oracle
    sel = [0, 777, 1500, N - 1]
    ref, _ = SC.oracle_forward(smplh_struct, {k: v[sel] for k, v in inp.items()}, True)
    assert (o2.v[sel].cpu() - ref.vertices).abs().max().item() < 1e-4


def test_errors_are_exceptions(gpu_lib, smplh_npz, dev):
    """Shape errors surface as Python exceptions (run_fitting.py:437-439 relies on catching them)."""
    import ctypes as C
    from humor_amd._lib import HumorAmdError
    with pytest.raises(HumorAmdError):
        gpu_lib.call('ha_smpl_forward', None, 0, 1, 22, None, None, None, None, None, None, None, None, 0, None)


def test_c5_size_dense_smpl(gpu_lib, smplh_npz, smplh_struct, dev):
    """BASELINE config C5 size (256 x 120 = 30 720 frames, 2.5 GB of vertices): finite, two kernels agree on a slice,
    oracle spot check."""
    from humor_amd.body_model import BodyModel
    N = 256 * 120
    inp = {k: v.detach() for k, v in SC.make_inputs(N, 5, dev).items()}
    o2 = BodyModel(smplh_npz, num_betas=16, use_vtx_selector=True, algo=2)(**inp)
    assert o2.v.shape == (N, 6890, 3)
    assert torch.isfinite(o2.v[::997]).all()
    sel = [0, 12345, N - 1]
    ref, _ = SC.oracle_forward(smplh_struct, {k: v[sel] for k, v in inp.items()}, True)
    assert (o2.v[sel].cpu() - ref.vertices).abs().max().item() < 1e-4
    assert (o2.Jtr[sel].cpu() - ref.joints).abs().max().item() < 1e-4
    o1 = BodyModel(smplh_npz, num_betas=16, use_vtx_selector=True, algo=1)(**{k: v[-64:] for k, v in inp.items()})
    assert (o1.v - o2.v[-64:]).abs().max().item() < 1e-4


def test_skin_kernel_variants(gpu_lib, smplh_npz, smplh_struct, dev):
    SC.check_skin_variants(gpu_lib, smplh_npz, smplh_struct, dev, N=5)


def test_chamfer_kernels(gpu_lib):
    """Chamfer nearest-neighbour search + gradient at the sizes of the point-cloud term (observed cloud vs 6890 SMPL vertices): int32
    indices and squared distances bit-exact against the oracle, ties included."""
    import chamfer_checks as CC
    dev = torch.device('cuda:0')
    for b, n, m, seed in ((2, 1024, 6890, 0), (3, 37, 1100, 1), (1, 4096, 6890, 2), (5, 300, 29, 3)):
        print('chamfer', b, n, m, CC.check_chamfer(gpu_lib, dev, b, n, m, seed))


@pytest.mark.gpu
def test_parts_api_equals_split_path(gpu_lib, dev, smplh_npz):
    """ha_smpl_forward_parts / _backward_parts / ha_seq_sum_add against the split path + autograd's additions, at a toy size and at the
    metric's batch (32 x 60)."""
    for B, T in ((2, 3), (32, 60)):
        print('parts api', B, T, SC.check_parts_api(gpu_lib, smplh_npz, dev, B=B, T=T, seed=B))
