"""CPU tier: the SMPL kernel sources executed by the host SIMT emulator (tests/simt_emu) against the oracle.
Tiny sizes (one OS thread per work-item); the real parity tests are tests/test_smpl_gpu.py."""
import pytest
import torch

import smpl_checks as SC

CPU = torch.device('cpu')


def test_emu_frame_kernel_dense(emu_lib, smplh_npz, smplh_struct):
    SC.check_forward_backward(emu_lib, smplh_npz, smplh_struct, N=5, device=CPU, algo=1)


def test_emu_subset_fwd_bwd(emu_lib, smplh_npz, smplh_struct):
    SC.check_forward_backward(emu_lib, smplh_npz, smplh_struct, N=6, device=CPU, subset=SC.KEYPT_VERTS)


def test_emu_subset_two_chunks_split_outputs(emu_lib, smplh_npz, smplh_struct):
    """A subset of 21 selector + 100 requested vertices = two 64-vertex chunks: the selector vertices leave with the joints, the rest as `v`
    (ha_smpl_forward_split / _backward_split), the block-wide blend exchange is reused between the chunks; and no selector at all (n_head = 0)."""
    sub = list(range(5, 6805, 68))          # 100 vertices spread over the mesh
    SC.check_forward_backward(emu_lib, smplh_npz, smplh_struct, N=3, device=CPU, subset=sub, seed=7)
    SC.check_forward_backward(emu_lib, smplh_npz, smplh_struct, N=2, device=CPU, subset=sub[:30], selector=False, seed=8)


def test_emu_parts_api(emu_lib, smplh_npz):
    """ha_smpl_forward_parts / _backward_parts / ha_seq_sum_add on the emulator (what the stage-3 composite nodes call)."""
    SC.check_parts_api(emu_lib, smplh_npz, CPU, B=2, T=2)


def test_emu_hands_dense_grad(emu_lib, smplh_npz, smplh_struct):
    SC.check_forward_backward(emu_lib, smplh_npz, smplh_struct, N=3, device=CPU, hands=True, selector=False, algo=1, dense_grad=True)


def test_emu_golden(emu_lib, smplh_npz):
    SC.check_golden(emu_lib, smplh_npz, CPU)


@pytest.mark.slow
def test_emu_mfma_dense_path(emu_lib, smplh_npz, smplh_struct):
    # MFMA fragment maps + streaming skinning kernel (window straddling two frames, ragged tail)
    SC.check_forward_backward(emu_lib, smplh_npz, smplh_struct, N=3, device=CPU, algo=2)


def test_no_cpu_fallback(smplh_npz):
    """The product BodyModel must refuse CPU tensors instead of silently computing somewhere else."""
    from humor_amd.body_model import BodyModel
    from humor_amd._lib import HumorAmdError
    bm = BodyModel(smplh_npz, num_betas=16)
    with pytest.raises((HumorAmdError, RuntimeError)):
        bm(root_orient=torch.zeros(1, 3), pose_body=torch.zeros(1, 63), betas=torch.zeros(1, 16), trans=torch.zeros(1, 3))


@pytest.mark.slow          # (two minutes on the emulator: 216 vertex tiles of MFMAs; the GPU tier checks the same bit-equality at five sizes)
def test_emu_fused_blend_skin_forward(emu_lib, smplh_npz, smplh_struct):
    """ha_smpl_forward algo 3 (forward-only dense calls) on the emulator: bit-identical to blend + lbs_skin; a frame count that is not a
    multiple of the 32-frame tile."""
    print('fused dense forward vs oracle', SC.check_fused_forward(emu_lib, smplh_npz, smplh_struct, N=3, device=CPU))


def test_emu_skin_kernel_variants(emu_lib, smplh_npz, smplh_struct):
    SC.check_skin_variants(emu_lib, smplh_npz, smplh_struct, torch.device('cpu'), N=2, variants=(5, 4, 2))


def test_emu_chamfer_kernels(emu_lib):
    """chamfer.hip on the SIMT emulator: indices / distances bit-exact vs the oracle incl. ties; a cloud larger than one LDS chunk."""
    import chamfer_checks as CC
    CC.check_chamfer(emu_lib, torch.device('cpu'), b=2, n=37, m=1100, seed=0)
    CC.check_chamfer(emu_lib, torch.device('cpu'), b=1, n=300, m=29, seed=1)


def _small_model(tmp_path, num_verts):
    import numpy as np
    from humor_amd import synth
    npz = synth.write_smplh_npz(str(tmp_path / f'model_{num_verts}.npz'), seed=1, num_verts=num_verts)
    data = np.load(npz)

    class DS:
        pass
    ds = DS()
    for k in data.files:
        setattr(ds, k, data[k])
    return npz, ds


@pytest.mark.slow          # (a minute and a half on the emulator; the GPU tier runs the dense backward at five sizes against the oracle)
def test_emu_dense_backward_kernels(emu_lib, tmp_path):
    """ha_smpl_backward_dense (streaming dL/dv_posed, 16x16x4 MFMA dL/dA, 32x32x2 MFMA dL/dcoeff with K split, chain adjoint) behind the
    MFMA forward: every vertex carries a gradient.  1100-vertex model (ragged last chunk), hands on (52 active joints, 476 blend
    coefficients = four 128-column groups), then body only (206 coefficients) with a coarser K split."""
    npz, ds = _small_model(tmp_path, 1100)
    SC.check_forward_backward(emu_lib, npz, ds, N=3, device=CPU, hands=True, selector=False, algo=2, dense_grad=True)
    emu_lib.call('ha_tune_set', b'dense_bwd_waves', 10)
    try:
        SC.check_forward_backward(emu_lib, npz, ds, N=2, device=CPU, hands=False, selector=False, algo=2, dense_grad=True, seed=3)
    finally:
        emu_lib.call('ha_tune_set', b'dense_bwd_waves', 0)


@pytest.mark.slow
@pytest.mark.parametrize('variant', [0, 1])
def test_emu_dense_backward_dA_variants(emu_lib, tmp_path, variant):
    """dL/dA of the dense backward behind ha_tune_set("dense_gA_sparse"): the dense 64-column MFMA product (0) and the joint lists (1); the
    default (2: the chunk-local compressed product -- the 1100-vertex model has chunks with one and with two 16-slot groups) is
    test_emu_dense_backward_kernels."""
    npz, ds = _small_model(tmp_path, 1100)
    emu_lib.call('ha_tune_set', b'dense_gA_sparse', variant)
    try:
        SC.check_forward_backward(emu_lib, npz, ds, N=2, device=CPU, hands=True, selector=False, algo=2, dense_grad=True, seed=4 + variant)
    finally:
        emu_lib.call('ha_tune_set', b'dense_gA_sparse', 2)


@pytest.mark.slow
def test_emu_dense_backward_full_model(emu_lib, smplh_npz, smplh_struct):
    SC.check_forward_backward(emu_lib, smplh_npz, smplh_struct, N=3, device=CPU, hands=True, selector=False, algo=2, dense_grad=True)
