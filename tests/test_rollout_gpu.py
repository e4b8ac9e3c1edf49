"""GPU tier (-m gpu): HuMoR roll-out + rotation kernels on a real MI355X through the C ABI."""
import pytest
import torch

import mlp_checks as MC
import rollout_checks as RC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


@pytest.fixture(params=[(1, 1), (1, 0), (0, 0)], ids=['persistent', 'persistent_fwd_chain_bwd', 'chain'])
def fwd_path(request, gpu_lib):
    """The paths of a roll-out: one persistent launch per direction (default: the weight-stationary kernels of rollout_persist.hip for
    <= 32 sequences, the layer-parallel pipelined kernels of rollout_pipe.inc for 33 .. 256, chunks of 256 beyond), persistent forward
    with the launch-chain adjoint reading its stash, and the launch chain in both directions."""
    gpu_lib.call('ha_tune_set', b'rollout_persist', request.param[0])
    gpu_lib.call('ha_tune_set', b'rollout_persist_bwd', request.param[1])
    yield request.param
    gpu_lib.call('ha_tune_set', b'rollout_persist', 1)
    gpu_lib.call('ha_tune_set', b'rollout_persist_bwd', 1)


@pytest.fixture
def chain_only(gpu_lib):
    gpu_lib.call('ha_tune_set', b'rollout_persist', 0)
    yield
    gpu_lib.call('ha_tune_set', b'rollout_persist', 1)


@pytest.mark.parametrize('B,S', [(1, 1), (2, 3), (4, 10), (32, 12), (33, 5), (70, 3), (130, 2)])   # 130 rows: full-K launch policy
def test_rollout_forward_backward(gpu_lib, dev, fwd_path, B, S):
    """Flat bars (1e-4 on every state / prior output, 1e-3 of the largest entry on every gradient) on the random network.  A sequence whose
    gradient misses the bar on the oracle's own ReLU branch must equal the fp64 oracle on a branch the reference side cannot tell apart
    (RC.kink_aware_grad_check: units within the kernel's forward deviation of their kink, decided from the oracle, not from the kernel)."""
    RC.check_rollout(gpu_lib, dev, B=B, S=S, seed=B)


@pytest.mark.parametrize('B,S', [(1, 1), (2, 3), (4, 10), (32, 12), (33, 5), (70, 3), (130, 2), (260, 2), (288, 2)])
def test_rollout_forward_backward_contractive(gpu_lib, dev, fwd_path, B, S):
    """The same grid on the well-conditioned (contractive) network, where a chain of steps does not amplify rounding: tight bars (2e-5 / 3e-4)
    catch rounding-level regressions the random network's flat bars would let through.  260 / 288 rows: a pipelined chunk of 256 plus a tail
    that goes to the B <= 32 kernels (4 rows) / to the pipelined kernels (32 rows)."""
    RC.check_rollout(gpu_lib, dev, B=B, S=S, seed=B, contractive=True, fwd_tol=2e-5, grad_rtol=3e-4)


@pytest.mark.parametrize('B', [1, 2, 3, 4, 5, 6, 7, 8])
def test_rollout_small_batches_on_poisoned_state(gpu_lib, dev, fwd_path, B):
    """Round-5 regression (NaN gradients at 4 x 6 on a fresh box, nothing on the builder's): 1 .. 8 sequences x 1 .. 8 steps on every path, every
    allocation NaN-filled (conftest) and every CU's LDS / registers NaN-filled in front of each persistent launch (cu_poison), against the oracle
    at the flat bars.  The partial-team sizes (B not a multiple of 4) and the odd / even step counts cover both LDS state buffers of the kernels."""
    import ctypes as C
    n = C.c_uint()
    gpu_lib.call('ha_debug_cu_poison', 1, C.byref(n), None)
    assert n.value == 256 * 40960, f'only {n.value} of {256 * 40960} LDS words hold the pattern in the next kernel: the poison hook does not cover this box'
    import os
    pattern = int(os.environ.get('HUMOR_AMD_CU_POISON') or '1', 0)
    gpu_lib.call('ha_tune_set', b'cu_poison', pattern)
    try:
        for S in range(1, 9):
            RC.check_rollout(gpu_lib, dev, B=B, S=S, seed=8 * B + S)
    finally:
        gpu_lib.call('ha_tune_set', b'cu_poison', pattern if os.environ.get('HUMOR_AMD_CU_POISON') else 0)


def test_rollout_without_prior(gpu_lib, dev, fwd_path):
    RC.check_rollout(gpu_lib, dev, B=5, S=4, with_prior=False)


def test_rollout_golden(gpu_lib, dev, fwd_path):
    RC.check_rollout_golden(gpu_lib, dev)


@pytest.mark.parametrize('rep', ['6d', '9d', 'nd'])
def test_rollout_output_rotation_representations(gpu_lib, dev, rep):
    """HumorModel(out_rot_rep='6d' / '9d') (humor_model.py:476-484) and HumorModel(output_delta=False) ('nd', :331-347) against the
    reference-generated fixture: flat 1e-4 / 1e-3 bars."""
    print(rep, RC.check_rollout_rotrep_golden(gpu_lib, dev, rep))


@pytest.mark.parametrize('rep,steps_in', RC.INREP_CASES)
def test_rollout_input_variants(gpu_lib, dev, rep, steps_in):
    """HumorModel(in_rot_rep='aa' | '6d', steps_in=2) on the device (prior through the fused MLP kernels, R -> axis-angle through its HIP kernel)
    against the reference-generated fixture: flat 1e-4 / 1e-3 bars."""
    print(rep, steps_in, RC.check_rollout_inrep(gpu_lib, dev, rep, steps_in))


@pytest.mark.parametrize('canon,uncanon', [(False, False), (True, False), (True, True)])
def test_generic_step_loop_equals_the_kernel_path(gpu_lib, dev, canon, uncanon):
    """The on-device step loop that serves HumorModel's input variants (_roll_out_generic) is a second implementation of the whole roll-out: in
    the fitting configuration ('mat' / 1), where the HIP kernels run, both must agree -- world states, prior outputs, gradients -- with and
    without canonicalize_input / uncanonicalize_output (humor_model.py:808-858)."""
    hm, _ = RC.make_model(gpu_lib, dev, seed=9, contractive=True)
    for p in hm.parameters():
        p.requires_grad_(False)
    g = torch.Generator().manual_seed(19)
    B, S = 5, 6
    past = RC.canonical_state(B, g)
    if canon:                         # a world-frame start: the canonical state turned about z and moved
        th = torch.rand(B, generator=g) * 2.0
        Rz = torch.zeros(B, 3, 3)
        Rz[:, 0, 0], Rz[:, 0, 1], Rz[:, 1, 0], Rz[:, 1, 1], Rz[:, 2, 2] = th.cos(), -th.sin(), th.sin(), th.cos(), 1.0
        off = torch.cat([torch.randn(B, 2, generator=g), torch.zeros(B, 1)], 1)
        rot = lambda v: torch.einsum('bij,bj->bi', Rz, v)
        joints = torch.einsum('bij,bkj->bki', Rz, past[:, 207:273].reshape(B, 22, 3)) + off.unsqueeze(1)
        past = torch.cat([rot(past[:, 0:3]) + off, rot(past[:, 3:6]), torch.matmul(Rz, past[:, 6:15].reshape(B, 3, 3)).reshape(B, 9), rot(past[:, 15:18]),
                          past[:, 18:207], joints.reshape(B, 66), torch.einsum('bij,bkj->bki', Rz, past[:, 273:339].reshape(B, 22, 3)).reshape(B, 66)], 1)
    past, z = past.to(dev), (0.5 * torch.randn(B, S, 48, generator=g)).to(dev)
    gw, gm = torch.randn(B, S, 348, generator=g).to(dev), torch.randn(B, S, 48, generator=g).to(dev)
    res = []
    for generic in (False, True):
        p, zz = past.clone().requires_grad_(True), z.clone().requires_grad_(True)
        x_past = p.reshape(B, 1, 339)
        if generic:
            out, (pm, pv) = hm._roll_out_generic(x_past, None, S, False, zz, True, False, canon, uncanon, None)
        else:
            out, (pm, pv) = hm.roll_out(x_past, None, S, z_seq=zz, return_prior=True, canonicalize_input=canon, uncanonicalize_output=uncanon)
        w = RC.world_of(out)
        ((w * gw).sum() + (pm * gm).sum() + pv.sum()).backward()
        res.append((w.detach(), pm.detach(), pv.detach(), p.grad.clone(), zz.grad.clone()))
    for name, a, b, tol in zip(('world', 'pm', 'pv', 'g_past', 'g_z'), res[0], res[1], (2e-5, 2e-5, 2e-5, 3e-4, 3e-4)):
        e = (a - b).abs().max().item() / max(1.0, a.abs().max().item())
        assert e <= tol, (name, e)


@pytest.mark.parametrize('name', ['c4', 'c3', 'c5'])
def test_rollout_baseline_lengths_flat_tolerance(gpu_lib, dev, fwd_path, name):
    """59 / 89 / 119 steps against the reference's own outputs (tests/golden/rollout_long.npz): flat 1e-4 on every step's
    state and prior output, 1e-3 relative on the gradients -- no conditioning allowance (well-conditioned synthetic prior)."""
    print(name, RC.check_rollout_long(gpu_lib, dev, name))


@pytest.mark.parametrize('B,S', [(32, 59), (256, 119)])
def test_rollout_full_tiles_flat_tolerance(gpu_lib, dev, fwd_path, B, S):
    """The metric's batch (one FULL 32-row tile, 59 steps) and the C5 batch (eight tiles, 119 steps) at the flat 1e-4 / 1e-3 bars."""
    print(B, S, RC.check_rollout_full_tiles(gpu_lib, dev, B, S))


def test_rollout_full_length(gpu_lib, dev, fwd_path):
    """BASELINE size (32 sequences x 59 steps): conditioning-aware parity (see check_rollout_conditioned) plus the
    gradient check on an 8-step chain judged against an fp64 oracle.  (The adjoint is amplified like the forward error:
    at 12-16 steps two fp32 implementations already differ by 1-10 % of a gradient whose scale itself varies over three
    decades between seeds -- tests/diagnostics/grad_accuracy.py tabulates GPU fp32 vs CPU fp32 vs fp64 by chain length.)"""
    RC.check_rollout_conditioned(gpu_lib, dev, B=32, S=59, seed=3)
    RC.check_rollout(gpu_lib, dev, B=32, S=8, seed=4, fwd_tol=1e-3, grad_rtol=1e-2, cond_aware=True)


def test_rollout_determinism(gpu_lib, dev, fwd_path):
    hm, _ = RC.make_model(gpu_lib, dev)
    g = torch.Generator().manual_seed(1)
    past = RC.canonical_state(8, g).to(dev)
    z = torch.randn(8, 20, 48, generator=g).to(dev)
    a = RC.world_of(hm.roll_out(past, None, 20, z_seq=z))
    b = RC.world_of(hm.roll_out(past, None, 20, z_seq=z))
    assert torch.equal(a, b)


def test_rollout_accumulate_policy(gpu_lib, dev, chain_only):
    """ha_tune_set("layer_acc", 1) (fp32-atomic accumulation of the K-split partial tiles, off by default) against the fixed-order
    partial-slab path: same values and gradients up to the summation order of <= 5 partials per element."""
    hm, _ = RC.make_model(gpu_lib, dev, contractive=True)
    g = torch.Generator().manual_seed(2)
    past = RC.canonical_state(8, g).to(dev)
    z = torch.randn(8, 20, 48, generator=g).to(dev)
    res = []
    try:
        for acc in (0, 1, 1):
            gpu_lib.call('ha_tune_set', b'layer_acc', acc)
            p, zz = past.clone().requires_grad_(True), z.clone().requires_grad_(True)
            out, (pm, pv) = hm.roll_out(p, None, 20, z_seq=zz, return_prior=True)
            (out['joints'].square().sum() + pm.sum() + pv.sum()).backward()
            res.append((RC.world_of(out).detach(), p.grad.clone(), zz.grad.clone()))
    finally:
        gpu_lib.call('ha_tune_set', b'layer_acc', 0)
    for k in (1, 2):
        for a, b in zip(res[0], res[k]):
            assert (a - b).abs().max().item() <= 1e-5 * max(1.0, a.abs().max().item())


@pytest.mark.parametrize('B,groups', [(70, 2), (70, 3), (130, 4)])
def test_rollout_row_groups(gpu_lib, dev, B, groups):
    """ha_tune_set("rollout_groups", n): the batch split into row groups on side streams (own stash regions, fork / join on the
    caller's stream) against the single-chain evaluation -- values, prior outputs and every gradient."""
    hm, _ = RC.make_model(gpu_lib, dev, contractive=True)
    g = torch.Generator().manual_seed(B)
    past = RC.canonical_state(B, g).to(dev)
    z = torch.randn(B, 12, 48, generator=g).to(dev)
    res = []
    gpu_lib.call('ha_tune_set', b'rollout_pipe', 0)          # (row groups on side streams are a launch-chain policy)
    try:
        for n in (1, groups):
            gpu_lib.call('ha_tune_set', b'rollout_groups', n)
            p, zz = past.clone().requires_grad_(True), z.clone().requires_grad_(True)
            out, (pm, pv) = hm.roll_out(p, None, 12, z_seq=zz, return_prior=True)
            w = RC.world_of(out)
            (w.square().sum() + pm.square().sum() + pv.sum()).backward()
            res.append((w.detach(), pm.detach(), pv.detach(), p.grad.clone(), zz.grad.clone()))
    finally:
        gpu_lib.call('ha_tune_set', b'rollout_groups', 0)
        gpu_lib.call('ha_tune_set', b'rollout_pipe', 1)
    for a, b in zip(*res):
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, a.abs().max().item())


def test_rotation_kernels(gpu_lib, dev):
    RC.check_rot_random(gpu_lib, dev, n=100000)
    RC.check_rotations_golden(gpu_lib, dev)
    RC.check_rot6d(gpu_lib, dev, n=50000)
    print('rot9d gradient: worst relative error', RC.check_rot9d(gpu_lib, dev, n=30000))
    print('R->aa gradient near pi: worst relative error', RC.check_rot_to_aa_near_pi(gpu_lib, dev, n=50000))


def test_c5_size_rollout(gpu_lib, dev):
    """BASELINE config C5 batch (256 sequences = 8 row tiles), 119 steps: finite and bit-equal, sequence by sequence, to the
    same sequences rolled out in a batch of 128 (row tiles are independent; same full-K launch policy), and equal within
    fp32 rounding over the first steps to a batch of 32 (split-K policy: another summation order)."""
    hm, _ = RC.make_model(gpu_lib, dev)
    g = torch.Generator().manual_seed(2)
    past = RC.canonical_state(256, g).to(dev)
    z = (0.5 * torch.randn(256, 119, 48, generator=g)).to(dev)
    big = RC.world_of(hm.roll_out(past, None, 119, z_seq=z))
    assert torch.isfinite(big).all()
    mid = RC.world_of(hm.roll_out(past[64:192], None, 119, z_seq=z[64:192]))
    assert torch.equal(big[64:192], mid)
    small = RC.world_of(hm.roll_out(past[64:96], None, 8, z_seq=z[64:96, :8]))
    assert (big[64:96, :8] - small).abs().max().item() < 1e-4


def test_sampling_and_canonicalize(gpu_lib, dev):
    RC.check_sampling_rollout(gpu_lib, dev, B=2, S=30)


def test_sampling_golden_vectors(gpu_lib, dev):
    RC.check_sampling_golden(gpu_lib, dev)


@pytest.mark.parametrize('N', [1, 33, 1920])
def test_fused_vposer_matches_module(gpu_lib, dev, N):
    """VPoser decode (+ 6-D -> R -> axis-angle) / encode through ha_mlp_* vs the PyTorch module + oracle R -> aa, forward and
    gradients (motion_optimizer.py:1041-1063), for the stand-in and for a module with VPoser v1.0's member names (BatchNorm folded)."""
    MC.check_vposer(gpu_lib, dev, N=N)
    MC.check_vposer(gpu_lib, dev, N=N, real_shaped=True, seed=3)


@pytest.mark.parametrize('ks', [0, 2])
def test_batched_gemm_k_split_policy(gpu_lib, dev, ks):
    """ha_tune_set("gemm_ks", 2) (default: K split over two waves per tile pair for GEMMs that leave most SIMDs idle) and 0 (off): the
    LeakyReLU (VPoser) and GroupNorm (posterior encoder, small-batch prior) epilogues behind both."""
    gpu_lib.call('ha_tune_set', b'gemm_ks', ks)
    try:
        MC.check_vposer(gpu_lib, dev, N=1920, real_shaped=True, seed=5)
        MC.check_vposer(gpu_lib, dev, N=33, real_shaped=True, seed=6)
        MC.check_posterior(gpu_lib, dev, N=40)
        RC.check_rollout(gpu_lib, dev, B=4, S=6, seed=11)
    finally:
        gpu_lib.call('ha_tune_set', b'gemm_ks', 2)


def test_fused_posterior_encoder_matches_module(gpu_lib, dev):
    MC.check_posterior(gpu_lib, dev, N=32 * 59)
    MC.check_posterior(gpu_lib, dev, N=5)


def test_infer_global_seq_on_device_matches_reference_fixture(gpu_lib, dev):
    """SURVEY 8(a) a15 on the GPU: infer_global_seq (+ the stage-3 velocity estimators) on device tensors -- the fused prior / posterior
    MLPs (ha_mlp_*), not the module branch -- against the reference-generated tests/golden/infer_global_seq.npz (2 x 9 and 4 x 60)."""
    print('infer_global_seq vs reference fixture: worst relative deviation', MC.check_infer_global_seq_golden(gpu_lib, dev))


def test_posterior_gives_parameter_gradients(gpu_lib, dev):
    """a training-mode infer_step must leave gradients on the encoder / prior weights (module forward), frozen nets run fused."""
    MC.check_posterior_param_grads(gpu_lib, dev, N=9)


@pytest.mark.parametrize('variant', [1, 3])
def test_persistent_forward_matches_launch_chain(gpu_lib, dev, variant):
    """ha_tune_set("rollout_persist"): ONE persistent launch for the whole decoder chain (register-stationary weights, XCD teams of
    4 sequences; variant 3 publishes write-through) against the 5-launches-per-step chain -- outputs and gradients."""
    for B, S in ((32, 59), (5, 7), (1, 3), (32, 1), (17, 20), (4, 119)):
        print('persistent vs chain', B, S, variant, RC.check_persistent_vs_chain(gpu_lib, dev, B, S, seed=B, variant=variant))


@pytest.mark.parametrize('B,S', [(64, 12), (40, 5), (100, 7), (256, 20), (300, 4), (260, 3), (288, 3)])
def test_pipelined_rollout_matches_launch_chain(gpu_lib, dev, B, S):
    """roll-outs of more than 32 sequences on the layer-parallel pipelined persistent kernels (rollout_pipe.inc: forward and
    adjoint, one launch each per chunk of <= 256 sequences) against the 5-launches-per-step chain on the same inputs -- world states,
    prior outputs and every gradient, for the pipelined forward + launch-chain adjoint and for pipelined forward + pipelined adjoint;
    partial last tiles (40, 100), full tiles (64, 256), two chunks (300), and a chunk of 256 whose tail of 4 / 32 sequences goes to the B <= 32
    kernels from inside the chunk loop (260, 288: ADVICE r5)."""
    print('pipelined vs chain', B, S, RC.check_persistent_vs_chain(gpu_lib, dev, B, S, seed=B, variant=1, launches_per_call=(B + 255) // 256))


@pytest.mark.parametrize('B', [260, 288, 100])
def test_pipelined_forward_with_the_launch_chain_adjoint_knob(gpu_lib, dev, B):
    """ha_tune_set("rollout_pipe_bwd", 0) -- pipelined forward (hidden slabs written), launch-chain adjoint for the pipelined chunks; with 260 /
    288 sequences the tail of 4 / 32 goes to the B <= 32 kernels, whose one-launch adjoint reads a stash that carries slabs (mode 1).  Against the launch
    chain in both directions."""
    hm, _ = RC.make_model(gpu_lib, dev, seed=B, contractive=True)
    g = torch.Generator().manual_seed(B)
    S = 3
    past, z = RC.canonical_state(B, g).to(dev), torch.randn(B, S, 48, generator=g).to(dev)
    gw = torch.randn(B, S, 348, generator=g).to(dev)
    res = []
    try:
        for persist, pipe_bwd in ((0, 1), (1, 0)):
            gpu_lib.call('ha_tune_set', b'rollout_persist', persist)
            gpu_lib.call('ha_tune_set', b'rollout_pipe_bwd', pipe_bwd)
            p, zz = past.clone().requires_grad_(True), z.clone().requires_grad_(True)
            out, (pm, pv) = hm.roll_out(p, None, S, z_seq=zz, return_prior=True)
            w = RC.world_of(out)
            ((w * gw).sum() + pm.sum() + pv.sum()).backward()
            res.append((w.detach(), pm.detach(), p.grad.clone(), zz.grad.clone()))
    finally:
        gpu_lib.call('ha_tune_set', b'rollout_persist', 1)
        gpu_lib.call('ha_tune_set', b'rollout_pipe_bwd', 1)
    for name, a, b, tol in zip(('world', 'pm', 'g_past', 'g_z'), res[0], res[1], (2e-5, 2e-5, 3e-4, 3e-4)):
        e = (a - b).abs().reshape(B, -1).amax(1) / max(1.0, a.abs().max().item())
        assert torch.isfinite(b).all(), name
        # (per sequence; at most one sequence on another ReLU branch per case: the kink bar of the pipelined-vs-chain test)
        assert (e > tol).sum().item() <= 1 and e.max().item() <= RC.KINK_RTOL, (name, e.max().item(), int((e > tol).sum()))


def test_pipelined_rollout_determinism_and_reuse(gpu_lib, dev):
    """Back-to-back pipelined launches on one network (fresh stash each): bit-identical results in both directions, no stale granules."""
    hm, _ = RC.make_model(gpu_lib, dev, contractive=True)
    g = torch.Generator().manual_seed(6)
    past, z = RC.canonical_state(96, g).to(dev), torch.randn(96, 30, 48, generator=g).to(dev)
    outs = []
    for _ in range(5):
        p, zz = past.clone().requires_grad_(True), z.clone().requires_grad_(True)
        out, (pm, pv) = hm.roll_out(p, None, 30, z_seq=zz, return_prior=True)
        w = RC.world_of(out)
        (w.square().sum() + pm.sum() + pv.square().sum()).backward()
        outs.append((w.detach(), p.grad.clone(), zz.grad.clone()))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)
    av, err, n = RC.persist_status(gpu_lib, hm, dev)
    assert av == 1 and err == 0 and (n & 0xffffffff) >= 5 and (n >> 32) >= 5, (av, err, n)


def test_persistent_forward_determinism_and_reuse(gpu_lib, dev):
    """Back-to-back persistent launches on the same network (fresh stash each): bit-identical results, no stale granules."""
    hm, _ = RC.make_model(gpu_lib, dev, contractive=True)
    g = torch.Generator().manual_seed(5)
    past, z = RC.canonical_state(32, g).to(dev), torch.randn(32, 59, 48, generator=g).to(dev)
    gpu_lib.call('ha_tune_set', b'rollout_persist', 1)
    try:
        outs = [RC.world_of(hm.roll_out(past, None, 59, z_seq=z)) for _ in range(6)]
    finally:
        gpu_lib.call('ha_tune_set', b'rollout_persist', 1)
    for o in outs[1:]:
        assert torch.equal(outs[0], o)
    assert RC.persist_status(gpu_lib, hm, dev)[1] == 0


def test_stash_mode_travels_with_the_stash(gpu_lib, dev):
    """The roll-out mode of a forward call (launch chain / one-launch with launch-chain slabs / one-launch without) is recorded for the stash
    it fills, together with its shape; the backward over that stash follows the RECORD, not the knobs of the moment: three forwards in three
    modes on addresses the allocator has recycled from stashes of other modes, knobs changed again, backwards out of order -- every
    gradient equals the launch chain's."""
    hm, _ = RC.make_model(gpu_lib, dev, seed=2, contractive=True)
    g = torch.Generator().manual_seed(21)
    B, S = 8, 5
    past, z = RC.canonical_state(B, g).to(dev), torch.randn(B, S, 48, generator=g).to(dev)
    gw = torch.randn(B, S, 348, generator=g).to(dev)

    def knobs(fwd, bwd):
        gpu_lib.call('ha_tune_set', b'rollout_persist', fwd)
        gpu_lib.call('ha_tune_set', b'rollout_persist_bwd', bwd)

    def forward():
        p, zz = past.clone().requires_grad_(True), z.clone().requires_grad_(True)
        out, (pm, pv) = hm.roll_out(p, None, S, z_seq=zz, return_prior=True)
        return p, zz, (RC.world_of(out) * gw).sum() + pm.sum() + pv.sum()
    try:
        knobs(0, 0)
        p, zz, loss = forward()
        loss.backward()
        ref = (p.grad.clone(), zz.grad.clone())
        for fwd, bwd in ((1, 1), (0, 0), (1, 0)):          # stashes of every mode, freed: their addresses come back below in another order
            knobs(fwd, bwd)
            forward()[2].backward()
        runs = []
        for fwd, bwd in ((1, 0), (1, 1), (0, 0)):          # mode 1 (slabs), mode 2 (no slabs), mode 0 (chain): all three stashes alive at once
            knobs(fwd, bwd)
            runs.append(forward())
        n0 = RC.persist_status(gpu_lib, hm, dev)[2] >> 32
        knobs(1, 0)
        runs[1][2].backward()                               # mode-2 stash: the one-launch adjoint although the knob says launch chain
        assert (RC.persist_status(gpu_lib, hm, dev)[2] >> 32) == n0 + 1
        knobs(0, 0)
        runs[0][2].backward()                               # mode-1 stash with the adjoint knob off: launch-chain adjoint over its slabs
        knobs(1, 1)
        runs[2][2].backward()                               # launch-chain stash: launch-chain adjoint whatever the knobs say
        assert (RC.persist_status(gpu_lib, hm, dev)[2] >> 32) == n0 + 1
    finally:
        knobs(1, 1)
    for p, zz, _ in runs:
        for a, b in zip((p.grad, zz.grad), ref):
            assert (a - b).abs().max().item() <= 3e-5 * max(1.0, b.abs().max().item())


def test_persistent_failure_poisons_results_and_is_reported(gpu_lib, dev):
    """a persistent launch whose team does not complete must not hand garbage to a caller that never passes an entry
    point (a captured hipGraph).  Test hook ha_tune_set("rollout_persist_inject"): one CU of team 0 leaves at once, the team's bounded
    waits run out.  Then: the team's output rows are NaN (the loss computed from them is NaN), the host-mapped error word is set, the next
    entry point returns the error once, and from then on the launch chain serves the network with correct results."""
    from humor_amd import _lib
    hm, _ = RC.make_model(gpu_lib, dev, seed=3, contractive=True)          # its own network handle (the failure is sticky per handle)
    g = torch.Generator().manual_seed(11)
    B, S = 12, 4
    past, z = RC.canonical_state(B, g).to(dev), torch.randn(B, S, 48, generator=g).to(dev)
    with torch.no_grad():
        ref = RC.world_of(hm.roll_out(past, None, S, z_seq=z))                # healthy persistent launch
        assert hm.persistent_rollout_status(dev)[:2] == (1, 0)
        gpu_lib.call('ha_tune_set', b'rollout_persist_inject', 1)
        try:
            bad = RC.world_of(hm.roll_out(past, None, S, z_seq=z))
            torch.cuda.synchronize()
        finally:
            gpu_lib.call('ha_tune_set', b'rollout_persist_inject', 0)
        assert torch.isnan(bad[:4]).all(), 'the incomplete team\'s rows must be NaN'
        assert torch.equal(bad[4:], ref[4:]), 'the other teams are unaffected'
        av, err, _ = hm.persistent_rollout_status(dev)
        assert av == 0 and (err & 0xf00) == 0x200, (av, hex(err))
        with pytest.raises(_lib.HumorAmdError):
            hm.roll_out(past, None, S, z_seq=z)                               # reported once
        again = RC.world_of(hm.roll_out(past, None, S, z_seq=z))             # the launch chain from now on
        assert torch.isfinite(again).all() and (again - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


def test_pipelined_failure_poisons_results_and_is_reported(gpu_lib, dev):
    """The failure protocol of the B <= 32 kernels on the pipelined ones (rollout_pipe.inc): with ha_tune_set("rollout_persist_inject") one
    CU of team 0 (a layer-0 CU) leaves at once, the team's bounded waits run out: every row the team owns -- rows 0..3 of EVERY 32-row tile --
    is NaN, the other teams' rows are untouched, the host-mapped error word is set, the next entry point returns the error once, and from
    then on the launch chain serves the network with correct results."""
    from humor_amd import _lib
    hm, _ = RC.make_model(gpu_lib, dev, seed=4, contractive=True)          # its own network handle (the failure is sticky per handle)
    g = torch.Generator().manual_seed(12)
    B, S = 70, 3
    past, z = RC.canonical_state(B, g).to(dev), torch.randn(B, S, 48, generator=g).to(dev)
    with torch.no_grad():
        ref = RC.world_of(hm.roll_out(past, None, S, z_seq=z))                # healthy pipelined launch
        assert hm.persistent_rollout_status(dev)[:2] == (1, 0)
        gpu_lib.call('ha_tune_set', b'rollout_persist_inject', 1)
        try:
            bad = RC.world_of(hm.roll_out(past, None, S, z_seq=z))
            torch.cuda.synchronize()
        finally:
            gpu_lib.call('ha_tune_set', b'rollout_persist_inject', 0)
        team0 = torch.zeros(B, dtype=torch.bool)
        for tile in range(3):
            team0[32 * tile: min(32 * tile + 4, B)] = True
        assert torch.isnan(bad[team0]).all(), 'the incomplete team\'s rows must be NaN'
        assert torch.equal(bad[~team0], ref[~team0]), 'the other teams are unaffected'
        av, err, _ = hm.persistent_rollout_status(dev)
        assert av == 0 and (err & 0xf00) == 0x600, (av, hex(err))
        with pytest.raises(_lib.HumorAmdError):
            hm.roll_out(past, None, S, z_seq=z)                               # reported once
        again = RC.world_of(hm.roll_out(past, None, S, z_seq=z))             # the launch chain from now on
        assert torch.isfinite(again).all() and (again - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
