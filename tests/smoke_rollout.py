"""Second half of __graft_entry__.smoke(): one tiny HuMoR roll-out (forward + backward) on the GPU against the oracle.
(Test infrastructure: lives under tests/ because it imports oracle/.)"""
import torch


def run(dev):
    from humor_amd import synth
    from humor_amd.humor_model import HumorModel
    from oracle import humor_restated as H
    from oracle.make_golden import canonical_state
    sd = synth.humor_state_dict(seed=0)
    hm = HumorModel(in_rot_rep='mat', out_rot_rep='aa', latent_size=48, model_data_config='smpl+joints+contacts', steps_in=1)
    hm.load_state_dict(sd)
    hm = hm.to(dev).eval()
    g = torch.Generator().manual_seed(0)
    B, S = 4, 6
    past_c = canonical_state(B, g).requires_grad_(True)
    z_c = torch.randn(B, S, 48, generator=g).requires_grad_(True)
    past, z = past_c.detach().to(dev).requires_grad_(True), z_c.detach().to(dev).requires_grad_(True)
    # every CU's LDS and vector registers hold NaN patterns when the persistent launches start (humor_amd/csrc/debug.hip): a read of state the
    # kernels did not write shows on every box, not only on the one whose previous tenant left NaNs behind (round 5)
    from humor_amd import _lib
    _lib.get_lib().call('ha_tune_set', b'cu_poison', 1)
    out, (pm, pv) = hm.roll_out(past, None, S, z_seq=z, return_prior=True)
    world = torch.cat([out[k] for k in ('trans', 'trans_vel', 'root_orient', 'root_orient_vel', 'pose_body', 'joints',
                                         'joints_vel', 'contacts')], 2)
    w_ref, (pm_r, _) = H.roll_out(sd, past_c, z_c)
    err = (world.detach().cpu() - w_ref).abs().max().item()
    (world.square().sum() + pm.sum()).backward()
    (w_ref.square().sum() + pm_r.sum()).backward()
    gerr = max((a.grad.cpu() - b.grad).abs().max().item() / max(1.0, b.grad.abs().max().item()) for a, b in ((z, z_c), (past, past_c)))
    _lib.get_lib().call('ha_tune_set', b'cu_poison', 0)
    av, word, launches = hm.persistent_rollout_status(dev)
    print(f'smoke: roll-out max|dworld|={err:.2e} rel grad err={gerr:.2e} | persistent path available={av} error word=0x{word:x} '
          f'launches fwd/bwd={launches & 0xffffffff}/{launches >> 32}')
    finite = all(bool(torch.isfinite(t).all()) for t in (world, pm, pv, z.grad, past.grad))
    assert finite, 'non-finite roll-out output or gradient'
    assert word == 0, 'the persistent roll-out reported an incomplete launch'
    assert err < 1e-4 and gerr < 1e-3
