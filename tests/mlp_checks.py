"""Parity checks of the ha_mlp_* path (VPoser decoder / encoder, HuMoR posterior encoder) against the plain PyTorch modules +
the oracle's R -> axis-angle (bit-pinned to transforms.py:243-389).  Emulator tier on CPU, gfx950 build on the GPU."""
import torch
import torch.nn as nn

from humor_amd import mlp as M
from humor_amd import synth
from oracle import humor_restated as H

import rollout_checks as RC


class RealShapedVPoser(nn.Module):
    """Member names and arithmetic of human_body_prior v1.0's VPoser (the snapshot fitting_utils.py:705-732 loads): eval-mode
    BatchNorm in front of both encoder Linears, dropout, LeakyReLU(0.2), continuous 6-D rotation decoder.  Random weights."""
    def __init__(self, seed=0, latentD=32, hidden=512, nj=21):
        super().__init__()
        torch.manual_seed(seed)
        self.latentD, self.num_joints, self.use_cont_repr = latentD, nj, True
        self.bodyprior_enc_bn1 = nn.BatchNorm1d(nj * 3)
        self.bodyprior_enc_fc1 = nn.Linear(nj * 3, hidden)
        self.bodyprior_enc_bn2 = nn.BatchNorm1d(hidden)
        self.bodyprior_enc_fc2 = nn.Linear(hidden, hidden)
        self.bodyprior_enc_mu = nn.Linear(hidden, latentD)
        self.bodyprior_enc_logvar = nn.Linear(hidden, latentD)
        self.dropout = nn.Dropout(p=.1)
        self.bodyprior_dec_fc1 = nn.Linear(latentD, hidden)
        self.bodyprior_dec_fc2 = nn.Linear(hidden, hidden)
        self.bodyprior_dec_out = nn.Linear(hidden, nj * 6)
        with torch.no_grad():
            for bn in (self.bodyprior_enc_bn1, self.bodyprior_enc_bn2):
                bn.running_mean.normal_(0, 0.3)
                bn.running_var.uniform_(0.5, 1.5)
                bn.weight.uniform_(0.7, 1.3)
                bn.bias.normal_(0, 0.1)
            self.bodyprior_dec_out.weight.mul_(0.3)
            self.bodyprior_dec_out.bias.copy_(torch.tensor([1., 0., 0., 1., 0., 0.]).repeat(nj))
        for p in self.parameters():
            p.requires_grad_(False)

    def encode(self, Pin):
        lr = lambda x: nn.functional.leaky_relu(x, negative_slope=.2)
        x = self.bodyprior_enc_bn1(Pin.reshape(Pin.size(0), -1))
        x = lr(self.bodyprior_enc_fc1(x))
        x = self.bodyprior_enc_bn2(x)
        x = lr(self.bodyprior_enc_fc2(x))
        return torch.distributions.normal.Normal(self.bodyprior_enc_mu(x), nn.functional.softplus(self.bodyprior_enc_logvar(x)))

    def decode(self, Zin, output_type='matrot'):
        lr = lambda x: nn.functional.leaky_relu(x, negative_slope=.2)
        x = lr(self.bodyprior_dec_fc1(Zin))
        x = self.dropout(x)
        x = lr(self.bodyprior_dec_fc2(x))
        x = self.bodyprior_dec_out(x)
        return synth._rot6d_to_mat(x).reshape(Zin.size(0), 1, -1, 9)


def _cmp(a, b, tol, what):
    e = (a.detach().cpu() - b.detach().cpu()).abs().max().item()
    s = max(1.0, b.detach().abs().max().item())
    assert e <= tol * s, f'{what}: max err {e:.3e} (scale {s:.3g})'


def check_vposer(lib, device, N=100, seed=0, real_shaped=False):
    vp = (RealShapedVPoser(seed) if real_shaped else synth.SynthVPoser(seed=seed)).eval()
    fv = M.FusedVPoser(vp, lib, device.index or 0 if device.type == 'cuda' else 0)
    g = torch.Generator().manual_seed(seed + 1)
    # ---- decode: latent -> axis-angle -------------------------------------------------------------------------------------
    zc = torch.randn(N, vp.latentD, generator=g).requires_grad_(True)
    z = zc.detach().clone().to(device).requires_grad_(True)
    aa_ref = H.rot_to_aa(vp.decode(zc).reshape(-1, 3, 3)).reshape(N, -1)
    aa = fv.decode_aa(z)
    assert aa.shape == (N, 63)
    _cmp(aa, aa_ref, 2e-5, 'decode')
    w = torch.randn(N, 63, generator=g)
    gz_ref = torch.autograd.grad((aa_ref * w).sum(), zc)[0]
    gz = torch.autograd.grad((aa * w.to(device)).sum(), z)[0]
    _cmp(gz, gz_ref, 1e-4, 'decode grad')
    # ---- encode: axis-angle -> posterior mean ------------------------------------------------------------------------------
    pc = (0.4 * torch.randn(N, 63, generator=g)).requires_grad_(True)
    p = pc.detach().clone().to(device).requires_grad_(True)
    mu_ref = vp.encode(pc).mean
    mu = fv.encode_mean(p)
    assert mu.shape == (N, vp.latentD)
    _cmp(mu, mu_ref, 2e-5, 'encode')
    w = torch.randn(N, vp.latentD, generator=g)
    gp_ref = torch.autograd.grad((mu_ref * w).sum(), pc)[0]
    gp = torch.autograd.grad((mu * w.to(device)).sum(), p)[0]
    _cmp(gp, gp_ref, 1e-4, 'encode grad')


def check_posterior(lib, device, N=70, seed=0):
    """HuMoR's posterior encoder [past 339 | next 339] -> 96 (GroupNorm(16) + ReLU) against the module's PyTorch forward, with
    gradients (the fitting path only reads the forward: infer_latent_motion's result is detached, motion_optimizer.py:356)."""
    hm, _ = RC.make_model(None, torch.device('cpu'), seed=seed, contractive=True)
    enc = hm.encoder
    f = M.humor_mlp(lib, device.index or 0 if device.type == 'cuda' else 0, enc)
    g = torch.Generator().manual_seed(seed + 2)
    xc = torch.cat([RC.canonical_state(N, g), RC.canonical_state(N, g)], 1).requires_grad_(True)
    x = xc.detach().clone().to(device).requires_grad_(True)
    y_ref = enc(xc)
    y = f(x)
    _cmp(y, y_ref, 1e-4, 'posterior')
    w = torch.randn(N, y_ref.shape[1], generator=g)
    g_ref = torch.autograd.grad((y_ref * w).sum(), xc)[0]
    gx = torch.autograd.grad((y * w.to(device)).sum(), x)[0]
    _cmp(gx, g_ref, 1e-3, 'posterior grad')


def check_posterior_param_grads(lib, device, N=5, seed=0):
    """HumorModel.posterior() / prior() in a training setting (parameters require grad, grad mode on): the encoder / prior weights must
    receive gradients (the fused ha_mlp_* path returns only dL/dx, so it may only serve frozen networks); with frozen parameters or
    under no_grad the same calls take the fused path and agree with the module's own forward."""
    hm, _ = RC.make_model(lib, device, seed=seed, contractive=True)
    hm.train()
    g = torch.Generator().manual_seed(seed + 3)
    past, nxt = RC.canonical_state(N, g).to(device), RC.canonical_state(N, g).to(device)
    (pm, pv), (qm, qv) = hm.infer_step(past, nxt)
    assert qm.requires_grad and pm.requires_grad
    (qm.square().sum() + qv.sum() + pm.square().sum() + pv.sum()).backward()
    for net in (hm.encoder, hm.prior_net):
        for p in net.parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all()
        assert any(p.grad.abs().max().item() > 0 for p in net.parameters())
    ref_q = qm.detach().clone()
    for p in hm.parameters():
        p.requires_grad_(False)
    hm._net_handles.clear()
    (_, _), (qm2, _) = hm.infer_step(past, nxt)          # frozen: the fused path
    assert len(hm._net_handles) > 0, 'frozen networks are expected to run through ha_mlp_*'
    _cmp(qm2, ref_q.cpu(), 1e-4, 'fused posterior vs module')


def check_infer_global_seq_golden(lib, device, tol=2e-5):
    """HumorModel.infer_global_seq (humor_model.py:1061-1165) and the stage-3 velocity estimators
    (motion_optimizer.py:744-800) on `device` tensors against the reference-generated fixture tests/golden/infer_global_seq.npz
    (oracle/make_golden_infer.py).  On a HIP device (and the emulator tier) the prior / posterior run through ha_mlp_*: the check
    asserts that the fused handles were built, i.e. that this is not the PyTorch-module branch."""
    import numpy as np
    from conftest import golden
    from humor_amd import frames, synth
    from humor_amd.humor_model import HumorModel
    gd = golden('infer_global_seq.npz')
    worst = 0.0
    for name in ('a', 'b'):
        hm = HumorModel(in_rot_rep='mat', out_rot_rep='aa', latent_size=48, model_data_config='smpl+joints+contacts', steps_in=1,
                        _lib_override=lib)
        hm.load_state_dict(synth.humor_state_dict(seed=int(gd[f'{name}_weight_seed'])))
        hm = hm.to(device).eval()
        for p in hm.parameters():
            p.requires_grad_(False)
        seq = {k: torch.from_numpy(gd[f'{name}_{k}']).to(device) for k in
               ('trans', 'trans_vel', 'root_orient', 'root_orient_vel', 'pose_body', 'joints', 'joints_vel')}
        with torch.no_grad():
            (pm, pv), (qm, qv) = hm.infer_global_seq(seq)
        fused = device.type == 'cuda' or (lib is not None and lib.emulator)
        if fused:
            assert {k[0] for k in hm._net_handles} >= {'encoder', 'prior_net'}, 'the fused MLP path did not run'
        for got, key in ((pm, 'prior_mu'), (pv, 'prior_var'), (qm, 'post_mu'), (qv, 'post_var')):
            ref = gd[f'{name}_{key}']
            assert tuple(got.shape) == ref.shape, (name, key, got.shape, ref.shape)
            e = np.abs(got.cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
            assert e < tol, (name, key, e)
            worst = max(worst, e)
        h = float(gd[f'{name}_vel_h'])
        x, rot = torch.from_numpy(gd[f'{name}_vel_x']).to(device), torch.from_numpy(gd[f'{name}_vel_rot']).to(device)
        lv, av = frames.estimate_linear_velocity(x, h).cpu().numpy(), frames.estimate_angular_velocity(rot, h).cpu().numpy()
        assert np.abs(lv - gd[f'{name}_lin_vel']).max() <= 1e-6 * max(1.0, np.abs(gd[f'{name}_lin_vel']).max()), name
        assert np.abs(av - gd[f'{name}_ang_vel']).max() <= 1e-5 * max(1.0, np.abs(gd[f'{name}_ang_vel']).max()), name
    return worst
