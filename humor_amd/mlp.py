"""Frozen MLPs through the ha_mlp_* entry points (include/humor_amd.h): the two VPoser networks MotionOptimizer.latent2pose /
pose2latent evaluate in every closure (humor/fitting/motion_optimizer.py:1041-1063) and HuMoR's posterior encoder
(humor/models/humor_model.py:180-190).  The host only extracts the weights (folding VPoser's eval-mode BatchNorm layers into the
Linear layers that follow them); every layer, the activation functions, the 6-D -> rotation -> axis-angle tail and all adjoints run
in the HIP library."""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib

ACT_GN_RELU, ACT_LEAKY_RELU = 0, 1
TAIL_NONE, TAIL_ROT6D_AA = 0, 1


class FusedMLP:
    """Device handle of one packed MLP.  `linears`: [(W [out,in], b [out])]; `gns`: [(gamma, beta)] for the GroupNorm in front of
    Linear 1.. (act='gn_relu') or None (act='leaky_relu')."""

    def __init__(self, lib, device_index, linears, act='leaky_relu', slope=0.2, gns=None):
        self.lib = lib
        self.ptr = C.c_void_p()
        self.in_dim, self.out_dim = int(linears[0][0].shape[1]), int(linears[-1][0].shape[0])
        keep = []
        d = _lib.MlpDesc()
        d.n_linear, d.in_dim, d.skip_dim = len(linears), self.in_dim, 0
        f = lambda t: t.detach().to(dtype=torch.float32, device='cpu').contiguous()
        for i, (w, b) in enumerate(linears):
            w, b = f(w), f(b)
            keep += [w, b]
            d.out_dims[i] = w.shape[0]
            d.w[i], d.b[i] = w.data_ptr(), b.data_ptr()
            if gns is not None and i > 0:
                g, be = f(gns[i - 1][0]), f(gns[i - 1][1])
                keep += [g, be]
                d.gn_gamma[i], d.gn_beta[i] = g.data_ptr(), be.data_ptr()
        act_id = ACT_GN_RELU if act == 'gn_relu' else ACT_LEAKY_RELU
        lib.call('ha_mlp_create', C.byref(self.ptr), int(device_index), C.byref(d), act_id, C.c_float(slope))

    def __del__(self):
        try:
            if self.ptr:
                self.lib.call('ha_mlp_destroy', self.ptr)
        except Exception:
            pass

    def __call__(self, x, tail=TAIL_NONE):
        return _MLPFunction.apply(x, self, tail)


class _MLPFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, handle, tail):
        lib = handle.lib
        x = x.contiguous().float()
        N = x.shape[0]
        if x.shape[1] != handle.in_dim:
            raise ValueError(f'FusedMLP: input width {x.shape[1]} != {handle.in_dim}')
        n = C.c_int64()
        lib.call('ha_mlp_workspace', handle.ptr, N, C.byref(n))
        ws = torch.empty(n.value, dtype=torch.float32, device=x.device)
        out_w = handle.out_dim if tail == TAIL_NONE else handle.out_dim // 2
        y = torch.empty(N, out_w, dtype=torch.float32, device=x.device)
        lib.call('ha_mlp_forward', handle.ptr, N, _lib.ptr(x), tail, _lib.ptr(y), _lib.ptr(ws), _lib.stream_ptr(x))
        ctx.handle, ctx.ws, ctx.tail, ctx.N = handle, ws, tail, N
        return y

    @staticmethod
    def backward(ctx, g_y):
        handle = ctx.handle
        g_y = g_y.contiguous().float()
        g_x = torch.empty(ctx.N, handle.in_dim, dtype=torch.float32, device=g_y.device)
        handle.lib.call('ha_mlp_backward', handle.ptr, ctx.N, _lib.ptr(g_y), ctx.tail, _lib.ptr(ctx.ws), _lib.ptr(g_x), _lib.stream_ptr(g_y))
        return g_x, None, None


# ------------------------------------------------------------------------------------------------------------------------
# weight extraction
# ------------------------------------------------------------------------------------------------------------------------
def _fold_bn_into_next(bn, lin):
    """Linear(BatchNorm_eval(x)) as one Linear: W' = W diag(gamma / sigma), b' = b + W (beta - mean gamma / sigma) (fp64 on the host)."""
    w, b = lin.weight.detach().double().cpu(), lin.bias.detach().double().cpu()
    if bn is None:
        return w.float(), b.float()
    sigma = torch.sqrt(bn.running_var.detach().double().cpu() + bn.eps)
    gamma = bn.weight.detach().double().cpu() if bn.affine else torch.ones_like(sigma)
    beta = bn.bias.detach().double().cpu() if bn.affine else torch.zeros_like(sigma)
    scale = gamma / sigma
    shift = beta - bn.running_mean.detach().double().cpu() * scale
    return (w * scale.unsqueeze(0)).float(), (b + w @ shift).float()


def _leaky_sequential(seq):
    """[(W, b)] of an nn.Sequential of Linear / LeakyReLU(slope) modules, and the slope; None if it is anything else."""
    lins, slope = [], None
    mods = list(seq)
    for i, m in enumerate(mods):
        if isinstance(m, nn.Linear):
            lins.append((m.weight, m.bias))
        elif isinstance(m, nn.LeakyReLU):
            if slope is not None and slope != m.negative_slope:
                return None
            slope = m.negative_slope
        else:
            return None
    return lins, slope


def vposer_decoder_layers(pose_prior):
    """Linear layers of VPoser's decoder (LeakyReLU(0.2) between them, 6-D rotations out) or None if the module is not one we know.
    Real VPoser v1.0: bodyprior_dec_fc1 / fc2 / out, dropout (identity in eval mode) and the continuous rotation decoder."""
    if pose_prior.training:
        return None
    if all(hasattr(pose_prior, n) for n in ('bodyprior_dec_fc1', 'bodyprior_dec_fc2', 'bodyprior_dec_out')):
        if not getattr(pose_prior, 'use_cont_repr', True):
            return None
        ls = [pose_prior.bodyprior_dec_fc1, pose_prior.bodyprior_dec_fc2, pose_prior.bodyprior_dec_out]
        return [(l.weight, l.bias) for l in ls], 0.2
    dec = getattr(pose_prior, 'dec', None)
    if isinstance(dec, nn.Sequential):
        return _leaky_sequential(dec)
    return None


def vposer_encoder_layers(pose_prior):
    """Linear layers of VPoser's encoder up to the posterior mean (what pose2latent reads), BatchNorm folded in; None if unknown."""
    if pose_prior.training:
        return None
    if all(hasattr(pose_prior, n) for n in ('bodyprior_enc_fc1', 'bodyprior_enc_fc2', 'bodyprior_enc_mu')):
        l1 = _fold_bn_into_next(getattr(pose_prior, 'bodyprior_enc_bn1', None), pose_prior.bodyprior_enc_fc1)
        l2 = _fold_bn_into_next(getattr(pose_prior, 'bodyprior_enc_bn2', None), pose_prior.bodyprior_enc_fc2)
        mu = pose_prior.bodyprior_enc_mu
        return [l1, l2, (mu.weight, mu.bias)], 0.2
    enc, mu = getattr(pose_prior, 'enc', None), getattr(pose_prior, 'enc_mu', None)
    if isinstance(enc, nn.Sequential) and isinstance(mu, nn.Linear):
        r = _leaky_sequential(enc)
        if r is None or not isinstance(list(enc)[-1], nn.LeakyReLU):
            return None
        return r[0] + [(mu.weight, mu.bias)], r[1]
    return None


class FusedVPoser:
    """latent -> axis-angle pose and pose -> posterior mean of a VPoser-shaped module through the HIP library."""

    def __init__(self, pose_prior, lib, device_index):
        dec, enc = vposer_decoder_layers(pose_prior), vposer_encoder_layers(pose_prior)
        if dec is None or enc is None or dec[0][-1][0].shape[0] % 6 != 0:
            raise NotImplementedError('not a VPoser v1.0-shaped module (Linear + LeakyReLU decoder with 6-D outputs, eval mode)')
        self.dec = FusedMLP(lib, device_index, dec[0], act='leaky_relu', slope=dec[1])
        self.enc = FusedMLP(lib, device_index, enc[0], act='leaky_relu', slope=enc[1])
        self.num_joints = dec[0][-1][0].shape[0] // 6

    def decode_aa(self, z):
        """[N, latentD] -> [N, 3 num_joints] axis-angle (decode(output_type='matrot') + rotation_matrix_to_angle_axis)."""
        return self.dec(z, TAIL_ROT6D_AA)

    def encode_mean(self, pose_aa):
        """[N, 3 num_joints] -> [N, latentD], the mean of encode(pose)."""
        return self.enc(pose_aa, TAIL_NONE)


def humor_mlp(lib, device_index, mlp):
    """FusedMLP of a humor_model.MLP without skip input (the posterior encoder)."""
    lin, gns, skip = mlp.describe()
    if skip != 0 or len(gns) != len(lin) - 1:
        raise NotImplementedError('FusedMLP: GroupNorm MLP without skip input expected')
    return FusedMLP(lib, device_index, [(l.weight, l.bias) for l in lin], act='gn_relu', gns=[(g.weight, g.bias) for g in gns])
