"""Drop-in replacement for the reference's ``HumorModel`` on the fitting path (humor/models/humor_model.py:102-1203).

Keeps the constructor signature, the parameter naming (``encoder|decoder|prior_net.net.<idx>.{weight,bias}``, so the
reference checkpoint loads with ``load_state_dict``), the attributes the optimiser reads (``latent_size``,
``use_conditional_prior``, ``model_data_config``, ``in_rot_rep``, ``data_names`` ...) and the methods it calls:
``roll_out`` (the autoregressive hot loop -> one C-ABI call forward, one backward, no per-step Python),
``infer_global_seq`` / ``prior`` / ``posterior`` / ``decode`` / ``sample_step`` / ``split_output`` / ``prepare_input``.

The roll-out kernels implement the configuration the fitting pipeline uses (in_rot_rep='mat', out_rot_rep='aa',
steps_in=1, 'smpl+joints(+contacts)', output_delta=True) and its output variants: out_rot_rep='6d' / '9d' (humor_model.py:476-484) and
output_delta=False (the decoder emits the state itself, :331-347); these take the launch-chain kernels, the persistent roll-out is
built for the residual 216-wide decoder.
The INPUT variants -- in_rot_rep 'aa' / '6d' (:462-478, 970-981) and steps_in > 1 (:838-850, 946-957) -- are off the fitting path (the released
checkpoint is 'mat' / 1): roll_out serves them with a step loop of on-device PyTorch operations (_roll_out_generic: prior through the fused MLP
kernels, decoder through its module), pinned to reference fixtures (tests/golden/rollout_inrep.npz).  model_use_smpl_joint_inputs (a training-time
option that needs one SMPL model file per gender) raises NotImplementedError.
"""
import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .tables import SMPL_JOINTS

IN_ROT_REPS = ['aa', '6d', 'mat']
OUT_ROT_REPS = ['aa', '6d', '9d']
ROT_REP_SIZE = {'aa': 3, '6d': 6, 'mat': 9, '9d': 9}
NUM_SMPL_JOINTS = len(SMPL_JOINTS)
NUM_BODY_JOINTS = NUM_SMPL_JOINTS - 1

# humor/datasets/amass_utils.py:28-91 (state vector composition)
_DATA_NAMES = ['trans', 'trans_vel', 'root_orient', 'root_orient_vel', 'pose_body', 'pose_body_vel', 'joints',
               'joints_vel', 'joints_orient_vel', 'verts', 'verts_vel', 'contacts']
_RETURN_CONFIGS = {
    'smpl+joints': {'trans', 'trans_vel', 'root_orient', 'root_orient_vel', 'pose_body', 'joints', 'joints_vel'},
    'smpl+joints+contacts': {'trans', 'trans_vel', 'root_orient', 'root_orient_vel', 'pose_body', 'joints', 'joints_vel', 'contacts'},
}


def data_name_list(return_config):
    if return_config not in _RETURN_CONFIGS:
        raise NotImplementedError(f"model_data_config '{return_config}' is not supported by humor_amd (fitting uses smpl+joints+contacts)")
    return [k for k in _DATA_NAMES if k in _RETURN_CONFIGS[return_config]]


def data_dim(dname, rot_rep_size=9):
    if dname in ['trans', 'trans_vel', 'root_orient_vel']:
        return 3
    if dname == 'root_orient':
        return rot_rep_size
    if dname == 'pose_body':
        return NUM_BODY_JOINTS * rot_rep_size
    if dname in ['joints', 'joints_vel']:
        return NUM_SMPL_JOINTS * 3
    if dname == 'contacts':
        return 9
    raise ValueError(dname)


class MLP(nn.Module):
    """Parameter container with the reference's module indexing (Linear@0, then GroupNorm@3k-2, ReLU@3k-1, Linear@3k);
    ``forward`` is the plain PyTorch evaluation used off the hot path (posterior inference once per fit)."""

    def __init__(self, layers=[3, 128, 128, 3], nonlinearity=nn.ReLU, use_gn=True, skip_input_idx=None):
        super(MLP, self).__init__()
        in_size, out_channels = layers[0], layers[1:]
        mods = [nn.Linear(in_size, out_channels[0])]
        skip_size = 0 if skip_input_idx is None else (in_size - skip_input_idx)
        for li in range(1, len(out_channels)):
            if use_gn:
                mods.append(nn.GroupNorm(16, out_channels[li - 1]))
            mods.extend([nonlinearity(), nn.Linear(out_channels[li - 1] + skip_size, out_channels[li])])
        self.net = nn.ModuleList(mods)
        self.skip_input_idx = skip_input_idx

    def forward(self, x):
        skip_in = x[:, self.skip_input_idx:] if self.skip_input_idx is not None else None
        for i, layer in enumerate(self.net):
            if skip_in is not None and i > 0 and isinstance(layer, nn.Linear):
                x = torch.cat([x, skip_in], dim=1)
            x = layer(x)
        return x

    def describe(self):
        """(n_linear, in_dim, skip_dim, out_dims, [(w,b)], [(gamma,beta) or None]) for packing."""
        lin = [m for m in self.net if isinstance(m, nn.Linear)]
        gns = [m for m in self.net if isinstance(m, nn.GroupNorm)]
        skip = 0 if self.skip_input_idx is None else lin[0].in_features - self.skip_input_idx
        return lin, gns, skip


class _NetHandle:
    def __init__(self, lib, device_index, decoder, prior, output_delta=True):
        self.lib = lib
        self.ptr = C.c_void_p()
        keep = []

        def desc(mlp):
            lin, gns, skip = mlp.describe()
            if len(gns) != len(lin) - 1:
                raise NotImplementedError('humor_amd roll-out kernels expect GroupNorm before every hidden Linear')
            d = _lib.MlpDesc()
            d.n_linear, d.in_dim, d.skip_dim = len(lin), lin[0].in_features, skip
            for i, l in enumerate(lin):
                w = l.weight.detach().float().cpu().contiguous()
                b = l.bias.detach().float().cpu().contiguous()
                keep.extend([w, b])
                d.out_dims[i] = l.out_features
                d.w[i], d.b[i] = w.data_ptr(), b.data_ptr()
                if i > 0:
                    g = gns[i - 1].weight.detach().float().cpu().contiguous()
                    be = gns[i - 1].bias.detach().float().cpu().contiguous()
                    keep.extend([g, be])
                    d.gn_gamma[i], d.gn_beta[i] = g.data_ptr(), be.data_ptr()
            return d
        dd, dp = desc(decoder), desc(prior)
        lib.call('ha_humor_net_create', C.byref(self.ptr), device_index, C.byref(dd), C.byref(dp))
        if not output_delta:
            lib.call('ha_humor_net_set_option', self.ptr, b'output_delta', 0)

    def __del__(self):
        try:
            if self.ptr:
                self.lib.call('ha_humor_net_destroy', self.ptr)
        except Exception:
            pass


class _RolloutFunction(torch.autograd.Function):
    """(past_in0 [B,339], z_seq [B,S,48]) -> (world [B,S,348], prior_mu [B,S,48], prior_var [B,S,48])."""

    @staticmethod
    def forward(ctx, past_in0, z_seq, handle, want_prior, z_thru=False):
        lib = handle.lib
        z_in = z_seq
        past_in0, z_seq = past_in0.contiguous().float(), z_seq.contiguous().float()
        B, S = z_seq.shape[0], z_seq.shape[1]
        dev = past_in0.device
        n = C.c_int64()
        lib.call('ha_humor_rollout_workspace', handle.ptr, B, S, C.byref(n))
        stash = torch.empty(n.value, dtype=torch.float32, device=dev)
        world = torch.empty(B, S, 348, dtype=torch.float32, device=dev)
        pm = torch.empty(B, S, 48, dtype=torch.float32, device=dev) if want_prior else None
        pv = torch.empty(B, S, 48, dtype=torch.float32, device=dev) if want_prior else None
        lib.call('ha_humor_rollout_forward', handle.ptr, B, S, _lib.ptr(past_in0), _lib.ptr(z_seq), _lib.ptr(world),
                 _lib.ptr(pm), _lib.ptr(pv), _lib.ptr(stash), _lib.stream_ptr(past_in0))
        ctx.handle, ctx.stash, ctx.dims, ctx.want_prior = handle, stash, (B, S), want_prior
        ctx.save_for_backward(z_seq)
        ctx.set_materialize_grads(False)
        # z_thru: the latent sequence handed through as a fourth output -- a later reader of it (the motion-prior loss term) then sends
        # its gradient here, where it is added inside the adjoint's last kernel (ha_humor_rollout_backward_ex) instead of by autograd
        return world, pm, pv, (z_in if z_thru else None)

    @staticmethod
    def backward(ctx, g_world, g_pm, g_pv, g_z_thru=None):
        handle, (B, S) = ctx.handle, ctx.dims
        lib = handle.lib
        z_seq, = ctx.saved_tensors
        dev = z_seq.device
        c = lambda t: None if t is None else t.contiguous().float()
        g_world, g_pm, g_pv, g_z_thru = c(g_world), c(g_pm), c(g_pv), c(g_z_thru)
        g_past = torch.empty(B, 339, dtype=torch.float32, device=dev)
        g_z = torch.empty(B, S, 48, dtype=torch.float32, device=dev)
        lib.call('ha_humor_rollout_backward_ex', handle.ptr, B, S, _lib.ptr(z_seq), _lib.ptr(g_world), _lib.ptr(g_pm), _lib.ptr(g_pv),
                 _lib.ptr(ctx.stash), _lib.ptr(g_past), _lib.ptr(g_z), _lib.ptr(g_z_thru), _lib.stream_ptr(z_seq))
        return g_past, g_z, None, None, None


def _rollout_sample(handle, past_in, eps, S):
    """Sampling roll-out through ha_humor_rollout_sample (no autograd)."""
    lib = handle.lib
    past_in = past_in.contiguous().float()
    B, dev = past_in.shape[0], past_in.device
    n = C.c_int64()
    lib.call('ha_humor_rollout_workspace', handle.ptr, B, S, C.byref(n))
    stash = torch.empty(n.value, dtype=torch.float32, device=dev)
    new = lambda d: torch.empty(B, S, d, dtype=torch.float32, device=dev)
    world, pm, pv, z = new(348), new(48), new(48), new(48)
    eps = None if eps is None else eps.contiguous().float()
    lib.call('ha_humor_rollout_sample', handle.ptr, B, S, _lib.ptr(past_in), _lib.ptr(eps), _lib.ptr(world), _lib.ptr(pm),
             _lib.ptr(pv), _lib.ptr(z), _lib.ptr(stash), _lib.stream_ptr(past_in))
    return world, pm, pv, z


class HumorModel(nn.Module):

    def __init__(self, in_rot_rep='aa', out_rot_rep='aa', latent_size=48, steps_in=1, conditional_prior=True,
                 output_delta=True, posterior_arch='mlp', decoder_arch='mlp', prior_arch='mlp',
                 model_data_config='smpl+joints+contacts', detach_sched_samp=True, model_use_smpl_joint_inputs=False,
                 model_smpl_batch_size=1, _lib_override=None):
        super(HumorModel, self).__init__()
        if out_rot_rep not in OUT_ROT_REPS:
            raise Exception('Not a valid output rotation representation: %s' % (out_rot_rep))
        if in_rot_rep not in IN_ROT_REPS:
            raise Exception('Not a valid input rotation representation: %s' % (in_rot_rep))
        if 'mlp' not in (posterior_arch, decoder_arch, prior_arch) or {posterior_arch, decoder_arch, prior_arch} != {'mlp'}:
            raise Exception('Only mlp architectures exist')
        if model_use_smpl_joint_inputs:
            raise NotImplementedError('model_use_smpl_joint_inputs is a training-time option outside the fitting path')
        self.ignore_keys = []
        self.steps_in, self.steps_out, self.out_step_size = steps_in, 1, 1
        self.detach_sched_samp = detach_sched_samp
        self.output_delta = output_delta
        self.out_rot_rep, self.in_rot_rep = out_rot_rep, in_rot_rep
        self.posterior_arch, self.decoder_arch, self.prior_arch = posterior_arch, decoder_arch, prior_arch
        self.data_names = data_name_list(model_data_config)
        self.aux_in_data_names = self.aux_out_data_names = None
        self.pred_contacts = False
        if 'contacts' in model_data_config:
            self.data_names.remove('contacts')
            self.aux_out_data_names = ['contacts']
            self.pred_contacts = True
        self.need_trans2joint = 'joints' in self.data_names
        self.model_data_config = model_data_config
        self.input_rot_dim = ROT_REP_SIZE[in_rot_rep]
        self.input_dim_list = [data_dim(d, self.input_rot_dim) for d in self.data_names]
        self.input_data_dim = sum(self.input_dim_list)
        self.output_rot_dim = ROT_REP_SIZE[out_rot_rep]
        self.output_dim_list = [data_dim(d, self.output_rot_dim) for d in self.data_names]
        self.delta_output_dim_list = [data_dim(d, 9) for d in self.data_names]
        if self.pred_contacts:
            self.output_dim_list.append(9)
            self.delta_output_dim_list.append(9)
        self.output_data_dim = sum(self.output_dim_list)
        self.latent_size = latent_size
        past_dim = steps_in * self.input_data_dim
        self.encoder = MLP([past_dim + self.input_data_dim, 1024, 1024, 1024, 1024, latent_size * 2])
        self.decoder = MLP([past_dim + latent_size, 1024, 1024, 512, self.output_data_dim], skip_input_idx=past_dim)
        self.use_conditional_prior = conditional_prior
        if conditional_prior:
            self.prior_net = MLP([past_dim, 1024, 1024, 1024, 1024, latent_size * 2])
        self.use_smpl_joint_inputs = False
        self.smpl_batch_size = model_smpl_batch_size
        self._lib = _lib_override
        self._net_handles = {}
        self._plists = {}           # parameter lists by sub-network name (None = all): walking the module tree costs ~0.15 ms per roll_out

    def _params_of(self, name=None):
        """The parameters of sub-network `name` (all when None) as a cached list.  Module._apply (.to / .cuda / .float) and
        load_state_dict may replace parameter objects: both reset the cache."""
        lst = self._plists.get(name)
        if lst is None:
            lst = self._plists[name] = list((self if name is None else getattr(self, name)).parameters())
        return lst

    def _apply(self, fn, *args, **kwargs):
        self._plists = {}
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._plists = {}
        return super().load_state_dict(*args, **kwargs)

    # ------------------------------------------------------------------------------------------------
    # single-step API (off the hot path; the prior / posterior MLPs run through ha_mlp_* on a HIP device)
    # ------------------------------------------------------------------------------------------------
    def _fused_forward(self, name, x):
        """The skip-free GroupNorm MLP `name` ('encoder' / 'prior_net') through ha_mlp_* (humor_amd/mlp.py) for inputs on a HIP device
        (or the emulator tier); the module's own PyTorch forward for host tensors."""
        net = getattr(self, name)
        on_dev = x.is_cuda or (self._lib is not None and self._lib.emulator)
        # the fused path snapshots the weights and returns only dL/dx: it serves the frozen-network uses (fitting, inference).
        # A caller that needs weight gradients (a training step whose KL term comes from infer()) gets the module's own forward.
        plist = self._params_of(name)
        needs_param_grad = torch.is_grad_enabled() and any(p.requires_grad for p in plist)
        if not on_dev or net.skip_input_idx is not None or needs_param_grad:
            return net(x)
        from .mlp import humor_mlp
        lib = self._lib if self._lib is not None else _lib.get_lib()
        ver = tuple(p._version for p in plist)
        key = (name, x.device.type, x.device.index or 0)
        cached = self._net_handles.get(key)
        if cached is None or cached[0] != ver:
            self._net_handles[key] = cached = (ver, humor_mlp(lib, x.device.index or 0, net))
        return cached[1](x)

    def prior(self, past_in):
        out = self._fused_forward('prior_net', past_in)
        return out[:, :self.latent_size], torch.exp(out[:, self.latent_size:])

    def posterior(self, past_in, t_in):
        out = self._fused_forward('encoder', torch.cat([past_in, t_in], dim=1))
        return out[:, :self.latent_size], torch.exp(out[:, self.latent_size:])

    def rsample(self, mu, var):
        return mu + torch.randn_like(mu) * torch.sqrt(var)

    def infer_step(self, past_in, t_in):
        qm, qv = self.posterior(past_in, t_in)
        if self.use_conditional_prior:
            pm, pv = self.prior(past_in)
        else:
            pm, pv = torch.zeros_like(qm), torch.ones_like(qv)
        return (pm, pv), (qm, qv)

    def infer(self, x_past, x_t):
        B = x_past.size(0)
        return self.infer_step(x_past.reshape(B, -1), x_t.reshape(B, -1))

    def single_step(self, past_in, t_in):
        """One training / evaluation step (humor_model.py:374-404): posterior from (past, next), prior from the past, z sampled from the
        posterior, decoded; returns the split prediction + 'posterior_distrib' / 'prior_distrib' as (mean, var).  Off the fitting path:
        the MLPs go through ha_mlp_* when no parameter gradient is needed, the residual composition is plain PyTorch (decode)."""
        B = past_in.size(0)
        (pm, pv), (qm, qv) = self.infer_step(past_in, t_in)
        z = self.rsample(qm, qv)
        x_pred = self.split_output(self.decode(z, past_in).reshape(B, self.steps_out, -1))
        x_pred['posterior_distrib'] = (qm, qv)
        x_pred['prior_distrib'] = (pm, pv)
        return x_pred

    def forward(self, x_past, x_t):
        """Single-step full forward pass with the posterior's sample (humor_model.py:352-372).  x_past [B, steps_in, D], x_t [B, steps_out, D]."""
        B = x_past.size(0)
        return self.single_step(x_past.reshape(B, -1), x_t.reshape(B, -1))

    def _delta_rotmat(self, raw):
        """Residual rotations from the decoder's raw output [N, rot_dim] -> [N,3,3] in plain PyTorch (convert_to_rotmat of the
        out_rot_rep, humor/utils/transforms.py:60-73): Rodrigues, 6-D Gram-Schmidt (:201-220) or the SVD projection (:222-241)."""
        from .frames import _rodrigues_torch
        if self.out_rot_rep == 'aa':
            return _rodrigues_torch(raw)
        if self.out_rot_rep == '6d':
            x = raw.reshape(-1, 3, 2)
            a1, a2 = x[:, :, 0], x[:, :, 1]
            b1 = torch.nn.functional.normalize(a1)
            b2 = torch.nn.functional.normalize(a2 - (b1 * a2).sum(1, keepdim=True) * b1)
            return torch.stack((b1, b2, torch.cross(b1, b2, dim=1)), dim=-1)
        x = raw.reshape(-1, 3, 3)
        u, _, vh = torch.linalg.svd(x)
        s_p = torch.eye(3, dtype=x.dtype, device=x.device).expand_as(x).clone()
        s_p[:, 2, 2] = torch.det(torch.matmul(u, vh))
        return torch.matmul(torch.matmul(u, s_p), vh)

    def _rep_to_rotmat(self, x, rep):
        """[N, rot_dim] in representation `rep` -> [N,3,3] (convert_to_rotmat, humor/utils/transforms.py:60-73); 'mat' / '9d'-as-matrix passes through."""
        from .frames import _rodrigues_torch
        if rep == 'mat':
            return x.reshape(-1, 3, 3)
        if rep == 'aa':
            return _rodrigues_torch(x.reshape(-1, 3))
        if rep == '6d':
            v = x.reshape(-1, 3, 2)
            a1, a2 = v[:, :, 0], v[:, :, 1]
            b1 = torch.nn.functional.normalize(a1)
            b2 = torch.nn.functional.normalize(a2 - (b1 * a2).sum(1, keepdim=True) * b1)
            return torch.stack((b1, b2, torch.cross(b1, b2, dim=1)), dim=-1)
        raise ValueError(rep)

    def _last_step_as_mat(self, past_in):
        """The most recent input step of past_in [B, steps_in * D_in] with its rotations as matrices: [B, 339] in the 'mat' layout
        (humor_model.py:462-478)."""
        B = past_in.size(0)
        step = past_in.reshape(B, self.steps_in, -1)[:, -1]
        if self.in_rot_rep == 'mat':
            return step
        w = self.input_rot_dim
        parts, o = [], 0
        for n, d in zip(self.data_names, self.input_dim_list):
            v = step[:, o:o + d]
            o += d
            if n == 'root_orient':
                v = self._rep_to_rotmat(v, self.in_rot_rep).reshape(B, 9)
            elif n == 'pose_body':
                v = self._rep_to_rotmat(v.reshape(B * NUM_BODY_JOINTS, w), self.in_rot_rep).reshape(B, NUM_BODY_JOINTS * 9)
            parts.append(v)
        return torch.cat(parts, dim=1)

    def decode(self, z, past_in):
        """One decoder evaluation + residual composition in plain PyTorch (humor_model.py:445-498): the canonical-frame state after
        one step, [B, 339 (+9 contact logits)], rotations as matrices.  Off the hot path: roll_out runs the same arithmetic in the HIP
        kernels for the fitting configuration; the input variants (in_rot_rep 'aa' / '6d', steps_in > 1) come through here."""
        B = z.size(0)
        past_in = past_in.reshape(B, -1)
        raw = self.decoder(torch.cat([past_in, z], dim=1))
        if not self.output_delta:
            return raw                       # the network's output is the state itself; split_output converts its rotations
        past_in = self._last_step_as_mat(past_in)
        w = self.output_rot_dim
        o_root, o_rvel, o_body, o_j = 6, 6 + w, 9 + w, 9 + 22 * w
        dR = self._delta_rotmat(raw[:, o_root:o_root + w])
        R_root = torch.matmul(dR, past_in[:, 6:15].reshape(B, 3, 3)).reshape(B, 9)
        dB = self._delta_rotmat(raw[:, o_body:o_body + NUM_BODY_JOINTS * w].reshape(B * NUM_BODY_JOINTS, w)).reshape(B, NUM_BODY_JOINTS, 3, 3)
        R_body = torch.matmul(dB, past_in[:, 18:207].reshape(B, NUM_BODY_JOINTS, 3, 3)).reshape(B, NUM_BODY_JOINTS * 9)
        out = torch.cat([raw[:, 0:3] + past_in[:, 0:3], raw[:, 3:6] + past_in[:, 3:6], R_root, raw[:, o_rvel:o_rvel + 3] + past_in[:, 15:18], R_body,
                         raw[:, o_j:o_j + 66] + past_in[:, 207:273], raw[:, o_j + 66:o_j + 132] + past_in[:, 273:339]], dim=1)
        if self.pred_contacts:
            out = torch.cat([out, raw[:, o_j + 132:o_j + 141]], dim=1)
        return out

    def sample_step(self, past_in, t_in=None, use_mean=False, z=None, return_prior=False, return_z=False):
        """One sampling step in plain PyTorch (humor_model.py:1019-1059): z ~ prior(past_in) (or its mean, or the given z),
        then decode.  Returns {'decoder_out': [B, 1, D]} (+ 'prior', 'z')."""
        B = past_in.size(0)
        past_in = past_in.reshape(B, -1)
        pm, pv = self.prior(past_in) if self.use_conditional_prior else (torch.zeros(B, self.latent_size, device=past_in.device),
                                                                          torch.ones(B, self.latent_size, device=past_in.device))
        if z is None:
            z = pm if use_mean else self.rsample(pm, pv)
        out = {'decoder_out': self.decode(z, past_in).reshape(B, 1, -1)}
        if return_prior:
            out['prior'] = (pm, pv)
        if return_z:
            out['z'] = z
        return out

    def split_output(self, decoder_out, convert_rots=True):
        B = decoder_out.size(0)
        decoder_out = decoder_out.reshape(B, self.steps_out, -1)
        names = self.data_names + (self.aux_out_data_names or [])
        dims = self.delta_output_dim_list if self.output_delta else self.output_dim_list
        out, s = {}, 0
        for n, d in zip(names, dims):
            out[n] = decoder_out[:, :, s:s + d]
            s += d
        if convert_rots and not self.output_delta:     # residual outputs already are rotation matrices (humor_model.py:341-346)
            for n in ('root_orient', 'pose_body'):
                if n in out:
                    v = out[n]
                    out[n] = self._delta_rotmat(v.reshape(-1, self.output_rot_dim)).reshape(B, self.steps_out, -1)
        return out

    def prepare_input(self, data_in, device, data_out=None, return_input_dict=False, return_global_dict=False):
        """Concatenates per-key data [B,T,...] into x_past [B,T,steps_in,D] (humor_model.py:233-314, input side)."""
        if data_out is not None or return_global_dict:
            raise NotImplementedError('training-side prepare_input (data_out / global dict) is outside the fitting path')
        parts = []
        for k in self.data_names:
            cur = data_in[k].to(device)
            parts.append(cur.reshape(cur.size(0), cur.size(1), self.steps_in, -1))
        x_past = torch.cat(parts, dim=3)
        if return_input_dict:
            return x_past, {k: v for k, v in zip(self.data_names, parts)}
        return x_past

    # ------------------------------------------------------------------------------------------------
    # roll-out (hot path)
    # ------------------------------------------------------------------------------------------------
    def _check_rollout_config(self):
        if not (self.in_rot_rep == 'mat' and self.steps_in == 1 and self.input_data_dim == 339):
            raise NotImplementedError("humor_amd roll-out kernels implement in_rot_rep='mat', out_rot_rep 'aa' / '6d' / '9d', steps_in=1, "
                                      "output_delta True / False, 'smpl+joints(+contacts)' (the fitting configuration and its output "
                                      "variants)")
        if not self.use_conditional_prior:
            raise NotImplementedError('roll-out kernels expect the conditional prior network')
        if not self.pred_contacts:
            raise NotImplementedError("roll-out kernels expect the contact head (model_data_config='smpl+joints+contacts': 216 / 282 / 348 decoder outputs)")

    def _net_handle(self, device):
        lib = self._lib if self._lib is not None else _lib.get_lib()
        if device.type == 'cuda':
            index = device.index if device.index is not None else torch.cuda.current_device()
        elif lib.emulator:
            index = 0
        else:
            raise _lib.HumorAmdError('HumorModel.roll_out runs on the GPU only (no CPU fallback): move the model inputs to a HIP device')
        ver = tuple(p._version for p in self._params_of())
        key = (device.type, index)
        cached = self._net_handles.get(key)
        if cached is None or cached[0] != ver:
            self._net_handles[key] = (ver, _NetHandle(lib, index, self.decoder, self.prior_net, output_delta=self.output_delta))
        return self._net_handles[key][1]

    def persistent_rollout_status(self, device):
        """(available, error word, launches) of the one-launch persistent roll-out kernels on `device`.  A persistent launch is asynchronous:
        a team that did not complete fills its results with NaN and sets the host-mapped error word (rollout_persist.hip); read this
        after the stream has been synchronised (e.g. after L-BFGS's host read) -- captured hipGraphs never pass an entry point that
        would report it.  `error word != 0` means: results since then are invalid, the launch chain serves this network from now on."""
        h = self._net_handle(device)
        av, err, n = C.c_int(), C.c_uint(), C.c_int64()
        h.lib.call('ha_humor_persist_status', h.ptr, C.byref(av), C.byref(err), C.byref(n))
        return av.value, err.value, n.value

    def acknowledge_persistent_failure(self, device):
        """The caller has read a non-zero error word (persistent_rollout_status) and handled the failure: later roll-out entry points on this
        network do not report it again (they run on the launch chain)."""
        h = self._net_handle(device)
        h.lib.call('ha_humor_persist_ack', h.ptr)

    def _cached_handle(self, ref):
        dev = ref.device
        index = (dev.index if dev.index is not None else torch.cuda.current_device()) if dev.type == 'cuda' else 0
        cached = self._net_handles.get((dev.type, index))
        return None if cached is None else cached[1]

    def roll_out(self, x_past, init_input_dict, num_steps, use_mean=False, z_seq=None, return_prior=False, gender=None,
                 betas=None, return_z=False, canonicalize_input=False, uncanonicalize_output=False, eps_seq=None, return_world=False):
        '''
        Rolls the model out from the initial state (humor_model.py:785-1017): with the given latent sequence (differentiable,
        the fitting path) or, with z_seq=None, sampling z_t = mu_t + eps_t * sigma_t from the conditional prior at every step
        (use_mean: z_t = mu_t; forward only).  eps_seq [B,S,48] is an extension for reproducible sampling (default: randn).
        Returns a dict of world-frame [B, num_steps, D] tensors (rotations as matrices), optionally (prior mean, var).
        '''
        if self.in_rot_rep != 'mat' or self.steps_in != 1:
            return self._roll_out_generic(x_past, init_input_dict, num_steps, use_mean, z_seq, return_prior, return_z, canonicalize_input,
                                          uncanonicalize_output, eps_seq)
        self._check_rollout_config()
        if x_past is not None:
            past_in = x_past.reshape(x_past.size(0), -1)
        elif init_input_dict is not None:
            past_in = torch.cat([init_input_dict[k][:, -1, :] for k in self.data_names], dim=1)
        else:
            raise ValueError('roll_out needs the initial state: x_past [B, steps_in, D] or init_input_dict {name: [B, steps_in, d]}')
        B = past_in.size(0)
        uncanon = None
        if canonicalize_input:
            # express the initial state in its own heading-aligned frame (humor_model.py:808-832)
            from .frames import canonicalize_state
            past_in, uncanon = canonicalize_state(past_in)
        handle = self._net_handle(past_in.device)
        z_out = None
        if z_seq is None:
            # sampling from the conditional prior (or its mean): forward only
            if use_mean:
                eps = None
            elif eps_seq is not None:
                eps = eps_seq[:, :num_steps]
            else:
                eps = torch.randn(B, num_steps, self.latent_size, device=past_in.device)
            world, pm, pv, z_out = _rollout_sample(handle, past_in.detach(), eps, num_steps)
        else:
            z_seq = z_seq[:, :num_steps]
            world, pm, pv, z_t = _RolloutFunction.apply(past_in, z_seq, handle, bool(return_prior), bool(return_world and return_z))
            z_out = z_seq if z_t is None else z_t
        if canonicalize_input and uncanonicalize_output:
            from .frames import uncanonicalize_world
            world = uncanonicalize_world(world, *uncanon)
        if return_world:
            # the undivided [B, S, 348] world-frame state (what the fused post-processing kernel reads) instead of the dict
            if return_z:
                # (extension) the latent sequence handed through the roll-out node: read THIS tensor in later loss terms instead of the one
                # passed in, and its gradient joins the roll-out's inside the adjoint kernel (no accumulation launch)
                return (world, (pm, pv), z_out) if return_prior else (world, z_out)
            return (world, (pm, pv)) if return_prior else world
        # one split (views forward, a single cat backward) instead of one slice + zero-fill + add per output
        dims = list(self.delta_output_dim_list)
        rest = world.size(2) - sum(dims)
        parts = torch.split(world, dims + ([rest] if rest > 0 else []), dim=2)
        out = dict(zip(self.data_names, parts))
        if self.pred_contacts:
            out['contacts'] = parts[len(self.data_names)]
        if return_z:
            out['z'] = z_out
        if return_prior:
            return out, (pm, pv)
        return out

    def _window_as_input(self, win):
        """Window {name: [B, S, d]} (rotations as matrices) -> past_in [B, S * D_in] with the rotations in in_rot_rep (humor_model.py:970-981:
        'aa' through rotation_matrix_to_angle_axis, '6d' = the first six entries of the row-major matrix, as the reference slices them)."""
        from . import ops
        parts = []
        for k in self.data_names:
            v = win[k]
            B, S = v.shape[0], v.shape[1]
            if k in ('root_orient', 'pose_body') and self.in_rot_rep != 'mat':
                nj = v.shape[2] // 9
                if self.in_rot_rep == 'aa':
                    v = ops.rotation_matrix_to_angle_axis(v.reshape(B * S * nj, 3, 3), _lib_override=self._lib).reshape(B, S, nj * 3)
                else:
                    v = v.reshape(B, S, nj, 9)[:, :, :, :6].reshape(B, S, nj * 6)
            parts.append(v)
        return torch.cat(parts, dim=2).reshape(parts[0].shape[0], -1)

    def _roll_out_generic(self, x_past, init_input_dict, num_steps, use_mean, z_seq, return_prior, return_z, canonicalize_input,
                          uncanonicalize_output, eps_seq):
        """HumorModel.roll_out (humor_model.py:785-1017) for the input variants the roll-out kernels do not implement: in_rot_rep 'aa' / '6d'
        and steps_in > 1 (the released checkpoint and every fitting configuration use 'mat' / 1: those run the HIP kernels above).  The same
        arithmetic as a step loop of on-device PyTorch operations (differentiable): prior through the fused MLP kernels when the network is
        frozen, decoder through its module, the frame changes of the whole input window per step (frames.window_to_local), the world-frame
        outputs from the accumulated transform.  init_input_dict holds the window {name: [B, <= steps_in, d]} with rotations as MATRICES, x_past
        the same window in the input representation [B, <= steps_in, D_in] (the reference needs both for in_rot_rep != 'mat': test_humor.py:
        210-224; here x_past alone is enough)."""
        from . import frames
        names = self.data_names
        if x_past is not None and x_past.dim() == 2:
            x_past = x_past.reshape(x_past.shape[0], -1, self.input_data_dim)
        if init_input_dict is not None:
            win = {k: init_input_dict[k] for k in names}
        elif x_past is not None:
            B0, S0 = x_past.shape[0], x_past.shape[1]
            win, o = {}, 0
            for k, d in zip(names, self.input_dim_list):
                v = x_past[:, :, o:o + d]
                o += d
                if k in ('root_orient', 'pose_body') and self.in_rot_rep != 'mat':
                    nj = d // self.input_rot_dim
                    v = self._rep_to_rotmat(v.reshape(B0 * S0 * nj, self.input_rot_dim), self.in_rot_rep).reshape(B0, S0, nj * 9)
                win[k] = v
        else:
            raise ValueError('roll_out needs the initial state: x_past [B, steps_in, D] or init_input_dict {name: [B, steps_in, d]}')
        ref = win[names[0]]
        B, dt, dev = ref.shape[0], ref.dtype, ref.device
        zero = torch.zeros(B, 1, dtype=dt, device=dev)
        W0 = wt0 = None
        if canonicalize_input:
            W0 = frames.world2aligned_mat(win['root_orient'][:, -1].reshape(B, 3, 3))
            wt0 = torch.cat([-win['trans'][:, -1, :2], zero], dim=1)
            t2j0 = torch.cat([-(win['joints'][:, -1, :2] + wt0[:, :2]), zero], dim=1) if self.need_trans2joint else torch.zeros(B, 3, dtype=dt, device=dev)
            win = frames.window_to_local(win, W0, wt0, t2j0)
        pad = self.steps_in - ref.shape[1]
        if pad > 0:
            win = {k: torch.cat([torch.zeros(B, pad, v.shape[2], dtype=dt, device=dev), v], dim=1) for k, v in win.items()}
        if x_past is None or canonicalize_input:
            past_in = self._window_as_input(win)
        else:
            xp = x_past
            if xp.shape[1] < self.steps_in:
                xp = torch.cat([torch.zeros(B, self.steps_in - xp.shape[1], xp.shape[2], dtype=dt, device=dev), xp], dim=1)
            past_in = xp.reshape(B, -1)
        G = torch.eye(3, dtype=dt, device=dev).unsqueeze(0).repeat(B, 1, 1)
        gt = torch.zeros(B, 3, dtype=dt, device=dev)
        if canonicalize_input and uncanonicalize_output:
            G, gt = W0, wt0
        t2j = torch.cat([-win['joints'][:, -1, :2], zero], dim=1) if self.need_trans2joint else torch.zeros(B, 3, dtype=dt, device=dev)
        dims = list(self.delta_output_dim_list)
        out_names = names + (self.aux_out_data_names or [])
        world, pms, pvs, zs = [], [], [], []
        for t in range(num_steps):
            pm = pv = None
            if return_prior or z_seq is None:
                pm, pv = self.prior(past_in) if self.use_conditional_prior else (torch.zeros(B, self.latent_size, dtype=dt, device=dev),
                                                                                  torch.ones(B, self.latent_size, dtype=dt, device=dev))
                pms.append(pm)
                pvs.append(pv)
            if z_seq is not None:
                z = z_seq[:, t]
            elif use_mean:
                z = pm
            else:
                eps = eps_seq[:, t] if eps_seq is not None else torch.randn(B, self.latent_size, dtype=dt, device=dev)
                z = pm + eps * torch.sqrt(pv)
            zs.append(z)
            state = self.decode(z, past_in)
            if not self.output_delta:
                state = torch.cat([v.reshape(B, -1) for v in self.split_output(state).values()], dim=1)
            pred = dict(zip(out_names, (v.unsqueeze(1) for v in torch.split(state, dims, dim=1))))
            win = {k: torch.cat([win[k][:, 1:], pred[k]], dim=1) for k in names}
            W = frames.world2aligned_mat(pred['root_orient'][:, 0].reshape(B, 3, 3))
            wt = torch.cat([-pred['trans'][:, 0, :2], zero], dim=1)
            win = frames.window_to_local(win, W, wt, t2j)
            wd = frames.window_to_world(pred, G, gt, t2j)
            world.append(wd)
            gt = torch.cat([-wd['trans'][:, 0, :2], zero], dim=1)
            G = torch.matmul(G, W)
            past_in = self._window_as_input(win)
        out = {k: torch.cat([w[k] for w in world], dim=1) for k in out_names}
        if return_z:
            out['z'] = torch.stack(zs, dim=1)
        if return_prior:
            return out, (torch.stack(pms, dim=1), torch.stack(pvs, dim=1))
        return out

    # ------------------------------------------------------------------------------------------------
    # posterior inference over a whole sequence (once per fit; PyTorch ops, all T-1 pairs batched)
    # ------------------------------------------------------------------------------------------------
    def infer_global_seq(self, global_seq, full_forward_pass=False):
        '''
        Canonicalises every consecutive frame pair (t, t+1) into frame t's aligned coordinate system and evaluates
        prior and posterior for all pairs at once (humor_model.py:1061-1165 does the same with a Python loop over t).
        '''
        if self.steps_in != 1 or self.in_rot_rep != 'mat':
            raise NotImplementedError("infer_global_seq implements steps_in=1, in_rot_rep='mat'")
        from .frames import canonicalize_pairs
        x_past, x_t = canonicalize_pairs(global_seq, self.data_names)      # [B*(T-1), D] each
        B, T = global_seq['trans'].shape[0], global_seq['trans'].shape[1]
        if full_forward_pass:
            # the full single-step pass for every pair (humor_model.py:1131-1160 stacks its per-step dictionaries along a new time axis);
            # the posterior's samples are drawn for all pairs at once (one draw per pair, not the reference's per-step stream)
            pred = self.single_step(x_past, x_t)
            r4 = lambda a: a.reshape(B, T - 1, *a.shape[1:])
            return {k: ((r4(v[0]), r4(v[1])) if k in ('posterior_distrib', 'prior_distrib') else r4(v)) for k, v in pred.items()}
        (pm, pv), (qm, qv) = self.infer_step(x_past, x_t)
        r = lambda a: a.reshape(B, T - 1, -1)
        return (r(pm), r(pv)), (r(qm), r(qv))
