"""ctypes binding of libhumor_amd.so (C ABI: include/humor_amd.h).

The product path loads humor_amd/csrc/libhumor_amd.so -- the hipcc/gfx950 build -- and raises if it is missing
or if the device is not a gfx950: there is no CPU or PyTorch fallback for the kernels.
(`load(path)` exists so the CPU-only test tier can bind the host SIMT-emulator build of the same sources.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, 'csrc', 'libhumor_amd.so')

HA_OK = 0
ABI_VERSION = 3

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int32)


class MlpDesc(C.Structure):
    _fields_ = [('n_linear', C.c_int), ('in_dim', C.c_int), ('skip_dim', C.c_int), ('out_dims', C.c_int * 8),
                ('w', C.c_void_p * 8), ('b', C.c_void_p * 8), ('gn_gamma', C.c_void_p * 8), ('gn_beta', C.c_void_p * 8)]


FIT_NTERMS = 17


class FitArgs(C.Structure):
    """ha_fit_args (include/humor_amd.h)."""
    _fields_ = [('B', C.c_int), ('T', C.c_int),
                ('cam_jtr', C.c_void_p), ('nj', C.c_int), ('cam_verts', C.c_void_p), ('nv', C.c_int),
                ('pri_joints', C.c_void_p), ('pri_nj', C.c_int), ('ro_joints', C.c_void_p), ('contacts_conf', C.c_void_p),
                ('latent_pose', C.c_void_p), ('dlp', C.c_int), ('betas', C.c_void_p), ('nb', C.c_int),
                ('latent_motion', C.c_void_p), ('prior_mu', C.c_void_p), ('prior_var', C.c_void_p), ('S', C.c_int), ('dz', C.c_int),
                ('floor', C.c_void_p), ('prev_tail', C.c_void_p), ('prev_betas', C.c_void_p), ('prev_floor', C.c_void_p),
                ('obs_j2d', C.c_void_p), ('smpl2op', C.c_void_p), ('op_mask', C.c_void_p), ('cam_f', C.c_void_p), ('cam_c', C.c_void_p),
                ('sigma', C.c_float), ('obs_j3d', C.c_void_p), ('obs_v3d', C.c_void_p), ('obs_floor', C.c_void_p), ('overlap', C.c_void_p),
                ('w', C.c_float * FIT_NTERMS), ('nsteps', C.c_float),
                ('terms', C.c_void_p), ('loss', C.c_void_p),
                ('g_cam_jtr', C.c_void_p), ('g_cam_verts', C.c_void_p), ('g_pri_joints', C.c_void_p), ('g_ro_joints', C.c_void_p),
                ('g_contacts_conf', C.c_void_p), ('g_latent_pose', C.c_void_p), ('g_betas', C.c_void_p), ('g_latent_motion', C.c_void_p),
                ('g_prior_mu', C.c_void_p), ('g_prior_var', C.c_void_p), ('g_floor', C.c_void_p),
                ('g_prev_tail', C.c_void_p), ('g_prev_betas', C.c_void_p), ('g_prev_floor', C.c_void_p), ('partial', C.c_void_p),
                ('gmm_nll', C.c_void_p), ('gmm_gx', C.c_void_p), ('gmm_w', C.c_float), ('gmm_D', C.c_int), ('gmm_nseg', C.c_int),
                ('gmm_g', C.c_void_p * 4), ('gmm_seg_width', C.c_int * 4), ('gmm_g_stride', C.c_int * 4), ('gmm_g_acc', C.c_int * 4),
                ('gmm_total', C.c_void_p)]


class FitPreArgs(C.Structure):
    """ha_fit_pre_args (include/humor_amd.h)."""
    _fields_ = ([('B', C.c_int)] + [(n, C.c_void_p) for n in (
        'floor', 'trans0', 'root0', 'pose0', 'jcam', 'trans_vel', 'joints_vel', 'root_orient_vel',
        'past_in', 'trans_p', 'root_p', 'joints_p', 'c2p_R', 'c2p_t', 'root_height',
        'g_past_in', 'g_trans_p', 'g_root_p', 'g_joints_p', 'g_c2p_R', 'g_c2p_t', 'g_root_height',
        'g_floor', 'g_trans0', 'g_root0', 'g_pose0', 'g_jcam', 'g_trans_vel', 'g_joints_vel', 'g_root_orient_vel')]
                + [('jcam_stride', C.c_int)] + [(n, C.c_void_p) for n in (
                    'add_floor', 'add_pose0', 'add_trans_vel', 'add_joints_vel', 'add_root_orient_vel')])


class RolloutPostArgs(C.Structure):
    """ha_rollout_post_args (include/humor_amd.h)."""
    _fields_ = [('B', C.c_int), ('S', C.c_int)] + [(n, C.c_void_p) for n in (
        'world', 'trans0', 'root0', 'pose0', 'joints0', 'c2p_R', 'c2p_t',
        'trans', 'root_orient', 'pose_body', 'joints', 'contacts_conf', 'contacts', 'cam_trans', 'cam_root_orient',
        'g_trans', 'g_root_orient', 'g_pose_body', 'g_joints', 'g_contacts_conf', 'g_cam_trans', 'g_cam_root_orient',
        'g_world', 'g_trans0', 'g_root0', 'g_pose0', 'g_joints0', 'g_c2p_R', 'g_c2p_t', 'partial')]


class RigidImageArgs(C.Structure):
    """ha_rigid_image_args (include/humor_amd.h)."""
    _fields_ = [('N', C.c_int), ('J', C.c_int), ('V', C.c_int)] + [(n, C.c_void_p) for n in (
        'joints', 'verts', 'root', 'trans', 'root2', 'trans2', 'joints2', 'verts2', 'g_joints2', 'g_verts2',
        'g_joints', 'g_verts', 'g_root', 'g_trans', 'g_root2', 'g_trans2', 'g_joints_add', 'g_verts_add')]


class GmmArgs(C.Structure):
    """ha_gmm_args (include/humor_amd.h)."""
    _fields_ = ([('B', C.c_int), ('K', C.c_int), ('D', C.c_int), ('nseg', C.c_int), ('seg', C.c_void_p * 4), ('seg_width', C.c_int * 4),
                 ('seg_stride', C.c_int * 4)] + [(n, C.c_void_p) for n in ('means', 'Linv', 'LinvT', 'cst', 'lp', 'gpart', 'nll', 'g_x')])


_SIGS = {
    'ha_last_error': (C.c_char_p, []),
    'ha_abi_version': (C.c_int, []),
    'ha_device_arch': (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    'ha_tune_set': (C.c_int, [C.c_char_p, C.c_int]),
    'ha_debug_cu_poison': (C.c_int, [C.c_uint, C.POINTER(C.c_uint), C.c_void_p]),
    'ha_smpl_model_create': (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6),
    'ha_smpl_model_destroy': (C.c_int, [C.c_void_p]),
    'ha_smpl_model_info': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    'ha_smpl_model_define_subset': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    'ha_smpl_forward': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_void_p]),
    'ha_smpl_workspace': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'ha_smpl_backward': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_void_p]),
    'ha_smpl_forward_split': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3),
    'ha_smpl_backward_split': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 6),
    'ha_smpl_forward_parts': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 3),
    'ha_smpl_backward_parts': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
                               + [C.c_void_p] * 10),
    'ha_seq_sum_add': (C.c_int, [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5),
    'ha_humor_rollout_backward_ex': (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 8 + [C.c_void_p]),
    'ha_smpl_backward_dense_workspace': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    'ha_smpl_backward_dense': (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 10 + [C.c_void_p]),
    'ha_lbs_skin': (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_void_p]),
    'ha_rodrigues_fwd': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ha_rodrigues_bwd': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ha_rotmat_to_aa_fwd': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ha_rotmat_to_aa_bwd': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ha_rot6d_to_rotmat_fwd': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ha_rot6d_to_rotmat_bwd': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ha_rot9d_to_rotmat_fwd': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ha_rot9d_to_rotmat_bwd': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ha_humor_net_create': (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(MlpDesc), C.POINTER(MlpDesc)]),
    'ha_humor_net_destroy': (C.c_int, [C.c_void_p]),
    'ha_humor_rollout_workspace': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    'ha_humor_rollout_forward': (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_void_p]),
    'ha_humor_rollout_sample': (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_void_p]),
    'ha_humor_rollout_backward': (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_void_p]),
    'ha_humor_net_set_option': (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    'ha_humor_persist_status': (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint), C.POINTER(C.c_int64)]),
    'ha_humor_persist_ack': (C.c_int, [C.c_void_p]),
    'ha_mlp_create': (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(MlpDesc), C.c_int, C.c_float]),
    'ha_mlp_destroy': (C.c_int, [C.c_void_p]),
    'ha_mlp_workspace': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]),
    'ha_mlp_forward': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ha_mlp_backward': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ha_fit_loss': (C.c_int, [C.POINTER(FitArgs), C.c_void_p]),
    'ha_fit_pre_forward': (C.c_int, [C.POINTER(FitPreArgs), C.c_void_p]),
    'ha_fit_pre_backward': (C.c_int, [C.POINTER(FitPreArgs), C.c_void_p]),
    'ha_rollout_post_forward': (C.c_int, [C.POINTER(RolloutPostArgs), C.c_void_p]),
    'ha_rollout_post_backward': (C.c_int, [C.POINTER(RolloutPostArgs), C.c_void_p]),
    'ha_gmm_nll': (C.c_int, [C.POINTER(GmmArgs), C.c_void_p]),
    'ha_rigid_image_forward': (C.c_int, [C.POINTER(RigidImageArgs), C.c_void_p]),
    'ha_rigid_image_backward': (C.c_int, [C.POINTER(RigidImageArgs), C.c_void_p]),
    'ha_lbfgs_coeffs': (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ha_lbfgs_gram': (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ha_lbfgs_gram_workspace': (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    'ha_lbfgs_pair_coeffs': (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ha_lbfgs_scalars': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ha_chamfer_forward': (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_void_p]),
    'ha_chamfer_backward': (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 7 + [C.c_void_p]),
}


class HumorAmdError(RuntimeError):
    pass


class Lib:
    """Thin checked wrapper: every C entry point becomes a method that raises HumorAmdError on non-zero status."""

    def __init__(self, path, emulator=False):
        if not os.path.exists(path):
            raise HumorAmdError(
                f'{path} not found: build the gfx950 extension first (python -m humor_amd.build, or '
                f'__graft_entry__.build()).  humor_amd has no CPU/PyTorch fallback for its kernels.')
        self.path = path
        self.emulator = emulator
        self._dll = C.CDLL(path)
        missing = []
        for name, (res, args) in _SIGS.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
        self.missing = missing
        if self._dll.ha_abi_version() != ABI_VERSION:
            raise HumorAmdError(f'{path}: ABI version mismatch')

    def exports(self, name):
        return name not in self.missing

    def call(self, name, *args):
        rc = getattr(self._dll, name)(*args)
        if rc != HA_OK:
            msg = self._dll.ha_last_error()
            raise HumorAmdError(f'{name} failed (status {rc}): {msg.decode() if msg else "?"}')

    def device_arch(self, device):
        buf = C.create_string_buffer(128)
        self.call('ha_device_arch', int(device), buf, 128)
        return buf.value.decode()


_lib = None


def load(path, emulator=False):
    return Lib(path, emulator=emulator)


def get_lib():
    """The product library (gfx950 build).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        # HUMOR_AMD_LIB: another gfx950 build of the same sources (A/B measurements of kernel variants); never a fallback
        _lib = Lib(os.environ.get('HUMOR_AMD_LIB') or DEFAULT_LIB)      # (an empty value means unset)
        # HUMOR_AMD_ROLLOUT_PERSIST=0: launch-chain roll-out only.  The persistent roll-out kernels need one resident block on every CU;
        # two PROCESSES sharing one GPU (the 2-rank validation runs on a 1-GPU box) can starve each other's launches until their
        # bounded waits run out.  One process per GPU -- the deployment this package is built for -- needs no setting.
        v = os.environ.get('HUMOR_AMD_ROLLOUT_PERSIST')
        if v:                                                            # (an empty value means unset, as for HUMOR_AMD_LIB)
            _lib.call('ha_tune_set', b'rollout_persist', int(v))
        # HUMOR_AMD_CU_POISON=1 (test tier, tools/nan_hunt.py): LDS and vector registers are not cleared between kernels -- a kernel that
        # reads a word it never wrote sees the previous kernel's data, which differs from box to box.  With this set every kernel launch of
        # the library is preceded by a kernel that fills them with NaN patterns (another value = that bit pattern; humor_amd/csrc/debug.hip).
        v = os.environ.get('HUMOR_AMD_CU_POISON')
        if v and int(v, 0):
            _lib.call('ha_tune_set', b'cu_poison', int(v, 0) if int(v, 0) < 2 ** 31 else int(v, 0) - 2 ** 32)
    return _lib


def ptr(t):
    """Raw data pointer of a contiguous tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), 'humor_amd kernels need contiguous tensors'
    return C.c_void_p(t.data_ptr())


def stream_ptr(t):
    """The current HIP stream of the tensor's device as void* (0 for host/emulator tensors)."""
    if t.is_cuda:
        import torch
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return C.c_void_p(0)
