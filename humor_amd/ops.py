"""Autograd wrappers of the stand-alone rotation kernels of libhumor_amd.so.

  batch_rodrigues(aa)                  replaces humor/utils/transforms.py:139-170
  rot6d_to_rotmat(x)                   replaces humor/utils/transforms.py:201-220
  rotation_matrix_to_angle_axis(R)     replaces humor/utils/transforms.py:243-389
Same names, argument meaning and output shapes as the reference functions; GPU tensors only (no CPU fallback).
"""
import torch

from . import _lib


def _lib_for(t, lib):
    if lib is not None:
        return lib
    lib = _lib.get_lib()
    if not t.is_cuda:
        raise _lib.HumorAmdError('humor_amd.ops run on the GPU only (got a CPU tensor); there is no CPU fallback')
    return lib


class _Rodrigues(torch.autograd.Function):
    @staticmethod
    def forward(ctx, aa, lib):
        aa = aa.contiguous().float()
        R = torch.empty(aa.shape[0], 9, dtype=torch.float32, device=aa.device)
        lib.call('ha_rodrigues_fwd', aa.shape[0], _lib.ptr(aa), _lib.ptr(R), _lib.stream_ptr(aa))
        ctx.lib = lib
        ctx.save_for_backward(aa)
        return R.view(-1, 3, 3)

    @staticmethod
    def backward(ctx, gR):
        aa, = ctx.saved_tensors
        gR = gR.contiguous().float().view(-1, 9)
        g = torch.empty_like(aa)
        ctx.lib.call('ha_rodrigues_bwd', aa.shape[0], _lib.ptr(aa), _lib.ptr(gR), _lib.ptr(g), _lib.stream_ptr(aa))
        return g, None


class _RotToAA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, R, lib):
        R = R.contiguous().float().view(-1, 9)
        aa = torch.empty(R.shape[0], 3, dtype=torch.float32, device=R.device)
        lib.call('ha_rotmat_to_aa_fwd', R.shape[0], _lib.ptr(R), _lib.ptr(aa), _lib.stream_ptr(R))
        ctx.lib = lib
        ctx.save_for_backward(R)
        return aa

    @staticmethod
    def backward(ctx, g_aa):
        R, = ctx.saved_tensors
        g_aa = g_aa.contiguous().float()
        gR = torch.empty_like(R)
        ctx.lib.call('ha_rotmat_to_aa_bwd', R.shape[0], _lib.ptr(R), _lib.ptr(g_aa), _lib.ptr(gR), _lib.stream_ptr(R))
        return gR.view(-1, 3, 3), None


class _Rot6d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lib):
        x = x.contiguous().float()
        R = torch.empty(x.shape[0], 9, dtype=torch.float32, device=x.device)
        lib.call('ha_rot6d_to_rotmat_fwd', x.shape[0], _lib.ptr(x), _lib.ptr(R), _lib.stream_ptr(x))
        ctx.lib = lib
        ctx.save_for_backward(x)
        return R.view(-1, 3, 3)

    @staticmethod
    def backward(ctx, gR):
        x, = ctx.saved_tensors
        gR = gR.contiguous().float().view(-1, 9)
        g = torch.empty_like(x)
        ctx.lib.call('ha_rot6d_to_rotmat_bwd', x.shape[0], _lib.ptr(x), _lib.ptr(gR), _lib.ptr(g), _lib.stream_ptr(x))
        return g, None


class _Rot9d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lib):
        x = x.contiguous().float()
        R = torch.empty(x.shape[0], 9, dtype=torch.float32, device=x.device)
        lib.call('ha_rot9d_to_rotmat_fwd', x.shape[0], _lib.ptr(x), _lib.ptr(R), _lib.stream_ptr(x))
        ctx.lib = lib
        ctx.save_for_backward(x)
        return R.view(-1, 3, 3)

    @staticmethod
    def backward(ctx, gR):
        x, = ctx.saved_tensors
        gR = gR.contiguous().float().view(-1, 9)
        g = torch.empty_like(x)
        ctx.lib.call('ha_rot9d_to_rotmat_bwd', x.shape[0], _lib.ptr(x), _lib.ptr(gR), _lib.ptr(g), _lib.stream_ptr(x))
        return g, None


def rot9d_to_rotmat(x, _lib_override=None):
    """[N,9] -> [N,3,3], the rotation closest to each 3x3 (SVD projection): humor/utils/transforms.py:222-241."""
    return _Rot9d.apply(x.reshape(-1, 9), _lib_for(x, _lib_override))


def rot6d_to_rotmat(x, _lib_override=None):
    """[N,6] (or anything viewable as [-1,3,2]) -> [N,3,3]: humor/utils/transforms.py:201-220."""
    return _Rot6d.apply(x.reshape(-1, 6), _lib_for(x, _lib_override))


def batch_rodrigues(rot_vecs, _lib_override=None):
    """[N,3] axis-angle -> [N,3,3]; theta = ||r + 1e-8||, no small-angle branch (reference semantics)."""
    return _Rodrigues.apply(rot_vecs.reshape(-1, 3), _lib_for(rot_vecs, _lib_override))


def rotation_matrix_to_angle_axis(rotation_matrix, _lib_override=None):
    """[N,3,3] (or [N,3,4]: the translation column is ignored, as in the reference) -> [N,3]."""
    if rotation_matrix.shape[-2:] == (3, 4):
        rotation_matrix = rotation_matrix[..., :3]
    return _RotToAA.apply(rotation_matrix.reshape(-1, 3, 3), _lib_for(rotation_matrix, _lib_override))
