"""``ChamferDistance`` with the reference's module interface (humor/utils/chamfer_distance/chamfer_distance.py:15-62):
``dist1, dist2 = ChamferDistance()(xyz1, xyz2)`` -- squared nearest-neighbour distances in both directions, differentiable
w.r.t. both clouds -- evaluated by the gfx950 kernels of humor_amd/csrc/chamfer.hip (GPU tensors only, no CPU fallback)."""
import torch

from . import _lib


class ChamferDistanceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2, lib):
        b, n, _ = xyz1.size()
        _, m, _ = xyz2.size()
        xyz1, xyz2 = xyz1.contiguous().float(), xyz2.contiguous().float()
        dev = xyz1.device
        dist1 = torch.empty(b, n, dtype=torch.float32, device=dev)
        dist2 = torch.empty(b, m, dtype=torch.float32, device=dev)
        idx1 = torch.empty(b, n, dtype=torch.int32, device=dev)
        idx2 = torch.empty(b, m, dtype=torch.int32, device=dev)
        lib.call('ha_chamfer_forward', b, n, _lib.ptr(xyz1), m, _lib.ptr(xyz2), _lib.ptr(dist1), _lib.ptr(idx1), _lib.ptr(dist2),
                 _lib.ptr(idx2), _lib.stream_ptr(xyz1))
        ctx.lib = lib
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.idx = (idx1, idx2)
        return dist1, dist2

    @staticmethod
    def backward(ctx, g1, g2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        b, n, _ = xyz1.size()
        m = xyz2.size(1)
        g1 = torch.zeros(b, n, dtype=torch.float32, device=xyz1.device) if g1 is None else g1.contiguous().float()
        g2 = torch.zeros(b, m, dtype=torch.float32, device=xyz1.device) if g2 is None else g2.contiguous().float()
        gx1, gx2 = torch.empty_like(xyz1), torch.empty_like(xyz2)
        ctx.lib.call('ha_chamfer_backward', b, n, _lib.ptr(xyz1), m, _lib.ptr(xyz2), _lib.ptr(g1), _lib.ptr(idx1), _lib.ptr(g2), _lib.ptr(idx2),
                     _lib.ptr(gx1), _lib.ptr(gx2), _lib.stream_ptr(xyz1))
        return gx1, gx2, None


class ChamferDistance(torch.nn.Module):
    def __init__(self, _lib_override=None):
        super().__init__()
        self._lib = _lib_override

    def forward(self, xyz1, xyz2, return_idx=False):
        lib = self._lib if self._lib is not None else _lib.get_lib()
        if not (xyz1.is_cuda or lib.emulator):
            raise _lib.HumorAmdError('humor_amd.ChamferDistance runs on the GPU only (no CPU fallback)')
        return ChamferDistanceFunction.apply(xyz1, xyz2, lib)


def nearest_indices(xyz1, xyz2, lib=None):
    """(idx1 [b,n], idx2 [b,m]) int32 nearest-neighbour indices (what the reference's kernels store for their backward pass)."""
    lib = lib if lib is not None else _lib.get_lib()
    b, n, _ = xyz1.size()
    m = xyz2.size(1)
    xyz1, xyz2 = xyz1.contiguous().float(), xyz2.contiguous().float()
    dev = xyz1.device
    d1, d2 = torch.empty(b, n, device=dev), torch.empty(b, m, device=dev)
    i1, i2 = torch.empty(b, n, dtype=torch.int32, device=dev), torch.empty(b, m, dtype=torch.int32, device=dev)
    lib.call('ha_chamfer_forward', b, n, _lib.ptr(xyz1), m, _lib.ptr(xyz2), _lib.ptr(d1), _lib.ptr(i1), _lib.ptr(d2), _lib.ptr(i2), _lib.stream_ptr(xyz1))
    return i1, i2
