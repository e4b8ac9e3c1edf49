"""Chamfer kernels against the oracle (numpy restatement of the reference's CPU path, itself pinned to the compiled reference in
oracle/_ref): int32 nearest-neighbour indices and squared distances bit-exact (ties included), gradients to fp32 rounding."""
import numpy as np
import torch

from humor_amd.chamfer import ChamferDistance, nearest_indices
from oracle import chamfer_restated as CR


def make_clouds(b, n, m, seed, ties=True):
    g = torch.Generator().manual_seed(seed)
    x1, x2 = torch.randn(b, n, 3, generator=g), torch.randn(b, m, 3, generator=g)
    if ties and m > 12 and n > 4:
        x2[:, 10] = x2[:, 5]          # duplicated candidates: the lower index must win
        x2[:, m - 1] = x2[:, 0]
        x1[:, 3] = x2[:, 5]           # a query sitting exactly on the duplicated point (distance 0 twice)
        x1[:, 4] = x2[:, 0]
    return x1, x2


def check_chamfer(lib, device, b, n, m, seed=0):
    x1c, x2c = make_clouds(b, n, m, seed)
    x1 = x1c.clone().to(device).requires_grad_(True)
    x2 = x2c.clone().to(device).requires_grad_(True)
    d1, d2 = ChamferDistance(_lib_override=lib)(x1, x2)
    i1, i2 = nearest_indices(x1.detach(), x2.detach(), lib)
    r1, ri1, r2, ri2 = CR.forward(x1c.numpy(), x2c.numpy())
    assert np.array_equal(i1.cpu().numpy(), ri1) and np.array_equal(i2.cpu().numpy(), ri2), 'nearest-neighbour indices differ'
    assert np.array_equal(d1.detach().cpu().numpy(), r1) and np.array_equal(d2.detach().cpu().numpy(), r2), 'squared distances differ'
    g = torch.Generator().manual_seed(seed + 1)
    g1, g2 = torch.randn(b, n, generator=g), torch.randn(b, m, generator=g)
    gx1, gx2 = torch.autograd.grad((d1 * g1.to(device)).sum() + (d2 * g2.to(device)).sum(), [x1, x2])
    rg1, rg2 = CR.backward(x1c.numpy(), x2c.numpy(), g1.numpy(), ri1, g2.numpy(), ri2)
    e1 = np.abs(gx1.cpu().numpy() - rg1).max() / max(1.0, np.abs(rg1).max())
    e2 = np.abs(gx2.cpu().numpy() - rg2).max() / max(1.0, np.abs(rg2).max())
    assert e1 < 1e-5 and e2 < 1e-5, (e1, e2)
    return e1, e2
