"""ORACLE (test infrastructure only).  numpy restatement of the reference's CPU chamfer distance
(humor/utils/chamfer_distance/chamfer_distance.cpp: nnsearch :59-87, gradient :114-187): squared distances in fp32 as
(dx*dx + dy*dy) + dz*dz, first minimum wins.  Pinned against the compiled reference (oracle/_ref, oracle/build_ref.py) by
tests/test_oracle.py::test_chamfer_restatement_matches_compiled_reference."""
import numpy as np


def nnsearch(xyz1, xyz2):
    """xyz1 [b,n,3], xyz2 [b,m,3] float32 -> (dist [b,n] float32, idx [b,n] int32)."""
    xyz1, xyz2 = np.asarray(xyz1, dtype=np.float32), np.asarray(xyz2, dtype=np.float32)
    b, n, _ = xyz1.shape
    dist, idx = np.empty((b, n), np.float32), np.empty((b, n), np.int32)
    for i in range(b):
        for j0 in range(0, n, 512):
            q = xyz1[i, j0:j0 + 512]                                   # [q,3]
            d = xyz2[i][None, :, :] - q[:, None, :]                    # float32 throughout
            sq = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
            k = np.argmin(sq, axis=1)                                  # first occurrence of the minimum
            idx[i, j0:j0 + 512] = k
            dist[i, j0:j0 + 512] = sq[np.arange(q.shape[0]), k]
    return dist, idx


def forward(xyz1, xyz2):
    d1, i1 = nnsearch(xyz1, xyz2)
    d2, i2 = nnsearch(xyz2, xyz1)
    return d1, i1, d2, i2


def backward(xyz1, xyz2, g1, i1, g2, i2):
    xyz1, xyz2 = np.asarray(xyz1, np.float32), np.asarray(xyz2, np.float32)
    gx1, gx2 = np.zeros_like(xyz1, dtype=np.float64), np.zeros_like(xyz2, dtype=np.float64)
    b = xyz1.shape[0]
    for i in range(b):
        v = (2.0 * g1[i])[:, None] * (xyz1[i] - xyz2[i][i1[i]])
        gx1[i] += v
        np.add.at(gx2[i], i1[i], -v)
        v = (2.0 * g2[i])[:, None] * (xyz2[i] - xyz1[i][i2[i]])
        gx2[i] += v
        np.add.at(gx1[i], i2[i], -v)
    return gx1.astype(np.float32), gx2.astype(np.float32)
