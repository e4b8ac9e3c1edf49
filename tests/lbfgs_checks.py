"""Direct check of the L-BFGS kernels (ha_lbfgs_gram -> ha_lbfgs_pair_coeffs -> d = M^T coef, ha_lbfgs_scalars) against the textbook
two-loop recursion of torch/optim/lbfgs.py evaluated in float64, at history lengths above one wavefront (k > 64: lanes own two rows
of the recurrences) and with the slots in a rotated order (evictions).  Shared by the emulator (CPU tier) and GPU tiers."""
import ctypes as C

import torch

from humor_amd import _lib


def two_loop(S, Y, g, H):
    """d = -H_k g by the two-loop recursion over the pairs (oldest first), float64."""
    k = S.shape[0]
    q = -g.clone()
    ro = [1.0 / (Y[i] @ S[i]) for i in range(k)]
    al = [None] * k
    for i in range(k - 1, -1, -1):
        al[i] = ro[i] * (S[i] @ q)
        q = q - al[i] * Y[i]
    r = q * H
    for i in range(k):
        be = ro[i] * (Y[i] @ r)
        r = r + (al[i] - be) * S[i]
    return r


def check_direction(lib, device, n=300, h=100, k=90, seed=0):
    gen = torch.Generator().manual_seed(seed)
    # a convex quadratic's pairs: y = A s with A SPD (diagonal, condition 100: no n x n matrix at n ~ 1e5) -> y.s > 0
    diag = torch.logspace(0, 2, n, dtype=torch.float64)[torch.randperm(n, generator=gen)]
    S = torch.randn(k, n, generator=gen, dtype=torch.float64)
    Y = S * diag
    g = torch.randn(n, generator=gen, dtype=torch.float64)
    H = float((Y[-1] @ S[-1]) / (Y[-1] @ Y[-1]))
    ref = two_loop(S, Y, g, H)
    # history as the optimiser keeps it: slots in rotated order, the newest pair in `slot`, the gradient in row 2h
    rot = 37
    order = [(rot + i) % h for i in range(k)]                # physical slot of the i-th oldest pair
    slot = order[-1]
    M = torch.zeros(2 * h + 1, n, dtype=torch.float32)
    for i, p in enumerate(order):
        M[p] = S[i].float()
        M[h + p] = Y[i].float()
    M[2 * h] = g.float()
    M = M.to(device)
    Mk = M[:2 * h]
    G = (Mk @ Mk.t()).contiguous()                            # Gram matrix of the stored pairs ...
    G[slot] = 0; G[:, slot] = 0; G[h + slot] = 0; G[:, h + slot] = 0      # ... the new pair's entries come from the Gram pass
    npart = C.c_int64()
    lib.call('ha_lbfgs_gram_workspace', n, 2 * h, C.byref(npart))
    z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)
    part, P, Mg, coef, scal = z(npart.value), z(2 * h, 3), z(2 * h), z(2 * h + 1), z(4)
    st = _lib.stream_ptr(M)
    lib.call('ha_lbfgs_gram', n, 2 * h, _lib.ptr(M), slot, h + slot, 2 * h, _lib.ptr(part), _lib.ptr(P), st)
    P_ref = Mk.double() @ torch.stack([M[slot], M[h + slot], M[2 * h]], 1).double()
    assert (P.double() - P_ref).abs().max().item() <= 2e-5 * P_ref.abs().max().item()
    oarr = (C.c_int32 * k)(*order)
    lib.call('ha_lbfgs_pair_coeffs', h, k, oarr, slot, _lib.ptr(P), _lib.ptr(G), _lib.ptr(Mg), _lib.ptr(coef), _lib.ptr(scal), st)
    d = torch.mv(M.t(), coef).double().cpu()
    ys, yy = scal[:2].tolist()
    assert abs(ys / yy - H) <= 1e-5 * H
    err = (d - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-3, err             # fp32 recurrences over k = 90 pairs of a condition-100 problem against float64
    # the installed Gram rows are the Gram pass's
    assert torch.equal(G[slot], P[:, 0]) and torch.equal(G[:, h + slot], P[:, 1])
    # scalars kernel
    a, b = torch.randn(n, generator=gen).to(device), torch.randn(n, generator=gen).to(device)
    extra = torch.tensor([3.25], device=device)
    out = z(4)
    lib.call('ha_lbfgs_scalars', n, _lib.ptr(a), _lib.ptr(b), _lib.ptr(extra), _lib.ptr(out), st)
    o = out.tolist()
    assert abs(o[0] - float(a.double() @ b.double())) <= 1e-4 * float(a.abs() @ b.abs())
    assert o[1] == float(a.abs().max()) and abs(o[2] - float(a.abs().double().sum())) <= 1e-5 * o[2] and o[3] == 3.25
    return err
