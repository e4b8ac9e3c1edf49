"""ORACLE (test infrastructure only).  Kink flags for the full-tile roll-out parity tests (tests/rollout_checks.check_rollout_full_tiles).

ReLU(GroupNorm(.)) has ~7e5 kinks per sequence and 119-step roll-out; a sequence with one unit within fp32 rounding of its kink has a
gradient that moves by 1e-3..1e-2 between two CORRECT fp32 evaluations (another summation order is enough).  Which sequences are
exposed is a property of the inputs and weights, not of the implementation, but finding them takes many evaluations: the set of
sequences whose ORACLE gradient moves by >= 2e-4 grows with the number of 1-ulp input perturbations tried and saturates (256 x 119:
34 sequences under an fp64 re-evaluation alone, 56 with four perturbations, 68 with eight, 69 with twelve).  That is too slow for the
GPU box's test run, so the union over an fp64 re-evaluation and NPERT perturbations is computed here, once, in the build container, and
committed: tests/golden/rollout_kink_flags.npz (key f'{B}x{S}' -> bool[B], True = stable).

    python -m oracle.make_kink_flags
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from humor_amd import synth                       # noqa: E402
from oracle import closure_cases as CC            # noqa: E402
from oracle import humor_restated as H            # noqa: E402

NPERT = 24
CASES = [(32, 59), (256, 119)]
OUT = os.path.join(ROOT, 'tests', 'golden', 'rollout_kink_flags.npz')


def case_inputs(B, S):
    """The inputs of check_rollout_full_tiles(B, S) (same generator, same order of draws)."""
    import rollout_checks as RC
    g = torch.Generator().manual_seed(1000 + B + S)
    return RC.canonical_state(B, g), torch.randn(B, S, 48, generator=g)


def grads(sd, past, z):
    p, zz = past.detach().clone().requires_grad_(True), z.detach().clone().requires_grad_(True)
    w, (pm, pv) = H.roll_out(sd, p, zz)
    gw, gm, gv = (CC.det_weights(t.shape, ph).to(w.dtype) for t, ph in ((w, 0.1), (pm, 0.2), (pv, 0.3)))
    return [x.detach().double().numpy() for x in torch.autograd.grad((w * gw).sum() + (pm * gm).sum() + (pv * gv).sum(), [p, zz])]


def per_seq_rel(a, b):
    return np.abs(a - b).reshape(a.shape[0], -1).max(axis=1) / max(1.0, np.abs(b).max())


def main():
    sd = synth.contractive_state_dict(0)
    sd64 = {k: v.double() for k, v in sd.items()}
    save = {'npert': NPERT, 'rtol': 2e-4}
    for B, S in CASES:
        past, z = case_inputs(B, S)
        ref = grads(sd, past, z)
        moved = np.zeros(B, dtype=bool)
        g64 = grads(sd64, past.double(), z.double())
        moved |= np.maximum(per_seq_rel(ref[0], g64[0]), per_seq_rel(ref[1], g64[1])) >= 2e-4
        counts = [int(moved.sum())]
        for k in range(NPERT):
            gp = torch.Generator().manual_seed(7 + k)
            pert = lambda v: v * (1.0 + ((torch.rand(v.shape, generator=gp) > 0.5).float() * 2 - 1) * 2.0 ** -23)
            gk = grads(sd, pert(past), pert(z))
            moved |= np.maximum(per_seq_rel(gk[0], ref[0]), per_seq_rel(gk[1], ref[1])) >= 2e-4
            counts.append(int(moved.sum()))
        print(f'{B}x{S}: sequences whose oracle gradient moves >= 2e-4 after fp64 + k perturbations: {counts}')
        save[f'{B}x{S}'] = ~moved
    np.savez_compressed(OUT, **save)
    print(OUT, os.path.getsize(OUT))


if __name__ == '__main__':
    main()
