"""Shared body of the HumorLoss (training path) parity tests: humor_amd.humor_loss.HumorLoss against the fixture the REFERENCE
HumorLoss produced (oracle/make_golden_humor_loss.py).  Tolerances: loss / stats 1e-5 relative (fp32 reductions in a different
order), gradients 1e-3 relative to each tensor's largest entry + 1e-7 absolute; the contact accuracy statistics are counts -> exact."""
import os

import numpy as np
import torch

from conftest import golden
from oracle import humor_loss_cases as HC
from oracle.make_golden_humor_loss import GENDER_SEEDS


def write_models(root):
    from humor_amd import synth
    for gname, seed in GENDER_SEEDS.items():
        os.makedirs(os.path.join(root, gname), exist_ok=True)
        synth.write_smplh_npz(os.path.join(root, gname, 'model.npz'), seed=seed)
    return root


def check_case(name, device, models_root, lib=None):
    from humor_amd.humor_loss import HumorLoss
    G = golden('humor_loss.npz')
    _, B, seed, weights = next(c for c in HC.CASES if c[0] == name)
    case = HC.make_case(B, seed)
    mod = HumorLoss(smpl_batch_size=32, smplh_path=models_root, _lib_override=lib, **weights)
    res = HC.evaluate(mod, case, device)
    keys = [k[len(name) + 1:] for k in G.files if k.startswith(name + '_')]
    assert sorted(keys) == sorted(res.keys()), (sorted(keys), sorted(res.keys()))     # same stats_dict keys as the reference
    worst = {}
    for k in keys:
        ref, got = np.asarray(G[f'{name}_{k}']), np.asarray(res[k])
        if k.startswith('grad_'):
            scale = max(float(np.abs(ref).max()), 1e-12)
            err = float(np.abs(got - ref).max())
            assert err <= 1e-3 * scale + 1e-7, (name, k, err, scale)
            worst[k] = err / scale
        elif 'acc' in k or k == 'stat_kl_anneal_weight':
            assert float(got) == float(ref), (name, k, got, ref)
        else:
            rel = abs(float(got) - float(ref)) / max(abs(float(ref)), 1e-12)
            assert rel <= 1e-5, (name, k, float(got), float(ref))
            worst[k] = rel
    return worst
