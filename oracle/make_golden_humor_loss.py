"""ORACLE (test infrastructure only).  Evaluates the REFERENCE HumorLoss (humor/losses/humor_loss.py, imported unmodified; its body
models are the reference BodyModel over the restated smplx layer, reading seeded synthetic male / female model files) on the
seeded training-step cases of oracle/humor_loss_cases.py -> tests/golden/humor_loss.npz (loss, every stats entry, gradients
w.r.t. the predicted tensors).  Build container only:  python -m oracle.make_golden_humor_loss"""
import importlib
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from humor_amd import synth                       # noqa: E402
from oracle import humor_loss_cases as HC         # noqa: E402
from oracle import ref_loader                     # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'humor_loss.npz')
GENDER_SEEDS = {'male': 0, 'female': 3}           # the two genders are different synthetic bodies


def write_models(root):
    for gname, seed in GENDER_SEEDS.items():
        os.makedirs(os.path.join(root, gname), exist_ok=True)
        synth.write_smplh_npz(os.path.join(root, gname, 'model.npz'), seed=seed)
    return root


def reference_loss(smpl_batch_size, **weights):
    ref_loader.load()
    mod = importlib.import_module('losses.humor_loss')
    mod.SMPLH_PATH = write_models(tempfile.mkdtemp())
    return mod.HumorLoss(smpl_batch_size=smpl_batch_size, **weights)


def main():
    torch.manual_seed(0)
    out = {}
    for name, B, seed, w in HC.CASES:
        case = HC.make_case(B, seed)
        loss_mod = reference_loss(32, **w)
        res = HC.evaluate(loss_mod, case, torch.device('cpu'))
        for k, v in res.items():
            out[f'{name}_{k}'] = np.asarray(v)
        print(name, 'loss', res['loss'], {k: round(v, 6) for k, v in res.items() if k.startswith('stat_')})
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
