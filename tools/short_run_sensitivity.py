#!/usr/bin/env python
"""How sensitive is a short L-BFGS fitting run to last-bit changes of its inputs?  Runs the c2 short-run problem of tests/fitting_checks.py
as is and with the observed joints perturbed by one ulp (random signs), prints both loss traces' relative deviation from the reference
trace and from each other.  usage: short_run_sensitivity.py [c2|c4]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fitting_checks as FC                       # noqa: E402
from conftest import golden                       # noqa: E402
from humor_amd import _lib, synth                 # noqa: E402
from oracle import closure_cases as CC            # noqa: E402


def run(name, kind, perturb_seed, dev, lib, npz):
    gd = golden(f'closure_{name}.npz')
    B, T, ov = int(gd['B']), int(gd['T']), int(gd['ov'])
    opt = FC.build(lib, dev, kind, B, T, npz, state_dict=synth.contractive_state_dict(int(gd['weight_seed'])))
    obs = CC.make_case(kind, B, T, seed=2, ov=None if ov < 0 else ov)['obs']
    if 'run_obs_joints3d' in gd.files:
        obs['joints3d'] = torch.from_numpy(gd['run_obs_joints3d'])
    if perturb_seed is not None:
        g = torch.Generator().manual_seed(perturb_seed)
        for k in obs:
            if obs[k].dtype == torch.float32:
                sign = (torch.rand(obs[k].shape, generator=g) > 0.5).float() * 2 - 1
                obs[k] = torch.where(torch.isfinite(obs[k]), obs[k] * (1.0 + sign * float(os.environ.get('PERT', 2.0 ** -23))), obs[k])
    obs = {k: v.clone().to(dev) for k, v in obs.items()}
    if os.environ.get('NO_GRAPHS'):
        opt.use_graphs = False
    opt.loss_trace = []
    opt.run(obs, data_fps=30, lr=1.0, num_iter=[int(x) for x in gd['run_num_iter']], lbfgs_max_iter=5)
    return np.array(opt.loss_trace, dtype=np.float64), gd['run_trace']


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'c2'
    kind = {'c2': 'amass', 'c3': 'rgb', 'c4': 'rgb'}[name]
    dev = torch.device('cuda:0')
    lib = _lib.get_lib()
    npz = synth.write_smplh_npz('/tmp/model_srs.npz', seed=0)
    base, ref = run(name, kind, None, dev, lib, npz)
    n = min(len(base), len(ref))
    rel = lambda a, b: np.abs(a[:n, 1] - b[:n, 1]) / np.abs(b[:n, 1])
    print('stage of every evaluation  ', base[:n, 0].astype(int).tolist())
    print('ours vs reference          ', np.array2string(rel(base, ref), precision=1))
    for s in ((1,) if os.environ.get('NO_GRAPHS') else (1, 2, 3)):
        pert, _ = run(name, kind, s, dev, lib, npz)
        m = min(n, len(pert))
        print(f'ours vs ours + 1 ulp ({s})   ', np.array2string(np.abs(pert[:m, 1] - base[:m, 1]) / np.abs(base[:m, 1]), precision=1))


if __name__ == '__main__':
    main()
