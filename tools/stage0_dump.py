import os, sys, numpy as np, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fitting_checks as FC
from conftest import golden
from humor_amd import _lib, synth
from oracle import closure_cases as CC
dev = torch.device('cuda:0'); lib = _lib.get_lib()
npz = synth.write_smplh_npz('/tmp/model_s0.npz', seed=0)
gd = golden('closure_c2.npz')
B, T = int(gd['B']), int(gd['T'])
opt = FC.build(lib, dev, 'amass', B, T, npz, state_dict=synth.contractive_state_dict(int(gd['weight_seed'])))
case = CC.make_case('amass', B, T, seed=2)
out = {}
g = torch.Generator().manual_seed(0)
for k in range(4):
    c2 = dict(case); c2['var'] = {n: v.clone() for n, v in case['var'].items()}
    if k:
        c2['var']['trans'] = c2['var']['trans'] + 0.02 * k * torch.randn(c2['var']['trans'].shape, generator=g)
        c2['var']['root_orient'] = c2['var']['root_orient'] + 0.02 * k * torch.randn(c2['var']['root_orient'].shape, generator=g)
    res = FC.eval_stage(opt, c2, 0, dev)
    for n, v in res.items():
        out[f'{k}_{n}'] = v.detach().cpu().numpy()
np.savez(os.path.join(ROOT, 'gpurun_out', 'stage0_' + sys.argv[1] + '.npz'), **out)
print('saved', list(out.keys()))
