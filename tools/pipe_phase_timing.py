#!/usr/bin/env python
"""In-kernel phase timestamps of the pipelined roll-out kernels (build: tools/build_variant.sh ptiming -DHA_PERSIST_TIMING; run with
HUMOR_AMD_LIB=tools/microbench/libhumor_amd_ptiming.so): per role one CU of team 0, steps 8..11, every group.
usage: pipe_phase_timing.py [B] [S] [bwd]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np                                  # noqa: E402
import torch                                        # noqa: E402
from humor_amd import _lib                          # noqa: E402
import rollout_checks as RC                         # noqa: E402

dev = torch.device('cuda:0')
lib = _lib.get_lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bwd = len(sys.argv) > 3 and sys.argv[3] == 'bwd'
hm, _ = RC.make_model(lib, dev, seed=0, contractive=True)
g = torch.Generator().manual_seed(3)
past, z = RC.canonical_state(B, g).to(dev), torch.randn(B, S, 48, generator=g).to(dev)
lib.call('ha_tune_set', b'rollout_pipe_bwd', 1 if bwd else 0)
for _ in range(2):
    p, zz = past.clone().requires_grad_(True), z.clone().requires_grad_(True)
    out, (pm, pv) = hm.roll_out(p, None, S, z_seq=zz, return_prior=True)
    if bwd:
        (RC.world_of(out).square().sum() + pm.sum() + pv.sum()).backward()
torch.cuda.synchronize()
buf = (C.c_uint64 * (2 * 4 * 4 * 8 * 12))()
f = lib._dll.ha_debug_pipe_timing
f.restype = C.c_int
assert f(buf) == 0
ts = np.array(list(buf), dtype=np.float64).reshape(2, 4, 4, 8, 12)[1 if bwd else 0]
NG = (B + 31) // 32
names = ['L0', 'L1', 'L2', 'L3+glue']
print(f'B={B} S={S} groups={NG}; cycles (s_memtime ticks at 100 MHz x 24 = shader cycles at 2.4 GHz are NOT assumed: raw clock64 units)')
for r in range(4):
    t = ts[r]           # [step][group][phase]
    step_len = [(t[k + 1, 0, 0] - t[k, 0, 0]) for k in range(3)]
    print(f'role {names[r]}: step length (first group start to next step) {step_len}')
    nph = 12 if r == 3 else 5
    for gq in range(NG):
        d = [t[1, gq, i + 1] - t[1, gq, i] for i in range(nph - 1)]
        nxt = (t[1, gq + 1, 0] if gq + 1 < NG else t[2, 0, 0]) - t[1, gq, 0]
        print(f'   group {gq}: phase deltas {[int(x) for x in d]}  | iteration {int(nxt)}')
