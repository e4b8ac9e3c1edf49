#!/usr/bin/env python
"""Launches the LBS skinning kernel a few times (for rocprofv3 --pmc runs).  usage: skin_once.py <variant> [N]"""
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from humor_amd import _lib, synth                      # noqa: E402
from humor_amd.body_model import BodyModel             # noqa: E402

V, J = 6890, 52
var = int(sys.argv[1]) if len(sys.argv) > 1 else -1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
dev = torch.device('cuda:0')
lib = _lib.get_lib()
npz = synth.write_smplh_npz(os.path.join(tempfile.mkdtemp(), 'm.npz'))
h = BodyModel(npz, num_betas=16)._handle_for(dev)
vp = torch.randn(N * V * 3 + 4, device=dev)
A = torch.randn(N, J, 12, device=dev)
tr = torch.randn(N, 3, device=dev)
out = torch.empty(N, V, 3, device=dev)
lib.call('ha_tune_set', b'skin_variant', var)
for _ in range(5):
    lib.call('ha_lbs_skin', h.ptr, N, _lib.ptr(vp), _lib.ptr(A), _lib.ptr(tr), _lib.ptr(out), _lib.stream_ptr(out))
# calibration launches for the PMC byte counters: a plain device copy of exactly the v_posed byte count
src = torch.randn(N * V * 3, device=dev)
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)
torch.cuda.synchronize()
print('done', var, N)
