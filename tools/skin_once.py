#!/usr/bin/env python
"""Launches the LBS skinning kernel a few times (for rocprofv3 --pmc runs), cycling over `sets` operand sets so that no launch
finds its operands in the 256 MiB Infinity Cache.  usage: skin_once.py <variant> [N] [sets]"""
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
R = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sets = [(torch.randn(N * V * 3 + 4, device=dev), torch.randn(N, J, 12, device=dev), torch.randn(N, 3, device=dev),
         torch.empty(N, V, 3, device=dev)) for _ in range(R)]
lib.call('ha_tune_set', b'skin_variant', var)
for i in range(5 if R == 1 else 3 * R):
    vp, A, tr, out = sets[i % R]
    lib.call('ha_lbs_skin', h.ptr, N, _lib.ptr(vp), _lib.ptr(A), _lib.ptr(tr), _lib.ptr(out), _lib.stream_ptr(out))
# calibration launches for the PMC byte counters: a plain device copy of exactly the v_posed byte count
src = torch.randn(N * V * 3, device=dev)
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)
torch.cuda.synchronize()
print('done', var, N)
