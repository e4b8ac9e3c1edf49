"""Phase timestamps of the wave-per-frame SMPL kernels (profiling build: tools/build_variant.sh stiming -DHA_SMPL_TIMING, run with
HUMOR_AMD_LIB=tools/microbench/libhumor_amd_stiming.so): wave 0 of block 0.  usage: smpl_phase_timing.py [N]"""
import ctypes as C, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from humor_amd import _lib, synth
from humor_amd.body_model import BodyModel
from humor_amd.tables import KEYPT_VERTS

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda:0')
lib = _lib.get_lib()
npz = synth.write_smplh_npz(os.path.join(tempfile.mkdtemp(), 'm.npz'))
bm = BodyModel(npz, num_betas=16, use_vtx_selector=True, vertex_subset=KEYPT_VERTS)
g = torch.Generator().manual_seed(0)
mk = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(dev).requires_grad_(True)
inp = dict(root_orient=mk(N, 3, sc=0.5), pose_body=mk(N, 63, sc=0.4), betas=mk(N, 16), trans=mk(N, 3))
for _ in range(5):
    for v in inp.values():
        v.grad = None
    o = bm(**inp)
    (o.v.sum() + o.Jtr.sum()).backward()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
fn = lib._dll.ha_debug_smpl_timing
fn.restype = C.c_int
assert fn(buf) == 0
t = np.array(list(buf), dtype=np.int64).reshape(2, 16)
fn_names = ['joint_forward', 'A + outputs', 'blend (block)', 'transform + store']
bn_names = ['joint_forward', 'A', 'blend (block)', 'transform, gA atomics', 'dcoeff rows', 'dcoeff rounds + dense part', 'chain backward', 'pose grad',
            'betas grad', 'transl grad']
print('N =', N, ' forward: total', t[0, 4] - t[0, 0], 'cycles')
for i, n in enumerate(fn_names):
    print('  %-28s %7d' % (n, t[0, i + 1] - t[0, i]))
print('backward: total', t[1, 10] - t[1, 0], 'cycles')
for i, n in enumerate(bn_names):
    print('  %-28s %7d' % (n, t[1, i + 1] - t[1, i]))
