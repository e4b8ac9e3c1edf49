"""Phase timestamps of the persistent roll-out kernel (profiling build: tools/build_variant.sh ptiming -DHA_PERSIST_TIMING, run with
HUMOR_AMD_LIB=tools/microbench/libhumor_amd_ptiming.so): one wave (team 0, member 5, wave 0) over eight consecutive steps."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from humor_amd import _lib, synth
from humor_amd.humor_model import HumorModel

dev = torch.device('cuda:0')
lib = _lib.get_lib()
hm = HumorModel(in_rot_rep='mat', out_rot_rep='aa', latent_size=48, model_data_config='smpl+joints+contacts', steps_in=1)
hm.load_state_dict(synth.contractive_state_dict(0))
hm = hm.to(dev).eval()
B, S = 32, 59
past, z = torch.randn(B, 339, device=dev), torch.randn(B, S, 48, device=dev)
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lib.call('ha_tune_set', b'rollout_persist', variant)
with torch.no_grad():
    for _ in range(3):
        hm.roll_out(past, None, S, z_seq=z)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (2 * 8 * 24))()
fn = lib._dll.ha_debug_persist_timing
fn.restype = C.c_int
assert fn(buf) == 0
both = np.array(list(buf), dtype=np.int64).reshape(2, 8, 24)
names = ['top', 'L0 mma', 'L0 publish', 'L1 sweep', 'L1 GN', 'L1 barrier', 'L1 mma', 'L1 publish', 'L2 sweep', 'L2 GN', 'L2 barrier', 'L2 mma',
         'L2 publish', 'L3 sweep', 'L3 GN', 'L3 barrier', 'L3 mma', 'L3 publish', 'raw sweep', 'raw barrier', 'glue']
for who, ts in (('member 5 (ordinary CU, layer-3 producer)', both[0]), ('member 0 (writer of the per-sequence results)', both[1])):
    d = np.diff(ts[:, :21], axis=1)
    step = ts[1:, 0] - ts[:-1, 0]
    print(who, '-- cycles per step (s_memtime = shader clock):', step.tolist())
    print('%-12s %s' % ('phase', 'cycles to reach it from the previous phase, steps 8..15 | median'))
    for i, n in enumerate(names[1:]):
        print('%-12s %s | %d' % (n, ' '.join('%5d' % v for v in d[:, i]), int(np.median(d[:, i]))))
    print('sum of medians', int(np.median(d, axis=0).sum()), 'median step', int(np.median(step)))
