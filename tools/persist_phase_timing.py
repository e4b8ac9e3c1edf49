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
def spins(label):
    h = (C.c_uint * 8)()
    f = lib._dll.ha_debug_persist_spins
    f.restype = C.c_int
    assert f(h, 1) == 0
    h = list(h)
    n = max(1, sum(h))
    print(label, 'sweeps by number of extra polls (0, 1, .. 6, >= 7):', h, ' mean extra polls %.2f' % (sum(i * v for i, v in enumerate(h)) / n))


spins('forward:')
buf = (C.c_ulonglong * (2 * 8 * 24))()
fn = lib._dll.ha_debug_persist_timing
fn.restype = C.c_int
assert fn(buf) == 0
both = np.array(list(buf), dtype=np.int64).reshape(2, 8, 24)
names = ['top', 'L0 mma', 'L0 publish', 'L1 sweep', 'L1 GN', 'L1 barrier', 'L1 mma', 'L1 publish', 'L2 sweep', 'L2 GN', 'L2 barrier', 'L2 mma',
         'L2 publish', 'L3 sweep', 'L3 GN', 'L3 barrier', 'L3 mma', 'L3 publish', 'raw sweep', 'raw barrier', 'glue ph1', 'glue barrier', 'glue ph2']
for who, ts in (('member 5 (ordinary CU, layer-3 producer)', both[0]), ('member 0 (writer of the per-sequence results)', both[1])):
    d = np.diff(ts[:, :23], axis=1)
    step = ts[1:, 0] - ts[:-1, 0]
    print(who, '-- cycles per step (s_memtime = shader clock):', step.tolist())
    print('%-12s %s' % ('phase', 'cycles to reach it from the previous phase, steps 8..15 | median'))
    for i, n in enumerate(names[1:]):
        print('%-12s %s | %d' % (n, ' '.join('%5d' % v for v in d[:, i]), int(np.median(d[:, i]))))
    print('sum of medians', int(np.median(d, axis=0).sum()), 'median step', int(np.median(step)))

# ---- adjoint ----------------------------------------------------------------------------------------------------------------
pz = past.clone().requires_grad_(True)
zz = z.clone().requires_grad_(True)
for _ in range(3):
    pz.grad = None; zz.grad = None
    out, (pm, pv) = hm.roll_out(pz, None, S, z_seq=zz, return_prior=True)
    (out['trans'].sum() + out['joints'].sum() + out['root_orient'].sum() + pm.sum()).backward()
torch.cuda.synchronize()
spins('forward + adjoint:')
bufb = (C.c_ulonglong * (8 * 24))()
fb = lib._dll.ha_debug_persist_timing_bwd
fb.restype = C.c_int
assert fb(bufb) == 0
tb = np.array(list(bufb), dtype=np.int64).reshape(8, 24)
bn = ['top', 'dx sweep', 'barrier', 'glue adj', 'barrier', 'L3T mma', 'L3T pub+dz', 'L2 sweep', 'L2 GN adj', 'L2 barrier', 'L2T mma', 'L2T pub+dz',
      'L1 sweep', 'L1 GN adj', 'L1 barrier', 'L1T mma', 'L1T pub+dz', 'L0 sweep', 'L0 GN adj', 'L0 barrier', 'L0T mma', 'L0T pub+dz']
d = np.diff(tb[:, :22], axis=1)
step = tb[1:, 0] - tb[:-1, 0]
print('adjoint, member 5 -- cycles per step:', step.tolist())
for i, n in enumerate(bn[1:]):
    print('%-12s %s | %d' % (n, ' '.join('%5d' % v for v in d[:, i]), int(np.median(d[:, i]))))
print('sum of medians', int(np.median(d, axis=0).sum()), 'median step', int(np.median(step)))

# ---- adjoint glue, finer: wave 0 (rotations) and wave 2 (vector tasks) ------------------------------------------------------------
bufc = (C.c_ulonglong * (2 * 8 * 8))()
fc = getattr(lib._dll, 'ha_debug_persist_timing_glue', None)
if fc is not None:
    fc.restype = C.c_int
    assert fc(bufc) == 0
    tc = np.array(list(bufc), dtype=np.int64).reshape(2, 8, 8)
    for w, names_c, n in ((0, ['start', 'rot set-up', 'body adjoints', 'end of glue (wave 0; the chain runs on lane 21)'], 4),
                          (1, ['start', 'loads + first half', 'dL/dW sums + publish', 'second half + partials', 'sums + owners', 'outputs'], 6)):
        d = np.diff(tc[w][:, :n], axis=1)
        print('adjoint glue, wave %d:' % (0 if w == 0 else 2), ' | '.join('%s %d' % (nm, int(np.median(d[:, i]))) for i, nm in enumerate(names_c[1:])))
