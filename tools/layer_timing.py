#!/usr/bin/env python
"""Phase timestamps inside mlp_layer_kernel (profiling build with -DHA_LAYER_TIMING, see tools/microbench/README).
Prints, for the most recent launches, cycles from kernel entry of (block 0, wave 0) to each phase."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from humor_amd import _lib, synth                      # noqa: E402
from humor_amd.humor_model import HumorModel           # noqa: E402


def main():
    B, S = int(os.environ.get('B', 32)), 4
    dev = torch.device('cuda:0')
    lib = _lib.load(os.path.join(ROOT, 'tools', 'microbench', 'libhumor_amd_timing.so'))
    lib.call('ha_tune_set', b'layer_acc', int(os.environ.get('ACC', 0)))
    hm = HumorModel(in_rot_rep='mat', out_rot_rep='aa', model_data_config='smpl+joints+contacts', _lib_override=lib)
    hm.load_state_dict(synth.humor_state_dict(seed=0))
    hm = hm.to(dev).eval()
    past = torch.randn(B, 339, device=dev, requires_grad=True)
    z = torch.randn(B, S, 48, device=dev, requires_grad=True)
    for it in range(3):
        out, (pm, pv) = hm.roll_out(past, None, S, z_seq=z, return_prior=True)
        (out['trans'].sum() + pm.sum()).backward()
    torch.cuda.synchronize()
    fn = lib._dll.ha_debug_layer_timing
    buf = (C.c_ulonglong * 640)()
    n = C.c_uint()
    fn(buf, C.byref(n))
    print('launches', n.value)
    names = ['entry', 'loads issued', 'loads landed', 'GN done', 'MFMA+LDS wr', 'barrier 1', 'stored', 'barrier 2']
    last = n.value
    print('slot  blocks   ' + '  '.join(f'{x:>12s}' for x in names[1:]) + '   entry-to-entry us (100 MHz wall clock)')
    prev = None
    for k in range(max(0, last - 48), last):
        r = [buf[(k & 63) * 10 + i] for i in range(10)]
        nb0 = r[9] >> 32
        gap = '' if prev is None else f'{(r[8] - prev) / 100.0:10.2f}'
        prev = r[8]
        print(f'{k:4d}  ({nb0:4d})   ' + '  '.join(f'{(r[i] - r[0]):12d}' for i in range(1, 8)) + '   ' + gap)


if __name__ == '__main__':
    main()
