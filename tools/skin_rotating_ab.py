#!/usr/bin/env python
"""LBS skinning kernel at the metric's batch (N = 1920) on rotating operand sets (HBM figure), launch variants side by side:
ha_tune_set("skin_variant", v): bits 0-1 waves per block 4 << b, +4 non-temporal stores, 13 = the kernel's copy-only mode.
+32 = the weight-stationary form (bit 0: 8 waves per block, upper bits (>> 6): pairs per block).
usage: skin_rotating_ab.py [N] [sets] [variants, comma separated]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                        # noqa: E402
from humor_amd import _lib, synth                   # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
    sets = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device('cuda:0')
    npz = synth.write_smplh_npz('/tmp/model_sra.npz', seed=0)
    lib = _lib.get_lib()
    for rnd in range(2):
        for v in ([int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else (-1, 4, 5, 6, 1, 13)):
            lib.call('ha_tune_set', b'skin_variant', v)
            r = bench.skin_roofline(dev, npz, N=N, rotate=sets)
            print(f'round {rnd} variant {v:3d}: {r["avg_launch_us"]:7.2f} us  {r["achieved"]:7.1f} GB/s  {r["frac"]:.3f}')
    lib.call('ha_tune_set', b'skin_variant', -1)


if __name__ == '__main__':
    main()
