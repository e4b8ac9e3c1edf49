#!/usr/bin/env python
"""A/B of the LBS skinning kernel launch variants on the GPU (interleaved rounds in one process).
skin_variant bits: 0-1 waves/block = 4 << b, +4 non-temporal stores, +8 copy-only profiling mode, +16 uniform-bone mode.
usage: python tools/skin_ab.py [N ...]"""
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from humor_amd import _lib, synth                      # noqa: E402
from humor_amd.body_model import BodyModel             # noqa: E402

V, J = 6890, 52
VARIANTS = [int(v) for v in os.environ.get('SKIN_VARIANTS', '5,4,6,1,13,21').split(',')]


def main():
    Ns = [int(a) for a in sys.argv[1:]] or [1920, 30720]
    dev = torch.device('cuda:0')
    lib = _lib.get_lib()
    npz = synth.write_smplh_npz(os.path.join(tempfile.mkdtemp(), 'm.npz'))
    h = BodyModel(npz, num_betas=16)._handle_for(dev)
    for N in Ns:
        vp = torch.randn(N * V * 3 + 4, device=dev)
        A = torch.randn(N, J, 12, device=dev)
        tr = torch.randn(N, 3, device=dev)
        out = torch.empty(N, V, 3, device=dev)
        st = _lib.stream_ptr(out)
        nbytes = N * (V * 24 + J * 48)
        res = {}
        for rnd in range(5):
            for var in VARIANTS:
                lib.call('ha_tune_set', b'skin_variant', var)
                for _ in range(3):
                    lib.call('ha_lbs_skin', h.ptr, N, _lib.ptr(vp), _lib.ptr(A), _lib.ptr(tr), _lib.ptr(out), st)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                it = 20
                e0.record()
                for _ in range(it):
                    lib.call('ha_lbs_skin', h.ptr, N, _lib.ptr(vp), _lib.ptr(A), _lib.ptr(tr), _lib.ptr(out), st)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(var, []).append(e0.elapsed_time(e1) / it)
        # plain device copy of the same bytes as a yardstick
        src = torch.empty(N * V * 3, device=dev)
        dst = torch.empty_like(src)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        cp = e0.elapsed_time(e1) / 20
        print(f'N={N}: torch copy of {N * V * 12 / 1e6:.0f} MB: {cp * 1e3:.1f} us = {2 * N * V * 12 / cp / 1e6:.0f} GB/s')
        for var, ts in res.items():
            ts = sorted(ts)
            med = ts[len(ts) // 2]
            print(f'  variant {var}: median {med * 1e3:8.1f} us  min {ts[0] * 1e3:8.1f} us  -> {nbytes / med / 1e6:7.0f} GB/s (algorithmic)')


if __name__ == '__main__':
    main()
