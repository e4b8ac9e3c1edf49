#!/usr/bin/env python
"""Sustained (power-steady) launch time of the LBS kernel at cache-free size against its own copy-only mode and a plain device copy:
80 back-to-back launches each, mean of the last 30 (the first ~20 ride the power-management transient, tools/skin_jitter.py).
usage: python tools/skin_sustained.py [N]"""
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from humor_amd import _lib, synth                      # noqa: E402
from humor_amd.body_model import BodyModel             # noqa: E402

V, J = 6890, 52


def series(fn, n=80):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 30720
    dev = torch.device('cuda:0')
    lib = _lib.get_lib()
    npz = synth.write_smplh_npz(os.path.join(tempfile.mkdtemp(), 'm.npz'))
    h = BodyModel(npz, num_betas=16)._handle_for(dev)
    vp = torch.randn(N * V * 3 + 4, device=dev)
    A = torch.randn(N, J, 12, device=dev)
    tr = torch.randn(N, 3, device=dev)
    out = torch.empty(N, V, 3, device=dev)
    st = _lib.stream_ptr(out)
    nbytes = N * (V * 24 + J * 48)
    src = vp[:N * V * 3]
    dst = out.view(-1)

    def report(tag, ts, bytes_):
        first, hump, tail = min(ts[:3]), max(ts[:20]), sum(ts[-30:]) / 30
        print(f'{tag:<34} first {first:7.1f} us ({bytes_ / first / 1e3:5.0f} GB/s)  hump {hump:7.1f}  sustained {tail:7.1f} us = {bytes_ / tail / 1e3:5.0f} GB/s '
              f'= {bytes_ / tail / 1e3 / 8000:.3f} of 8 TB/s', flush=True)

    for rnd in range(2):
        for var, tag in ((-1, 'lbs_skin (shipped variant)'), (13, 'lbs_skin copy-only mode (variant 13)'), (21, 'lbs_skin uniform-bone mode (21)')):
            lib.call('ha_tune_set', b'skin_variant', var)
            ts = series(lambda: lib.call('ha_lbs_skin', h.ptr, N, _lib.ptr(vp), _lib.ptr(A), _lib.ptr(tr), _lib.ptr(out), st))
            report(tag, ts, nbytes)
            torch.cuda.synchronize()
            import time
            time.sleep(1.0)
        lib.call('ha_tune_set', b'skin_variant', -1)
        ts = series(lambda: dst.copy_(src))
        report('torch device copy', ts, 2 * N * V * 12)
        time.sleep(1.0)
    os.system('rocm-smi --showpower --showclocks 2>/dev/null | grep -i -E "power|sclk|mclk|fclk" | head -6')


if __name__ == '__main__':
    main()
