#!/usr/bin/env python
"""Why does the C5-size LBS launch time move between 0.89 and 1.07 ms inside one bench.py process?  Per-launch HIP-event times of
ha_lbs_skin at N = 30720 at several points of the bench flow (fresh process / after closures / after the RCCL self-check / after the
L-BFGS profile), with the buffer addresses.  usage: python tools/skin_jitter.py"""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                           # noqa: E402
from humor_amd import _lib, synth                      # noqa: E402
from humor_amd.body_model import BodyModel             # noqa: E402

V, J = 6890, 52


def per_launch(dev, npz, tag, N=30720, n=24, realA=False):
    lib = _lib.get_lib()
    h = BodyModel(npz, num_betas=16)._handle_for(dev)
    vp = torch.randn(N * V * 3 + 4, device=dev)
    A = torch.randn(N, J, 12, device=dev)
    if realA:
        A = A * 0.01
        A[:, :, 0] += 1; A[:, :, 5] += 1; A[:, :, 10] += 1
    tr = torch.randn(N, 3, device=dev)
    out = torch.empty(N, V, 3, device=dev)
    st = _lib.stream_ptr(out)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(n):
        lib.call('ha_lbs_skin', h.ptr, N, _lib.ptr(vp), _lib.ptr(A), _lib.ptr(tr), _lib.ptr(out), st)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]
    s = sorted(ts)
    print(f'{tag:<34} N={N} min {s[0]:7.1f} med {s[n // 2]:7.1f} max {s[-1]:7.1f} us | first 8: ' + ' '.join(f'{t:.0f}' for t in ts[:8]) +
          f' | last 4: ' + ' '.join(f'{t:.0f}' for t in ts[-4:]) + f' | vp {vp.data_ptr():#x} out {out.data_ptr():#x}', flush=True)
    del vp, A, tr, out


def main():
    dev = torch.device('cuda:0')
    npz = synth.write_smplh_npz(os.path.join(tempfile.mkdtemp(), 'm.npz'))
    os.system('rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E "sclk|mclk|fclk|power" | head -8')
    per_launch(dev, npz, 'fresh process')
    per_launch(dev, npz, 'fresh process, again')
    per_launch(dev, npz, 'fresh, near-identity A', realA=True)
    torch.cuda.empty_cache()
    per_launch(dev, npz, 'after empty_cache')
    fc = bench.FitClosure(dev, npz, 1, 0, None, use_graphs=False)
    for _ in range(10):
        fc.step()
    torch.cuda.synchronize()
    per_launch(dev, npz, 'after 10 closures (live)')
    del fc
    torch.cuda.empty_cache()
    per_launch(dev, npz, 'after closures freed')
    r = bench.rccl_selfcheck(dev, npz)
    print('rccl ok', r.get('ok'))
    torch.cuda.empty_cache()
    per_launch(dev, npz, 'after rccl self-check')
    bench.lbfgs_profile(dev, npz, k=2)
    torch.cuda.empty_cache()
    per_launch(dev, npz, 'after lbfgs profile')
    per_launch(dev, npz, 'N=1920 for reference', N=1920, n=40)
    os.system('rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E "sclk|mclk|fclk|power" | head -8')


if __name__ == '__main__':
    main()
