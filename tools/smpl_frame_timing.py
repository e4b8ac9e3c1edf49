#!/usr/bin/env python
"""Event-timed wave-per-frame SMPL kernels as the fitting closure runs them (64-vertex subset, N = 32 x 60 frames): forward and
forward+backward of BodyModel(vertex_subset=KEYPT_VERTS, use_vtx_selector=True).  usage: smpl_frame_timing.py [N]"""
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from humor_amd import synth                        # noqa: E402
from humor_amd.body_model import BodyModel         # noqa: E402
from humor_amd.tables import KEYPT_VERTS           # noqa: E402


def timed(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
    dev = torch.device('cuda:0')
    npz = synth.write_smplh_npz(os.path.join(tempfile.mkdtemp(), 'm.npz'))
    bm = BodyModel(npz, num_betas=16, use_vtx_selector=True, vertex_subset=KEYPT_VERTS)
    g = torch.Generator().manual_seed(0)
    mk = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(dev).requires_grad_(True)
    inp = dict(root_orient=mk(N, 3, sc=0.5), pose_body=mk(N, 63, sc=0.4), betas=mk(N, 16), trans=mk(N, 3))

    def fwd():
        with torch.no_grad():
            return bm(**inp)

    def fwdbwd():
        for v in inp.values():
            v.grad = None
        o = bm(**inp)
        (o.v.sum() + o.Jtr.sum()).backward()
    tf, tb = timed(fwd), timed(fwdbwd)
    print(f'N={N}: subset SMPL forward {tf:.1f} us, forward+backward {tb:.1f} us (incl. ~6 ATen launches of the autograd glue)')


if __name__ == '__main__':
    main()
