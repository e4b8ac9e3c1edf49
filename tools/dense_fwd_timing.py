#!/usr/bin/env python
"""Event-timed dense SMPL forward (6890 vertices, MFMA blend + streaming skinning) at several N.  usage: dense_fwd_timing.py [N ...]"""
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                           # noqa: E402
from humor_amd import synth                            # noqa: E402
from humor_amd.body_model import BodyModel             # noqa: E402


def main():
    Ns = [int(a) for a in sys.argv[1:]] or [1920, 7680, 30720]
    dev = torch.device('cuda:0')
    npz = synth.write_smplh_npz(os.path.join(tempfile.mkdtemp(), 'm.npz'))
    for N in Ns:
        B = N // 60
        root, body, trans = synth.smooth_pose_sequence(B, 60, seed=1)
        args = dict(root_orient=root.reshape(N, 3).to(dev), pose_body=body.reshape(N, 63).to(dev), trans=trans.reshape(N, 3).to(dev),
                    betas=torch.randn(N, 16, device=dev))
        for name, algo in (('auto: fused blend + skin (algo 3)', 0), ('blend, then lbs_skin (algo 2)', 2)):
            bm = BodyModel(npz, num_betas=16, use_vtx_selector=True, algo=algo)
            bm.fused_min_frames = 0
            with torch.no_grad():
                ms = bench.time_events(lambda: bm(**args), iters=20, warm=10)
            print(f'N={N}: dense SMPL forward, {name}: {ms:.4f} ms = {N * 6890 / ms / 1e6:.1f} G verts/s', flush=True)


if __name__ == '__main__':
    main()
