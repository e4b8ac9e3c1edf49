#!/usr/bin/env python
"""Runs N eager stage-3 closure evaluations of the C4 problem after 3 warm-up evaluations (kernel-trace driver for
tools/closure_trace_diff.py).  usage: closure_n.py N"""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from humor_amd import synth
    n = int(sys.argv[1])
    dev = torch.device('cuda:0')
    npz = synth.write_smplh_npz(os.path.join(tempfile.mkdtemp(), 'm.npz'), seed=0)
    fc = bench.FitClosure(dev, npz, 1, 0, None, use_graphs=False)
    for _ in range(3):
        fc.step()
    torch.cuda.synchronize()
    for _ in range(n):
        fc.step()
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
