#!/usr/bin/env python
"""cProfile of the eager stage-3 closure (host side): where the Python/dispatch time goes."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                             # noqa: E402
from humor_amd import synth                              # noqa: E402


def main():
    dev = torch.device('cuda:0')
    npz = synth.write_smplh_npz('/tmp/model_hp.npz', seed=0)
    fc = bench.FitClosure(dev, npz, 1, 0, None, use_graphs=False)
    for _ in range(5):
        fc.step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        fc.step()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(28)
    st.sort_stats('cumulative').print_stats(22)


if __name__ == '__main__':
    main()
