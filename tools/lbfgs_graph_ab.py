"""Real L-BFGS outer iterations of the C4 fit (bench.lbfgs_profile) with the closure launch policy varied: use_graphs = 'auto' (hipGraph replay
for the short closures only -- the default), True (every closure, the full-length stage-3 refine closure included), False (eager).
Since the persistent roll-out a stage-3 closure is ~76 dispatches (was 709): what was a loss for graph replay in round 2 may have turned.
usage: lbfgs_graph_ab.py [k]"""
import json
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                        # noqa: E402
from humor_amd import synth                         # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device('cuda:0')
npz = synth.write_smplh_npz(os.path.join(tempfile.mkdtemp(), 'm.npz'), seed=0)
orig = bench.build_optimizer
for mode in ('auto', True, False, 'auto', True):
    bench.build_optimizer = lambda d, n, B, shard=None, use_graphs=False, T=None, _m=mode: orig(d, n, B, shard=shard, use_graphs=_m, T=T)
    r = bench.lbfgs_profile(dev, npz, k=k)
    print('use_graphs =', mode, '| whole-fit outer it/s', r['whole_fit_outer_iters_per_sec'], '| stage-3 it/s', r.get('stage3_outer_iters_per_sec'),
          '|', {n: (p['outer_iters_per_sec'], p['ms_per_closure_eval']) for n, p in r['phases'].items()}, flush=True)
