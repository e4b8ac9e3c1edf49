#!/usr/bin/env python
"""Where does the time of a stage-3 phase go?  MotionOptimizer.run on the bench's C4 problem with k outer iterations per phase (as bench.py's
lbfgs_profile), one device synchronise per OUTER iteration: milliseconds and closure evaluations of every outer iteration of the tune-init /
frozen-init / refine phases (one-off costs -- hipGraph capture, the first step of a new optimiser -- show as the first iteration of a phase),
and the fused optimiser's host-side timeline per phase.   usage: lbfgs_phase_profile.py [k] [speculate 0/1]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                        # noqa: E402
import bench                                        # noqa: E402
from humor_amd import lbfgs as L                    # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
spec = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device('cuda:0')
import tempfile                                     # noqa: E402
from humor_amd import synth                         # noqa: E402
npz = synth.write_smplh_npz(os.path.join(tempfile.mkdtemp(), 'model.npz'), seed=0)
orig_init = L.LBFGS.__init__
made = []


def init(self, *a, **kw):
    orig_init(self, *a, **kw)
    self.speculate = bool(spec)
    self.profile = {}
    made.append(self)


L.LBFGS.__init__ = init
for rep in range(2):          # (second repetition: allocator and graphs of the process warm)
    opt = bench.build_optimizer(dev, npz, bench.B_SEQ, use_graphs='auto')
    opt.stage3_tune_init_freeze_start, opt.stage3_tune_init_freeze_end = k, 2 * k
    opt.iter_log = []
    del made[:]
    obs, _ = bench.make_problem(bench.B_SEQ, bench.T_SEQ, seed=100, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.run(obs, data_fps=30, lr=1.0, num_iter=[3 * k, 3 * k, 3 * k], lbfgs_max_iter=20)
    torch.cuda.synchronize()
    print(f'--- repetition {rep}: whole run {time.perf_counter() - t0:.3f} s, speculate={spec}')
    prev_t, prev_e, cur = None, None, None
    rows = {}
    for name, t, e in opt.iter_log:
        if prev_t is not None:
            rows.setdefault(name, []).append((1e3 * (t - prev_t), e - prev_e))
        prev_t, prev_e = t, e
    for name, r in rows.items():
        ms = [x[0] for x in r]
        ev = [x[1] for x in r]
        tail = r[1:] if len(r) > 1 else r
        print(f'{name}: ms per outer iteration {[round(x, 1) for x in ms]}  evaluations {ev}  | without the first: '
              f'{sum(x[0] for x in tail) / max(1, sum(x[1] for x in tail)):.3f} ms per evaluation')
    for o in made[-3:]:
        tot = sum(o.profile.values())
        print('   optimiser host timeline (s):', {kk: round(v, 4) for kk, v in o.profile.items()}, 'speculative issued / rolled back', o.spec_stats)
