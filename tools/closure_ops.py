#!/usr/bin/env python
"""Host-side view of one stage-3 closure: ATen op counts (forward+backward) and their host time, plus the Python-level
sections of the forward (dispatch counts per section).  Guides op-count reductions (the eager closure is host-bound in
its PyTorch regions, which is what bounds the multi-GPU path where the closure is not graph-captured)."""
import collections
import os
import sys
import time

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                             # noqa: E402
from humor_amd import synth                              # noqa: E402


class Counter(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.counts = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        self.counts[str(func.overloadpacket)] += 1
        return func(*args, **(kwargs or {}))


def main():
    dev = torch.device('cuda:0')
    npz = synth.write_smplh_npz('/tmp/model_ops.npz', seed=0)
    fc = bench.FitClosure(dev, npz, 1, 0, None, use_graphs=False)
    for _ in range(3):
        fc.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fc.step()
    torch.cuda.synchronize()
    print(f'eager closure wall: {(time.perf_counter() - t0) * 100:.2f} ms')
    o = fc.opt
    # forward op counts per wrapped method
    sections = collections.OrderedDict()
    names = ['latent2pose', 'pose2latent', 'smpl_results', 'rollout_latent_motion', 'apply_cam2prior', '_halo']
    originals = {}

    def wrap(obj, name):
        fn = getattr(obj, name)
        originals[(obj, name)] = fn

        def w(*a, **k):
            with Counter() as c:
                r = fn(*a, **k)
            sections.setdefault(name, collections.Counter()).update({'calls': 1, 'ops': sum(c.counts.values())})
            return r
        setattr(obj, name, w)
    for n in names:
        wrap(o, n)
    wrap(o.fitting_loss, 'motion_fit')
    for n in ['joints2d_loss', 'overlap_verts_loss', 'joints3d_smooth_loss', 'motion_prior_loss', 'init_motion_prior_loss',
              'bone_length_loss', 'contact_vel_loss', 'contact_height_loss', 'floor_reg_loss', 'joints3d_loss']:
        wrap(o.fitting_loss, n)
    with Counter() as total:
        loss, _ = o._stage3_objective(fc.obs_local, None, fc.prior_params, False, 15, 1.0, fc.og_w, True, 'neutral')
    print('forward dispatches total:', sum(total.counts.values()))
    for k, v in sections.items():
        print(f'  {k:24s} calls {v["calls"]:3d}  dispatches {v["ops"]:5d}')
    print('  top forward ops:', total.counts.most_common(14))
    for (obj, name), fn in originals.items():
        setattr(obj, name, fn)
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        fc.step()
    print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=8, max_name_column_width=50))


if __name__ == '__main__':
    main()
