#!/usr/bin/env python
"""Row groups on side streams under hipGraph replay: eager launches are host-bound as soon as two chains advance side by side
(~3.5 us per launch on the host), a captured roll-out is not.  Times HumorModel.roll_out fwd+bwd, eager vs torch.cuda.graph replay,
for several group counts.  usage: rollout_graph_groups.py B S [groups ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from humor_amd import _lib, synth                 # noqa: E402
from humor_amd.humor_model import HumorModel      # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    B, S = int(sys.argv[1]), int(sys.argv[2])
    groups = [int(a) for a in sys.argv[3:]] or [1, 2, 4]
    dev = torch.device('cuda:0')
    lib = _lib.get_lib()
    hm = HumorModel(in_rot_rep='mat', out_rot_rep='aa', model_data_config='smpl+joints+contacts')
    hm.load_state_dict(synth.humor_state_dict(seed=0))
    hm = hm.to(dev).eval()
    for p in hm.parameters():
        p.requires_grad_(False)
    past = torch.randn(B, 339, device=dev, requires_grad=True)
    z = torch.randn(B, S, 48, device=dev, requires_grad=True)

    def fwdbwd():
        past.grad = None
        z.grad = None
        out, (pm, pv) = hm.roll_out(past, None, S, z_seq=z, return_prior=True)
        (out['trans'].sum() + out['joints'].sum() + pm.sum()).backward()

    for g in groups:
        lib.call('ha_tune_set', b'rollout_groups', g)
        it = 5 if B * S > 4000 else 10
        t_eager = timed(fwdbwd, it)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fwdbwd()
            fwdbwd()
        torch.cuda.current_stream().wait_stream(side)
        past.grad = None
        z.grad = None
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph):
                out, (pm, pv) = hm.roll_out(past, None, S, z_seq=z, return_prior=True)
                (out['trans'].sum() + out['joints'].sum() + pm.sum()).backward()
            t_graph = timed(graph.replay, it)
            print(f'B={B} S={S} groups={g}: eager fwd+bwd {t_eager:8.3f} ms   hipGraph replay {t_graph:8.3f} ms', flush=True)
        except Exception as e:
            print(f'B={B} S={S} groups={g}: eager fwd+bwd {t_eager:8.3f} ms   capture failed: {type(e).__name__}: {str(e)[:200]}', flush=True)
            torch.cuda.synchronize()
    lib.call('ha_tune_set', b'rollout_groups', 0)


if __name__ == '__main__':
    main()
