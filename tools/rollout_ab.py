#!/usr/bin/env python
"""A/B of the roll-out layer-kernel launch policy (slices per block, waves per block): event-timed forward and
forward+backward of HumorModel.roll_out.  usage: rollout_ab.py B S "spb,finish[,hsum[,acc]]" ...   (spb 0 = default; finish 0 off / 1 auto / 2 forced; hsum / acc 0 / 1)"""
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
    cfgs = [tuple(int(x) for x in a.split(',')) for a in sys.argv[3:]] or [(4, 4)]
    dev = torch.device('cuda:0')
    lib = _lib.get_lib()
    sd = synth.humor_state_dict(seed=0)
    past = torch.randn(B, 339, device=dev, requires_grad=True)
    z = torch.randn(B, S, 48, device=dev, requires_grad=True)
    ref = None
    for cfg in cfgs:
        spb, nw = cfg[0], cfg[1]
        hsum = cfg[2] if len(cfg) > 2 else 1                                       # third number: summed-h write-back (default on)
        acc = cfg[3] if len(cfg) > 3 else 0                                        # fourth number: fp32-atomic accumulate policy (default off)
        lib.call('ha_tune_set', b'layer_acc', acc)
        groups = cfg[4] if len(cfg) > 4 else 0                                     # fifth number: row groups on side streams (0 = auto)
        lib.call('ha_tune_set', b'rollout_groups', groups)
        lib.call('ha_tune_set', b'layer_spb', spb)
        lib.call('ha_tune_set', b'layer_finish', nw if nw in (0, 1, 2) else 1)   # second number: 0 off, 1 auto, 2 forced
        lib.call('ha_tune_set', b'layer_hsum', hsum)
        hm = HumorModel(in_rot_rep='mat', out_rot_rep='aa', model_data_config='smpl+joints+contacts')
        hm.load_state_dict(sd)
        hm = hm.to(dev).eval()

        def fwd():
            with torch.no_grad():
                return hm.roll_out(past, None, S, z_seq=z, return_prior=True)

        def fwdbwd():
            out, (pm, pv) = hm.roll_out(past, None, S, z_seq=z, return_prior=True)
            (out['trans'].sum() + out['joints'].sum() + pm.sum()).backward()

        out, _ = fwd()
        w = out['joints'][:, min(S, 8) - 1].clone()
        if ref is None:
            ref = w
        it = 5 if B * S > 4000 else 10
        tf, tb = timed(fwd, it), timed(fwdbwd, it)
        print(f'B={B} S={S} spb={spb} nw={nw} hsum={hsum} acc={acc} groups={groups}: fwd {tf:8.3f} ms  fwd+bwd {tb:8.3f} ms   max|joints - first cfg| (step<=8) '
              f'{(w - ref).abs().max().item():.2e}', flush=True)


if __name__ == '__main__':
    main()
