"""Roll-out forward / forward+backward at the metric's batch: launch chain vs the persistent one-launch forward (ha_tune_set
"rollout_persist" 0 / 1 / 3), HIP-event timed on the launch stream; also B = 4 (the per-GPU share of the 8-GPU strong-scaling job)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humor_amd import _lib, synth
from humor_amd.humor_model import HumorModel

dev = torch.device('cuda:0')
lib = _lib.get_lib()
hm = HumorModel(in_rot_rep='mat', out_rot_rep='aa', latent_size=48, model_data_config='smpl+joints+contacts', steps_in=1)
hm.load_state_dict(synth.contractive_state_dict(0))
hm = hm.to(dev).eval()
for p in hm.parameters():
    p.requires_grad_(False)


def ev(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

QUICK = len(sys.argv) > 1 and sys.argv[1] == 'quick'      # the metric's batch, persistent path only
res = {}
for B, S in (((32, 59),) if QUICK else ((32, 59), (4, 59), (32, 119))):
    past = torch.randn(B, 339, device=dev, requires_grad=True)
    z = torch.randn(B, S, 48, device=dev, requires_grad=True)

    def fwd():
        with torch.no_grad():
            hm.roll_out(past, None, S, z_seq=z, return_prior=True)

    def fwd_noprior():
        with torch.no_grad():
            hm.roll_out(past, None, S, z_seq=z, return_prior=False)

    def fb():
        past.grad = None; z.grad = None
        out, (pm, pv) = hm.roll_out(past, None, S, z_seq=z, return_prior=True)
        (out['trans'].sum() + out['joints'].sum() + pm.sum()).backward()
    for knob, bwd in (((1, 1),) if QUICK else ((0, 0), (1, 0), (1, 1), (3, 1))):
        lib.call('ha_tune_set', b'rollout_persist', knob)
        lib.call('ha_tune_set', b'rollout_persist_bwd', bwd)
        r = {'fwd_ms': round(ev(fwd), 4), 'fwd_noprior_ms': round(ev(fwd_noprior), 4), 'fwd_bwd_ms': round(ev(fb), 4)}
        r['us_per_step_decoder_chain'] = round(1e3 * r['fwd_noprior_ms'] / S, 2)
        res[f'{B}x{S} persist={knob} bwd={bwd}'] = r
        print(B, S, knob, bwd, r, flush=True)
lib.call('ha_tune_set', b'rollout_persist', 1)
lib.call('ha_tune_set', b'rollout_persist_bwd', 1)
print(json.dumps(res))
