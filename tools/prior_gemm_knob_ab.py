import sys, os
sys.path.insert(0, '/root/repo')
import torch
from humor_amd import _lib, synth
from humor_amd.humor_model import HumorModel
dev = torch.device('cuda:0'); lib = _lib.get_lib()
hm = HumorModel(in_rot_rep='mat', out_rot_rep='aa', latent_size=48, model_data_config='smpl+joints+contacts', steps_in=1)
hm.load_state_dict(synth.contractive_state_dict(0)); hm = hm.to(dev).eval()
for p in hm.parameters(): p.requires_grad_(False)
def ev(fn, iters=40, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
B, S = 32, 59
past = torch.randn(B, 339, device=dev, requires_grad=True); z = torch.randn(B, S, 48, device=dev, requires_grad=True)
def fb():
    past.grad = None; z.grad = None
    out, (pm, pv) = hm.roll_out(past, None, S, z_seq=z, return_prior=True)
    (out['joints'].sum() + pm.sum() + pv.sum()).backward()
for rep in range(2):
    for name, knobs in (('default', {}), ('gemm_ks=3', {'gemm_ks': 3}), ('gemm_ks=0', {'gemm_ks': 0}), ('gemm_rm=2', {'gemm_rm': 2})):
        for k, v in knobs.items(): lib.call('ha_tune_set', k.encode(), v)
        t = ev(fb)
        for k in knobs: lib.call('ha_tune_set', k.encode(), {'gemm_ks': 2, 'gemm_rm': 0}[k])
        print(f'{name:12s} roll-out 32 x 59 forward + backward (prior incl.): {t:.4f} ms', flush=True)
