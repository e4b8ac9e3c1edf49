#!/usr/bin/env python
"""BASELINE config C5 (synthetic AMASS-shaped batch=256 seq=120, 6890-vertex SMPL+H): per-kernel roofline figures.
Prints one JSON object: dense SMPL forward verts/s, LBS kernel GB/s (HBM roofline), pose-blend GEMM TFLOP/s (fp32 MFMA
roofline), roll-out fwd / fwd+bwd time and its MLP FLOP rate."""
import json
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from humor_amd import _lib, synth                      # noqa: E402
from humor_amd.body_model import BodyModel             # noqa: E402
from humor_amd.humor_model import HumorModel           # noqa: E402

V, J, NB = 6890, 52, 16
HBM_PEAK, MFMA_F32_PEAK = 8000.0, 157.3               # GB/s, TFLOP/s (MI355X_MICROARCH.md)


def ev(fn, iters=5, warm=2):
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


def measure(B=256, T=120, dev=None):
    N = B * T
    dev = dev if dev is not None else torch.device('cuda:0')
    lib = _lib.get_lib()
    npz = synth.write_smplh_npz(os.path.join(tempfile.mkdtemp(), 'm.npz'))
    bm = BodyModel(npz, num_betas=NB, use_vtx_selector=True)
    h = bm._handle_for(dev)
    g = torch.Generator().manual_seed(0)
    pose = torch.cat([0.3 * torch.randn(N, 66, generator=g), torch.zeros(N, 90)], 1).to(dev)
    betas = torch.randn(N, NB, generator=g).to(dev)
    tr = torch.randn(N, 3, generator=g).to(dev)
    import ctypes as C
    nv, nc = C.c_int64(), C.c_int64()
    lib.call('ha_smpl_workspace', h.ptr, N, 22, C.byref(nv), C.byref(nc))
    ws_v, ws_c = torch.empty(nv.value, device=dev), torch.empty(nc.value, device=dev)
    verts, joints, A = torch.empty(N, V, 3, device=dev), torch.empty(N, J, 3, device=dev), torch.empty(N, J, 12, device=dev)
    st = _lib.stream_ptr(verts)
    p = _lib.ptr

    def dense():
        lib.call('ha_smpl_forward', h.ptr, 0, N, 22, p(pose), p(betas), p(tr), p(verts), p(joints), p(A), p(ws_v), p(ws_c), 2, st)

    def skin():
        lib.call('ha_lbs_skin', h.ptr, N, p(ws_v), p(A), p(tr), p(verts), st)

    def chain_only():
        lib.call('ha_smpl_forward', h.ptr, 0, N, 22, p(pose), p(betas), p(tr), None, p(joints), p(A), p(ws_v), p(ws_c), 2, st)
    # sustained figures: the first ~20 launches at this size ride a power-management transient (profiles/r02_run20_skin_jitter.txt)
    def fused():          # forward-only callers: blend + skin in one kernel, v_posed never written (ha_smpl_forward algo 3)
        lib.call('ha_smpl_forward', h.ptr, 0, N, 22, p(pose), p(betas), p(tr), p(verts), p(joints), p(A), None, p(ws_c), 3, st)
    ms_dense, ms_skin, ms_chain = ev(dense, 10, 10), ev(skin, 20, 20), ev(chain_only, 10, 3)
    ms_fused = ev(fused, 10, 10)
    ms_blend = ms_dense - ms_skin - ms_chain
    Kc = NB + 1 + 21 * 9
    blend_flops = 2.0 * N * Kc * V * 3
    skin_bytes = N * (V * 24 + J * 48)
    res = {'config': f'C5 B={B} T={T} N={N}', 'smpl_dense_fwd_ms': round(ms_dense, 3),
           'smpl_verts_per_sec': round(N * V / (ms_dense * 1e-3), 1),
           'smpl_dense_fwd_fused_ms': round(ms_fused, 3), 'smpl_verts_per_sec_fused_forward_only': round(N * V / (ms_fused * 1e-3), 1),
           'fused_blend_skin_TFLOPs': round(2.0 * N * (NB + 1 + 21 * 9) * V * 3 / (ms_fused - ms_chain) / 1e9, 1),
           'lbs_skin': {'ms': round(ms_skin, 3), 'GBps': round(skin_bytes / ms_skin / 1e6, 1), 'frac_hbm': round(skin_bytes / ms_skin / 1e6 / HBM_PEAK, 4)},
           'pose_blend_mfma': {'ms_est': round(ms_blend, 3), 'TFLOPs': round(blend_flops / ms_blend / 1e9, 1),
                               'frac_fp32_mfma': round(blend_flops / ms_blend / 1e9 / MFMA_F32_PEAK, 4)},
           'joint_chain_ms': round(ms_chain, 3)}
    del verts, ws_v
    torch.cuda.empty_cache()
    # roll-out at B sequences x (T-1) steps
    hm = HumorModel(in_rot_rep='mat', out_rot_rep='aa', model_data_config='smpl+joints+contacts')
    hm.load_state_dict(synth.humor_state_dict(seed=0))
    hm = hm.to(dev).eval()
    past = torch.randn(B, 339, generator=g).to(dev).requires_grad_(True)
    z = torch.randn(B, T - 1, 48, generator=g).to(dev).requires_grad_(True)

    def fwd():
        with torch.no_grad():
            hm.roll_out(past, None, T - 1, z_seq=z, return_prior=True)

    def fwdbwd():
        out, (pm, pv) = hm.roll_out(past, None, T - 1, z_seq=z, return_prior=True)
        (out['trans'].sum() + pm.sum()).backward()
    ms_f, ms_fb = ev(fwd, iters=3, warm=1), ev(fwdbwd, iters=3, warm=1)

    def fwd_chain_only():
        with torch.no_grad():
            hm.roll_out(past, None, T - 1, z_seq=z, return_prior=False)
    ms_chain = ev(fwd_chain_only, iters=3, warm=1)
    mlp_flops = 11.55e6 * B * (T - 1)                  # 11.55 MFLOP per row-step (prior + decoder)
    dec_flops = 2 * 2171736.0 * B * (T - 1)            # the decoder chain's share (the recurrence: what the pipelined kernels run)
    res['rollout'] = {'path': 'pipelined persistent kernels (rollout_pipe.inc), one launch per direction' if 32 < B <= 256 else
                              ('persistent kernels (rollout_persist.hip)' if B <= 32 else 'pipelined persistent kernels, chunks of 256'),
                      'fwd_ms': round(ms_f, 2), 'fwd_bwd_ms': round(ms_fb, 2), 'steps_per_sec_fwd': round(B * (T - 1) / (ms_f * 1e-3), 1),
                      'mlp_TFLOPs_fwd': round(mlp_flops / ms_f / 1e9, 2), 'frac_fp32_mfma_fwd': round(mlp_flops / ms_f / 1e9 / MFMA_F32_PEAK, 4),
                      'decoder_chain_fwd_ms': round(ms_chain, 2), 'decoder_chain_frac_fp32_mfma': round(dec_flops / ms_chain / 1e9 / MFMA_F32_PEAK, 4)}
    # the launch chain on the same inputs (what rounds 1-4 ran at this size), for the record
    lib.call('ha_tune_set', b'rollout_persist', 0)
    try:
        ms_f0, ms_fb0 = ev(fwd, iters=2, warm=1), ev(fwdbwd, iters=2, warm=1)
        res['rollout']['launch_chain'] = {'fwd_ms': round(ms_f0, 2), 'fwd_bwd_ms': round(ms_fb0, 2), 'frac_fp32_mfma_fwd': round(mlp_flops / ms_f0 / 1e9 / MFMA_F32_PEAK, 4)}
    finally:
        lib.call('ha_tune_set', b'rollout_persist', 1)
    return res


def main():
    B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 120)
    print(json.dumps(measure(B, T)))


if __name__ == '__main__':
    main()
