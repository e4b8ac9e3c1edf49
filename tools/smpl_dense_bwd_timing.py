#!/usr/bin/env python
"""Dense-gradient SMPL backward (every vertex carries a gradient: point-cloud / chamfer term): the batched streaming + MFMA path
(ha_smpl_backward_dense) against the wave-per-frame adjoint (ha_smpl_backward, slot 0), same inputs; prints ms per call and the
largest difference between the two gradients."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from humor_amd import _lib, synth                        # noqa: E402
from humor_amd.body_model import _get_handle             # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
    n_active = int(sys.argv[2]) if len(sys.argv) > 2 else 22
    dev = torch.device('cuda:0')
    lib = _lib.get_lib()
    npz = synth.write_smplh_npz('/tmp/model_dbt.npz', seed=0)
    h = _get_handle(lib, npz, 16, 0)
    g = torch.Generator().manual_seed(0)
    r = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(dev)
    pose = torch.zeros(N, h.J * 3, device=dev)
    pose[:, :n_active * 3] = r(N, n_active * 3, sc=0.4)
    betas, transl = r(N, 16), r(N, 3)
    gV, gJ = r(N, h.V, 3), r(N, h.J, 3)
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    verts, joints, A = new(N, h.V, 3), new(N, h.J, 3), new(N, h.J, 12)
    nv, nc, nw = C.c_int64(), C.c_int64(), C.c_int64()
    lib.call('ha_smpl_workspace', h.ptr, N, n_active, C.byref(nv), C.byref(nc))
    ws_v, ws_c = new(nv.value), new(nc.value)
    st = _lib.stream_ptr(pose)
    p = _lib.ptr
    lib.call('ha_smpl_forward', h.ptr, 0, N, n_active, p(pose), p(betas), p(transl), p(verts), p(joints), p(A), p(ws_v), p(ws_c), 2, st)
    lib.call('ha_smpl_backward_dense_workspace', h.ptr, N, n_active, C.byref(nw))
    ws = new(nw.value)
    out = {k: (new(N, h.J * 3), new(N, 16), new(N, 3)) for k in ('frame', 'dense')}

    def frame():
        lib.call('ha_smpl_backward', h.ptr, 0, N, n_active, p(pose), p(betas), p(gV), p(gJ), *[p(t) for t in out['frame']], st)

    def dense():
        lib.call('ha_smpl_backward_dense', h.ptr, N, n_active, p(pose), p(betas), p(gV), p(gJ), p(ws_v), p(A), p(ws), *[p(t) for t in out['dense']], st)

    def timeit(fn, n):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    t_dense = timeit(dense, 20)
    t_frame = timeit(frame, 3)
    print(f'N={N} n_active={n_active} workspace {nw.value * 4 / 2**20:.0f} MiB: wave-per-frame adjoint {t_frame:.3f} ms, batched dense backward {t_dense:.3f} ms '
          f'({t_frame / t_dense:.1f}x)')
    for name, a, b in zip(('g_pose', 'g_betas', 'g_transl'), out['frame'], out['dense']):
        print(f'  {name}: max |frame - dense| = {(a - b).abs().max().item():.3e}  (scale {a.abs().max().item():.3e})')
    lib.call('ha_tune_set', b'dense_gA_sparse', 0)
    t_mfma = timeit(dense, 20)
    lib.call('ha_tune_set', b'dense_gA_sparse', 1)
    t_list = timeit(dense, 20)
    g_list = [t.clone() for t in out['dense']]
    lib.call('ha_tune_set', b'dense_gA_sparse', 2)
    dense()
    torch.cuda.synchronize()
    d_comp = max((a - b).abs().max().item() for a, b in zip(g_list, out['dense']))
    print(f'  dL/dA variants (ha_tune_set dense_gA_sparse): chunk-compressed MFMA product (2, default) {t_dense:.3f} ms; joint lists (1) {t_list:.3f} ms; '
          f'dense 64-column MFMA product (0) {t_mfma:.3f} ms (max |difference| default vs joint lists {d_comp:.2e})')
    for waves in [int(w) for w in os.environ.get("WAVES", "1600").split(",")]:
        lib.call('ha_tune_set', b'dense_bwd_waves', waves)
        lib.call('ha_smpl_backward_dense_workspace', h.ptr, N, n_active, C.byref(nw))
        ws = new(nw.value)
        print(f'  dense_bwd_waves={waves}: {timeit(dense, 10):.3f} ms')
    lib.call('ha_tune_set', b'dense_bwd_waves', 0)


if __name__ == '__main__':
    main()
