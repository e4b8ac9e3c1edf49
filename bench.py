#!/usr/bin/env python
"""bench.py -- headline benchmark of the HuMoR test-time-optimisation hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` (N>1 launched through torch.distributed.run,
one rank per GPU) prints ONE JSON line from rank 0.

Workload (BASELINE.json `metric`: batch=32 seq=60): config C4 -- 32 sub-sequences x 60 frames (N = 1920 SMPL+H
frames) per GPU (weak scaling: every rank owns its own 32 sub-sequences; the reference shards sub-sequence
batches across GPUs and only the overlap-consistency gradient crosses ranks).
One "step" = one evaluation of the fitting hot path over the batch, forward AND backward, with inputs resident
in HBM: see `HotPath.step`.  `value` = steps (closure evaluations) per second over all ranks.
Secondary figures in the same line: `smpl_verts_per_sec` (dense 6890-vertex SMPL forward), the `roofline` of the
dominant kernel measured with HIP events on the launch stream, and `cpu_baseline` (the oracle restatement of the
reference path timed on the host cores of the same machine, on a bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_SEQ, T_SEQ = 32, 60          # BASELINE.json metric: batch=32 seq=60
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
V, J = 6890, 52
# algorithmic HBM bytes per frame of the LBS skinning kernel = read v_posed (V*12) + A (J*48),
# write verts (V*12); the joint/keypoint gathers are not part of this kernel in our pipeline.
SKIN_BYTES_PER_FRAME = V * 12 * 2 + J * 48


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    return ap.parse_args()


class HotPath:
    """The per-closure hot path at C4 size on one GPU."""

    def __init__(self, dev, npz, seed):
        from humor_amd import synth
        from humor_amd.body_model import BodyModel, SMPLH_SELECTOR_VERTS
        self.dev = dev
        N = B_SEQ * T_SEQ
        self.N = N
        root, body, trans = synth.smooth_pose_sequence(B_SEQ, T_SEQ, seed=seed)
        g = torch.Generator().manual_seed(seed)
        self.root = root.reshape(N, 3).to(dev).requires_grad_(True)
        self.body = body.reshape(N, 63).to(dev).requires_grad_(True)
        self.trans = trans.reshape(N, 3).to(dev).requires_grad_(True)
        self.betas_seq = torch.randn(B_SEQ, 16, generator=g).to(dev).requires_grad_(True)
        from humor_amd.tables import KEYPT_VERTS
        self.bm_fit = BodyModel(npz, num_betas=16, batch_size=N, use_vtx_selector=True, vertex_subset=KEYPT_VERTS)
        self.bm_dense = BodyModel(npz, num_betas=16, batch_size=N, use_vtx_selector=True)
        self.obs_j = torch.randn(N, 73, 3, generator=g).to(dev)
        self.obs_v = torch.randn(N, len(KEYPT_VERTS), 3, generator=g).to(dev)
        # motion prior roll-out: 32 sequences x 59 steps, latent sequence + initial state are the optimisation variables
        from humor_amd.humor_model import HumorModel
        self.hm = HumorModel(in_rot_rep='mat', out_rot_rep='aa', latent_size=48, model_data_config='smpl+joints+contacts',
                             steps_in=1)
        self.hm.load_state_dict(synth.humor_state_dict(seed=0))
        self.hm = self.hm.to(dev).eval()
        for p_ in self.hm.parameters():
            p_.requires_grad_(False)
        from oracle.make_golden import canonical_state   # input generator only (oracle is never the timed path)
        self.past0 = canonical_state(B_SEQ, g).to(dev).requires_grad_(True)
        self.z = (0.5 * torch.randn(B_SEQ, T_SEQ - 1, 48, generator=g)).to(dev).requires_grad_(True)
        self.obs_w = torch.randn(B_SEQ, T_SEQ - 1, 348, generator=g).to(dev)

    def step(self):
        """One hot-path evaluation: SMPL (the 64 vertices + 73 joints the losses consume) forward, a joints/keypoint
        data term, backward to pose/shape/translation."""
        for t in (self.root, self.body, self.trans, self.betas_seq, self.past0, self.z):
            t.grad = None
        betas = self.betas_seq.unsqueeze(1).expand(B_SEQ, T_SEQ, 16).reshape(self.N, 16)
        out = self.bm_fit(root_orient=self.root, pose_body=self.body, betas=betas, trans=self.trans)
        loss = (out.Jtr - self.obs_j).square().sum() + (out.v - self.obs_v).square().sum()
        pred, (pm, pv) = self.hm.roll_out(self.past0, None, T_SEQ - 1, z_seq=self.z, return_prior=True)
        world = torch.cat([pred[k] for k in ('trans', 'trans_vel', 'root_orient', 'root_orient_vel', 'pose_body', 'joints',
                                              'joints_vel', 'contacts')], dim=2)
        loss = loss + (world - self.obs_w).square().sum() + ((self.z - pm).square() / pv).sum()
        loss.backward()
        return loss

    def dense_forward(self):
        with torch.no_grad():
            betas = self.betas_seq.unsqueeze(1).expand(B_SEQ, T_SEQ, 16).reshape(self.N, 16)
            return self.bm_dense(root_orient=self.root, pose_body=self.body, betas=betas, trans=self.trans)


def time_events(fn, iters, warm=2):
    """Average duration (ms) of fn() measured with HIP events on the current (launch) stream."""
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


def skin_roofline(dev, npz):
    """The streaming LBS kernel alone (ha_lbs_skin) at N = 1920: algorithmic bytes / event-timed launch duration.
    Launched on torch's current stream, so torch.cuda.Event brackets exactly these launches."""
    from humor_amd import _lib
    from humor_amd.body_model import BodyModel
    lib = _lib.get_lib()
    N = B_SEQ * T_SEQ
    bm = BodyModel(npz, num_betas=16)
    h = bm._handle_for(dev)
    vposed = torch.randn(N * V * 3 + 4, device=dev)
    A = torch.randn(N, J, 12, device=dev)
    transl = torch.randn(N, 3, device=dev)
    verts = torch.empty(N, V, 3, device=dev)
    st = _lib.stream_ptr(verts)

    def launch():
        lib.call('ha_lbs_skin', h.ptr, N, _lib.ptr(vposed), _lib.ptr(A), _lib.ptr(transl), _lib.ptr(verts), st)
    ms = time_events(launch, iters=50, warm=5)
    bytes_per_launch = SKIN_BYTES_PER_FRAME * N
    gbs = bytes_per_launch / (ms * 1e-3) / 1e9
    return {'kernel': 'lbs_skin_kernel', 'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(gbs / HBM_PEAK_GBS, 4), 'traffic': None, 'avg_launch_us': round(ms * 1e3, 2),
            'bytes_per_launch': bytes_per_launch, 'frames_per_launch': N}


def cpu_baseline(npz, seed):
    """The oracle restatement of the same step on the host cores, bounded sample: 4 sequences x 60 frames
    (1/8 of the C4 batch), dense 6890-vertex smplx-style LBS forward + backward as the reference runs it."""
    from oracle import lbs_restated as L
    from humor_amd import synth
    from humor_amd.tables import KEYPT_VERTS
    ncores = os.cpu_count()
    torch.set_num_threads(ncores)
    data = np.load(npz)

    class DS:
        pass
    ds = DS()
    for k in data.files:
        setattr(ds, k, data[k])
    Bs = 4
    N = Bs * T_SEQ
    root, body, trans = synth.smooth_pose_sequence(Bs, T_SEQ, seed=seed)
    g = torch.Generator().manual_seed(seed)
    root = root.reshape(N, 3).requires_grad_(True)
    body = body.reshape(N, 63).requires_grad_(True)
    trans = trans.reshape(N, 3).requires_grad_(True)
    betas = torch.randn(Bs, 16, generator=g).requires_grad_(True)
    layer = L.SMPLHLayer(data_struct=ds, num_betas=16, batch_size=N, vertex_ids=L.VERTEX_IDS_SMPLH)
    obs_j = torch.randn(N, 73, 3, generator=g)
    obs_v = torch.randn(N, len(KEYPT_VERTS), 3, generator=g)

    def step():
        b = betas.unsqueeze(1).expand(Bs, T_SEQ, 16).reshape(N, 16)
        out = layer(betas=b, global_orient=root, body_pose=body, transl=trans)
        loss = (out.joints - obs_j).square().sum() + (out.vertices[:, KEYPT_VERTS] - obs_v).square().sum()
        loss.backward()
    step()
    reps, t0 = 0, time.time()
    while reps < 3 or time.time() - t0 < 8.0:
        step()
        reps += 1
    dt = (time.time() - t0) / reps
    # scale the sample (4 sequences) to the C4 batch (32 sequences): the CPU path is linear in frames
    return {'value': round(1.0 / (dt * (B_SEQ / Bs)), 4), 'unit': 'closure-evals/s', 'cores': ncores, 'kind': 'port',
            'sample': f'{Bs}x{T_SEQ} frames (1/8 of the 32x60 batch), {reps} reps, {dt * 1e3:.1f} ms each; '
                      f'oracle/lbs_restated.py (smplx-0.1.28 op sequence, dense 6890-vertex LBS fwd+bwd) scaled x{B_SEQ // Bs}'}


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(f'cuda:{local}'))
    dev = torch.device(f'cuda:{local}')
    torch.cuda.set_device(dev)

    from humor_amd import synth
    tmp = tempfile.mkdtemp(prefix='humor_amd_bench_')
    npz = synth.write_smplh_npz(os.path.join(tmp, f'model_{rank}.npz'), seed=0)
    hp = HotPath(dev, npz, seed=100 + rank)

    for _ in range(args.warmup):
        hp.step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hp.step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    ms_dense = time_events(hp.dense_forward, iters=10, warm=2)
    verts_per_sec = hp.N * V / (ms_dense * 1e-3) * world

    if rank == 0:
        res = {
            'metric': 'fitting closure evaluations/s (hot-path fwd+bwd), batch=32 seq=60 per GPU',
            'value': round(args.steps * world / dt, 3), 'unit': 'closure-evals/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'C4 fit batch: 32 sub-sequences x 60 frames per GPU (N=1920 SMPL+H frames, 6890 verts, '
                                   '52 joints, 16 betas), SMPL 64-vertex fitting subset fwd+bwd per step',
                       'global_batch': B_SEQ * world, 'seq_len': T_SEQ, 'parallelism': f'dp{world}'},
            'smpl_verts_per_sec': round(verts_per_sec, 1),
            'smpl_dense_fwd_ms': round(ms_dense, 4),
            'roofline': skin_roofline(dev, npz),
        }
        if not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(npz, seed=100)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
