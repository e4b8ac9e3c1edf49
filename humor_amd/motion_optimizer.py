"""``MotionOptimizer`` with the reference's constructor and ``run`` signatures (humor/fitting/motion_optimizer.py:29-676),
driving the MI355X kernels: the 3-stage L-BFGS schedule, optimisation variables, closures and result dictionaries are the
reference's; what each closure evaluates goes through libhumor_amd.so:

  * SMPL: only the rows that exist and only the 43 + 21 vertices the losses read -- no expansion of single frames to the
    full B*T batch, no zero padding, no dense 6890-vertex output (values are identical because the
    reference slices those rows/vertices back out, motion_optimizer.py:1075-1100);
  * roll-out: one forward and one backward C-ABI call (HumorModel.roll_out), R -> axis-angle on the HIP kernel;
  * per-term logging (a host sync per loss term per closure in the reference, fitting_utils.py:261-272) is off unless
    ``verbose=True``.

Multi-GPU: pass ``shard=humor_amd.distributed.Shard(B, group)`` -- replicated L-BFGS, closure evaluated on the local slice
of sub-sequences, one packed all-reduce per closure (humor_amd/distributed.py).
"""
import os
import math
import time

import numpy as np
import torch

from . import frames, ops
from .body_model import BodyModel
from .fitting_loss import FittingLoss
from .tables import (CONTACT_INDS, KEYPT_VERTS, OP_EDGE_LIST, OP_IGNORE_JOINTS, SMPL_JOINTS, SMPLH_TO_OPENPOSE25)

LINE_SEARCH = 'strong_wolfe'
J_BODY = len(SMPL_JOINTS) - 1
CONTACT_THRESH = 0.5


class MotionOptimizer():
    ''' Fits SMPL shape and motion to observation sequence '''

    def __init__(self, device, body_model, num_betas, batch_size, seq_len, observed_modalities, loss_weights, pose_prior,
                 motion_prior=None, init_motion_prior=None, optim_floor=False, camera_matrix=None, robust_loss_type='none',
                 robust_tuning_const=4.6851, joint2d_sigma=100, stage3_tune_init_state=True, stage3_tune_init_num_frames=15,
                 stage3_tune_init_freeze_start=30, stage3_tune_init_freeze_end=50, stage3_contact_refine_only=False,
                 use_chamfer=False, im_dim=(1080, 1080), shard=None, verbose=False, use_graphs='auto', fused_loss=True, lbfgs='fused', fused_post=True, fused_pre=True, fused_vposer=True,
                 rigid_cam_body=True, fused_stage3=True):
        B, T = batch_size, seq_len
        self.device = device
        self.batch_size, self.seq_len = B, T
        self.num_betas = num_betas
        self.optim_floor = optim_floor
        self.stage3_tune_init_state = stage3_tune_init_state
        self.stage3_tune_init_num_frames = stage3_tune_init_num_frames
        self.stage3_tune_init_freeze_start = stage3_tune_init_freeze_start
        self.stage3_tune_init_freeze_end = stage3_tune_init_freeze_end
        self.stage3_contact_refine_only = stage3_contact_refine_only
        self.im_dim = im_dim
        self.shard = shard
        self.verbose = verbose
        # whole-closure hipGraph capture (replay costs ~1.5 us of node hand-off per kernel, eager costs host time that the
        # roll-out's long launches partly hide: which one wins depends on the host -- bench.py times both)
        # use_graphs: False | True (every closure) | 'auto' (the short, host-bound closures only: stages 1-2 and the 15-frame
        # tune-init phase of stage 3 replay 2-3x faster than they launch; the full-length stage-3 closure is GPU-bound and its
        # ~1100 graph nodes cost more in node hand-off than eager launches cost in host time)
        self.use_graphs = (use_graphs if use_graphs == 'auto' else bool(use_graphs)) if torch.device(device).type == 'cuda' else False
        if self.use_graphs and shard is not None:
            # capturing the closure would put the halo all-gather / gradient all-reduce (and, with gloo, host staging) inside the
            # graph; a rank-divergent capture failure leaves collectives in flight.  Sharded closures always launch eagerly.
            if self.use_graphs is True:
                print('humor_amd: use_graphs is ignored for a sharded MotionOptimizer (collectives are not captured)')
            self.use_graphs = False
        # 'fused': humor_amd.lbfgs.LBFGS (same algorithm as torch.optim.LBFGS, a handful of launches per inner iteration);
        # 'torch': torch.optim.LBFGS itself (what the reference uses; ~4 launches per stored pair and iteration)
        if lbfgs not in ('fused', 'torch'):
            raise ValueError("lbfgs must be 'fused' or 'torch'")
        self.lbfgs_impl = lbfgs
        self.fused_post = bool(fused_post)
        self.fused_pre = bool(fused_pre)
        # stage 3 evaluates the body model on the same pose and shape under two root trajectories (prior frame, camera frame):
        # the second evaluation as the rigid image of the first (ha_rigid_image_*) instead of a second SMPL forward + backward
        self.rigid_cam_body = bool(rigid_cam_body)
        # stage 3 as three composite autograd nodes (humor_amd/stage3.py): no accumulation / cat / expand launches between the kernels
        self.fused_stage3 = bool(fused_stage3)
        # sharded closures: gradients of the variables land in one persistent packed arena that is all-reduced in place (distributed.GradArena)
        # (p.grad is then a VIEW of the arena, which the next closure evaluation zero-fills: an optimiser that keeps p.grad across
        # evaluations must clone it -- humor_amd.lbfgs.LBFGS and torch.optim.LBFGS both copy the flat gradient -- or set this to False)
        self.grad_arena = True
        # VPoser decode (+ 6-D -> R -> axis-angle) / encode through ha_mlp_* instead of the module's ATen ops (humor_amd/mlp.py)
        self.fused_vposer = bool(fused_vposer)
        self._vposer_handle = None
        self.closure_evals = 0
        self._graph_states = []     # capture states of the closures made so far (invalidated when the persistent roll-out reports a failure)
        self.stage_profile = None   # set to {} before run(): wall time / closure evaluations / outer iterations per stage (3 syncs per phase)
        self.loss_trace = None      # set to a list to record (stage, loss) of every closure evaluation (host sync per eval)
        self.iter_log = None        # set to a list to record (phase, wall clock, closure evaluations so far) after every stage-3 outer iteration
        if motion_prior is None:
            raise ValueError('Need the motion prior to use all-implicit parameterization!')
        # the body model evaluates only what the losses consume (43 key vertices + 21 selector vertices) unless a point-cloud term
        # needs the whole surface; it shares the packed constants with the model we were given
        self.dense_smpl = bool(use_chamfer) or 'points3d' in observed_modalities
        self.body_model = body_model
        if isinstance(body_model, BodyModel):
            self.fit_bm = BodyModel(body_model.bm_path, num_betas=num_betas, batch_size=B * T,
                                    use_vtx_selector=body_model.use_vtx_selector, model_type=body_model.model_type,
                                    vertex_subset=None if self.dense_smpl else KEYPT_VERTS, _lib_override=body_model._lib)
        else:
            raise TypeError('humor_amd.MotionOptimizer needs a humor_amd.BodyModel')

        # optimisation variables (motion_optimizer.py:70-84)
        self.pose_prior = pose_prior
        self.latent_pose_dim = pose_prior.latentD
        # stage-1/2 variables back to back in one allocation, in the order the stage-2 optimiser lists them (stage 1 = a prefix):
        # the fused L-BFGS then works on the flat buffer without copying (humor_amd/lbfgs.py)
        from .lbfgs import flat_arena
        _, (self.trans, self.root_orient, self.betas, self.latent_pose) = flat_arena(
            [(B, T, 3), (B, T, 3), (B, num_betas), (B, T, self.latent_pose_dim)], device)
        self.root_orient[:, :, 0] = np.pi
        self.motion_prior = motion_prior
        self.init_motion_prior = init_motion_prior
        self.latent_motion = None
        self.latent_motion_dim = motion_prior.latent_size
        self.cond_prior = motion_prior.use_conditional_prior
        self.trans_vel = self.root_orient_vel = self.joints_vel = None
        self.init_fidx = np.zeros((B), dtype=np.int64)
        self._contact_idx = torch.as_tensor(CONTACT_INDS, dtype=torch.long, device=device)

        self.cam_f = self.cam_center = None
        if optim_floor:
            if camera_matrix is None:
                raise ValueError('Must have camera intrinsics (camera_matrix) to optimize the floor plane!')
            self.floor_plane = torch.zeros((B, 3), device=device)
            self.floor_plane[:, 2] = 1.0
            self.cam2prior_R = torch.eye(3, device=device).reshape(1, 3, 3).expand(B, 3, 3)
            self.cam2prior_t = torch.zeros((B, 3), device=device)
            self.cam2prior_root_height = torch.zeros((B, 1), device=device)
            self.cam_f = torch.stack([camera_matrix[:, 0, 0], camera_matrix[:, 1, 1]], dim=1)
            self.cam_center = torch.stack([camera_matrix[:, 0, 2], camera_matrix[:, 1, 2]], dim=1)
        self.use_camera = self.cam_f is not None

        self.smpl2op_map = list(SMPLH_TO_OPENPOSE25)
        cam_f, cam_c = self.cam_f, self.cam_center
        if shard is not None and cam_f is not None:
            cam_f, cam_c = shard.sl(cam_f), shard.sl(cam_c)
        self.fitting_loss = FittingLoss(loss_weights, self.init_motion_prior, self.smpl2op_map, OP_IGNORE_JOINTS, cam_f, cam_c,
                                        robust_loss_type, robust_tuning_const, joints2d_sigma=joint2d_sigma,
                                        use_chamfer=self.dense_smpl, fused=fused_loss, _lib_override=self.fit_bm._lib).to(device)

    # ------------------------------------------------------------------------------------------------
    # small helpers
    # ------------------------------------------------------------------------------------------------
    def _local(self, x):
        """This rank's slice of a per-sequence tensor (identity without sharding).  Inside a sharded closure the optimisation variables
        come through the gradient arena (distributed.GradArena): same rows, no slice_backward fill + copy per variable."""
        if self.shard is None:
            return x
        arena = getattr(self, '_arena', None)
        if arena is not None and x.requires_grad and x.is_leaf:
            r = arena.rows(x)
            if r is not None:
                return r
        return self.shard.sl(x)

    def _local_obs(self, observed_data, nsteps=None):
        out = {}
        for k, v in observed_data.items():
            if k == 'prev_batch_overlap_res':
                out[k] = v
            elif k == 'seq_interval':
                v = v.cpu()      # frame intervals are host-side integers (no device sync inside the closure)
                if self.shard is None or self.shard.rank == 0:
                    out[k] = self._local(v)
                else:
                    out[k] = v[self.shard.b0 - 1:self.shard.b1]
                if self.shard is not None:
                    iv = v.tolist()
                    ovs = [int(iv[b - 1][1]) - int(iv[b][0]) for b in range(1, len(iv))]
                    b1 = self.shard.b1
                    # frames exchanged per side (identical on every rank) and this rank's overlap with the NEXT rank's first sequence
                    self._pair_info = {'ovm': max(1, min(self.seq_len, max(ovs))) if ovs else 1,
                                       'ov_next': ovs[b1 - 1] if b1 < len(iv) else None}
            else:
                v = self._local(v)
                out[k] = v[:, :nsteps] if nsteps is not None else v
        return out

    def _halo(self, pred_verts3d, betas, floor_plane, active):
        """Forward-only exchange (SURVEY 8(e) option B) of the sequences at the rank boundaries: ONE all_gather of each rank's
        [tail of its last sequence | head of its first sequence] (predicted key vertices of the frames that can overlap, betas, floor).
        A rank evaluates the consistency pair (b-1, b) that straddles its lower boundary from the gathered copy of b-1 (counted in
        the loss) and the pair at its upper boundary from the copy of the next rank's first sequence (FittingLoss.
        overlap_next_side: gradient only, zero value), so every pair is counted once, every variable gets its exact gradient
        from its own rank, and no collective runs in the backward pass."""
        if self.shard is None:
            return None
        sh = self.shard
        if not active:
            return {'first': sh.rank == 0, 'prev_tail': None, 'prev_betas': None, 'prev_floor': None}
        from .distributed import all_gather_flat
        T, nb = pred_verts3d.size(1), self.num_betas
        ovm = min(self._pair_info['ovm'], T)
        zf = pred_verts3d.new_zeros(3) if floor_plane is None else None      # (placeholder of the floor slot: one fill launch, only when needed)
        with torch.no_grad():
            packed = torch.cat([pred_verts3d[-1, T - ovm:].reshape(-1), betas[-1].reshape(-1), floor_plane[-1].reshape(-1) if floor_plane is not None else zf,
                                pred_verts3d[0, :ovm].reshape(-1), betas[0].reshape(-1), floor_plane[0].reshape(-1) if floor_plane is not None else zf])
            allg = all_gather_flat(packed, sh.group)
        nv = ovm * len(KEYPT_VERTS) * 3
        half = nv + nb + 3
        halo = {'first': sh.rank == 0, 'prev_tail': None, 'prev_betas': None, 'prev_floor': None}
        if sh.rank > 0:
            prev = allg[sh.rank - 1, :half]
            tail = prev[:nv].reshape(ovm, len(KEYPT_VERTS), 3)
            # the loss reads the predecessor as a whole [T,43,3] sequence and takes its last `overlap` frames
            halo['prev_tail'] = tail if ovm == T else torch.cat([tail.new_zeros(T - ovm, len(KEYPT_VERTS), 3), tail], dim=0)
            halo['prev_betas'] = prev[nv:nv + nb]
            if floor_plane is not None:
                halo['prev_floor'] = prev[nv + nb:half]
        if sh.rank < sh.world - 1 and self._pair_info['ov_next'] is not None:
            nxt = allg[sh.rank + 1, half:]
            halo['next_head'] = nxt[:nv].reshape(ovm, len(KEYPT_VERTS), 3)
            halo['next_betas'] = nxt[nv:nv + nb]
            halo['next_floor'] = nxt[nv + nb:half] if floor_plane is not None else None
            halo['ov_next'] = min(self._pair_info['ov_next'], T)
        return halo

    def make_closure(self, objective, params, optim=None, short=True):
        """Returns the L-BFGS closure for `objective()` -> (loss, stats).  With use_graphs the objective and its backward are
        captured once into a hipGraph (torch.cuda.graphs) and replayed per evaluation; `short` marks the closures that
        use_graphs='auto' captures (see __init__)."""
        if not self.use_graphs or (self.use_graphs == 'auto' and not short):
            def closure():
                for p in params:          # = optim.zero_grad(set_to_none=True) of the reference closures
                    p.grad = None
                if self.shard is not None and self.grad_arena:
                    from .distributed import GradArena
                    if getattr(self, '_arena', None) is None or not self._arena.matches(params):
                        self._arena = GradArena(params, self.shard)
                    self._arena.begin()
                try:
                    loss, stats = objective()
                    return self._finish_closure(loss, params, stats)
                finally:
                    if self.shard is not None and getattr(self, '_arena', None) is not None:
                        self._arena._rows = {}
            closure.discard_last = self._discard_last_eval
            return closure
        state = {'graph': None, 'loss': None, 'grads': None, 'failed': False}
        self._graph_states.append(state)

        def eager():
            for p in params:
                p.grad = None
            loss, stats = objective()
            return self._finish_closure(loss, params, stats)

        def closure():
            if state['failed']:
                return eager()
            if state['graph'] is None:
                try:
                    # warm-up evaluations and the capture itself are not closure evaluations of the optimiser: keep them out of
                    # the evaluation counter and the loss trace
                    evals, trace = self.closure_evals, self.loss_trace
                    self.loss_trace = None
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        for _ in range(2):
                            eager()
                    torch.cuda.current_stream().wait_stream(side)
                    for p in params:
                        p.grad = None
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        loss, _ = objective()
                        loss = self._finish_closure(loss, params, None)
                        state['loss'] = loss.detach()
                    self.closure_evals, self.loss_trace = evals, trace
                    state['grads'] = [p.grad for p in params]
                    state['graph'] = g
                except Exception as e:          # capture is an optimisation: fall back to eager evaluation
                    self.closure_evals, self.loss_trace = evals, trace
                    print('humor_amd: hipGraph capture of the closure failed (%s: %s); running eagerly' % (type(e).__name__, str(e)[:600]))
                    state['failed'] = True
                    self.graph_failures = getattr(self, 'graph_failures', 0) + 1
                    torch.cuda.synchronize()
                    return eager()
            for p, gbuf in zip(params, state['grads']):
                p.grad = gbuf
            state['graph'].replay()
            self.closure_evals += 1
            if self.loss_trace is not None and self.shard is None:
                self.loss_trace.append((self.fitting_loss.cur_stage_idx, float(state['loss'])))
            return state['loss']
        closure.discard_last = self._discard_last_eval
        return closure

    def _rollout_error_word(self):
        """The persistent roll-out's host-mapped error word on this fit's device (0: no failure reported so far; also 0 without a GPU)."""
        if self.motion_prior is None or self.device is None or torch.device(self.device).type != 'cuda':
            return 0
        status = getattr(self.motion_prior, 'persistent_rollout_status', None)
        if status is None:
            return 0
        return status(torch.device(self.device))[1]

    def _check_rollout_health(self):
        """Called after every stage-3 outer iteration (L-BFGS has just read its scalars: the stream is drained).  The persistent roll-out
        kernels report a team that did not complete -- another process's kernel held part of the chip, say -- only through NaN results
        and a host-mapped error word; closures replayed from a hipGraph never pass an entry point that returns it.  On a failure reported
        DURING this fit: the captured graphs (they contain the persistent launches) are dropped, the network serves later calls through
        the launch chain, and this fit is aborted with an error (run_fitting.py:437-439 skips the batch).  The word is sticky for the
        lifetime of the network handle, which outlives a fit (run_fitting.py makes one MotionOptimizer per batch around one HumorModel):
        a failure an EARLIER fit already reported must not abort this one -- its evaluations ran on the launch chain and are valid --
        so run() records the word it starts with and only a change counts."""
        err = self._rollout_error_word()
        if err != getattr(self, '_rollout_err0', 0):
            self._rollout_err0 = err
            for st in self._graph_states:
                st['graph'], st['failed'] = None, True
            # this fit is the one the failure aborts: the next fit around the same network must not be handed it again by its first entry point
            ack = getattr(self.motion_prior, 'acknowledge_persistent_failure', None)
            if ack is not None:
                ack(torch.device(self.device))
            raise RuntimeError('humor_amd: the persistent roll-out reported an incomplete launch (error word 0x%x): the objective values of '
                               'this fit are invalid; later evaluations use the launch-chain roll-out' % err)

    def _check_finite(self, optim, stage, it):
        """Called after every outer iteration with the objective value and the largest gradient entry L-BFGS has just read on the host (no
        extra synchronisation).  A non-finite one means the iterates are (or are about to become) NaN: the fit is aborted with an error, so
        that the caller skips the batch (run_fitting.py:437-439) instead of writing NaN results -- round 5 returned a NaN fit silently."""
        loss, gmax = getattr(optim, 'last_loss', 0.0), getattr(optim, 'last_gmax', 0.0)
        if not (math.isfinite(loss) and math.isfinite(gmax)):
            for st in self._graph_states:
                st['graph'], st['failed'] = None, True
            raise RuntimeError('humor_amd: non-finite objective in %s, outer iteration %d (loss %r, largest gradient entry %r): the fit is '
                               'aborted, no results are returned for this batch' % (stage, it, loss, gmax))

    def _discard_last_eval(self):
        """humor_amd.lbfgs.LBFGS issues the first trial evaluation of an iteration before it has read the direction's scalars; in the
        rare cases where torch would have stopped before that evaluation the optimiser discards it and says so here, so that the
        evaluation counter and the loss trace stay those of the reference's torch.optim.LBFGS run."""
        self.closure_evals -= 1
        if self.loss_trace:
            self.loss_trace.pop()

    def _mark(self, key, iters=0):
        """Stage timing for bench.py: closes the interval `key` (seconds since the previous mark, closure evaluations, outer iterations)."""
        if self.stage_profile is None:
            return
        if self.device is not None and torch.device(self.device).type == 'cuda':
            torch.cuda.synchronize()
        now = time.perf_counter()
        prev = self.stage_profile.get('_last', (now, self.closure_evals))
        if key is not None:
            self.stage_profile[key] = {'seconds': now - prev[0], 'closure_evals': self.closure_evals - prev[1], 'outer_iters': iters}
        self.stage_profile['_last'] = (now, self.closure_evals)

    def _finish_closure(self, loss, params, stats=None):
        if loss.dim() == 0 and loss.dtype == torch.float32:
            from .fit_kernels import unit_seed
            loss.backward(gradient=unit_seed(loss))      # (a cached 1.0: no ones_like launch, see fit_kernels.unit_seed)
        else:
            loss.backward()
        self.closure_evals += 1
        if self.shard is not None:
            arena = getattr(self, '_arena', None)
            if self.grad_arena and arena is not None and arena.matches(params):
                loss = arena.allreduce(loss, self.shard.group)
            else:
                from .distributed import allreduce_loss_and_grads
                loss = allreduce_loss_and_grads(loss, params, self.shard.group)
        if self.loss_trace is not None:
            self.loss_trace.append((self.fitting_loss.cur_stage_idx, float(loss.detach())))
        if self.verbose and stats is not None:
            print('LOSS: %f' % loss.item(), {k: float(v) for k, v in stats.items()})
        return loss

    def _make_lbfgs(self, params, lr, max_iter):
        if self.lbfgs_impl == 'torch':
            return torch.optim.LBFGS(params, max_iter=max_iter, lr=lr, line_search_fn=LINE_SEARCH)
        from .lbfgs import LBFGS
        return LBFGS(params, max_iter=max_iter, lr=lr, line_search_fn=LINE_SEARCH, _lib_override=self.fit_bm._lib)

    # ------------------------------------------------------------------------------------------------
    def initialize(self, observed_data):
        '''Floor from the observation; depth from the focal length and bone-length ratio (motion_optimizer.py:141-199).'''
        if not self.optim_floor:
            return
        fp = observed_data['floor_plane']
        self.floor_plane = (fp[:, :3] * fp[:, 3:]).to(torch.float).clone().detach()
        self.floor_plane.requires_grad = True
        if 'points3d' in observed_data:
            self.trans = torch.mean(observed_data['points3d'], dim=2).clone().detach()      # mean of the point cloud (motion_optimizer.py:152-156)
        elif 'joints2d' in observed_data:
            body_pose = self.latent2pose(self.latent_pose[:, :1])
            pred, _ = self.smpl_results(self.trans[:, :1], self.root_orient[:, :1], body_pose, self.betas)
            full = torch.cat([pred['joints3d'], pred['joints3d_extra']], dim=2)
            j3d_op = full[:, 0][:, self.smpl2op_map]                       # [B,25,3] (pose is constant over time here)
            j2d = observed_data['joints2d'][:, :, :, :2]
            conf = observed_data['joints2d'][:, :, :, 2]
            best = torch.max(torch.sum(conf > 0.0, dim=2), dim=1)[1]
            e0 = [p[0] for p in OP_EDGE_LIST]
            e1 = [p[1] for p in OP_EDGE_LIST]
            bone3d = torch.norm(j3d_op[:, e0] - j3d_op[:, e1], dim=2)     # [B,E]
            ar = torch.arange(self.batch_size, device=j2d.device)
            j2b, cb = j2d[ar, best], conf[ar, best]
            bone2d = torch.norm(j2b[:, e0] - j2b[:, e1], dim=2)
            minconf = torch.min(cb[:, e0], cb[:, e1])
            mean3d = torch.mean(bone3d, dim=1)
            mean2d = torch.mean(bone2d * (minconf > 0.0), dim=1)
            init_z = self.cam_f[:, 0] * (mean3d / mean2d)
            self.trans[:, :, 2] = init_z.unsqueeze(1).expand(self.batch_size, self.seq_len).detach()

    # ------------------------------------------------------------------------------------------------
    def run(self, observed_data, data_fps=30, lr=1.0, num_iter=[30, 70, 70], lbfgs_max_iter=20, stages_res_out=None,
            fit_gender='neutral'):
        if len(num_iter) != 3:
            raise ValueError('Must have num iters for 3 stages! But %d stages were given!' % len(num_iter))
        per_stage_outputs = {}
        T = self.seq_len
        self._rollout_err0 = self._rollout_error_word()      # (sticky per network handle: only a change during THIS fit aborts it)
        self.initialize(observed_data)
        obs_local = self._local_obs(observed_data)
        has_overlap = 'seq_interval' in observed_data

        # ---- Stage I: global root translation and orientation -----------------------------------------
        self.fitting_loss.set_stage(0)
        self.trans.requires_grad = True
        self.root_orient.requires_grad = True
        self.betas.requires_grad = False
        self.latent_pose.requires_grad = False
        params = [self.trans, self.root_orient]
        optim = self._make_lbfgs(params, lr, lbfgs_max_iter)
        closure1 = self.make_closure(lambda: self._stage1_objective(obs_local, has_overlap), params, optim)
        self._mark(None)
        for i in range(num_iter[0]):
            self.fitting_loss.cur_optim_step = i

            optim.step(closure1)
            self._check_finite(optim, 'stage 1', i)
        self._mark('stage1', num_iter[0])
        per_stage_outputs['stage1'] = self._stage_snapshot(stages_res_out, 'stage1_results.npz')

        # ---- Stage II: full pose and shape ------------------------------------------------------------
        self.fitting_loss.set_stage(1)
        self.betas.requires_grad = True
        self.latent_pose.requires_grad = True
        params = [self.trans, self.root_orient, self.betas, self.latent_pose]
        optim = self._make_lbfgs(params, lr, lbfgs_max_iter)
        closure2 = self.make_closure(lambda: self._stage2_objective(obs_local, has_overlap), params, optim)
        self._mark(None)
        for i in range(num_iter[1]):
            optim.step(closure2)
            self._check_finite(optim, 'stage 2', i)
        self._mark('stage2', num_iter[1])
        per_stage_outputs['stage2'] = self._stage_snapshot(stages_res_out, 'stage2_results.npz')
        stage2_cam = None
        if self.optim_floor and stages_res_out is not None:
            with torch.no_grad():
                stage2_cam = {'trans': self.trans.clone().detach(), 'root_orient': self.root_orient.clone().detach(),
                              'pose_body': self.latent2pose(self.latent_pose).clone().detach(), 'betas': self.betas.clone().detach()}

        # ---- Stage III set-up -------------------------------------------------------------------------
        self.fitting_loss.set_stage(2)
        og_overlap_w = self.fitting_loss.loss_weights['rgb_overlap_consist']
        motion_params, prior_opt_params = self.setup_stage3(data_fps)

        with torch.no_grad():
            rr, cam_rr = self.rollout_latent_motion(self.trans, self.root_orient, self.latent2pose(self.latent_pose), self.betas,
                                                    prior_opt_params, self.latent_motion, fit_gender=fit_gender)
            init_pred, _ = self.smpl_results(cam_rr['trans'], cam_rr['root_orient'], cam_rr['pose_body'], self.betas)
            if 'contacts' in rr:
                init_pred['contacts'] = rr['contacts']
        per_stage_outputs['stage3_init'] = init_pred
        if stages_res_out is not None:
            # camera-frame and (with a floor) prior-frame state the motion stage starts from (motion_optimizer.py:422-456)
            self._save_dicts(stages_res_out, 'stage3_init_results.npz', self.betas, cam_rr['trans'], cam_rr['root_orient'], cam_rr['pose_body'],
                             contacts=rr.get('contacts'), floor=self.floor_plane if self.optim_floor else None)
            if self.optim_floor:
                self._save_dicts(stages_res_out, 'stage3_init_results_prior.npz', self.betas, rr['trans'], rr['root_orient'], cam_rr['pose_body'],
                                 contacts=rr.get('contacts'))

        mk = lambda ps: self._make_lbfgs(ps, lr, lbfgs_max_iter)
        motion_optim = mk(motion_params)
        optim_frozen = optim_refine = None
        if self.stage3_tune_init_state:
            frozen_params = [self.latent_motion, self.betas] + ([self.floor_plane] if self.optim_floor else [])
            optim_frozen, optim_refine = mk(frozen_params), mk(motion_params)
        n_init = self.stage3_tune_init_num_frames
        saved_ch = self.fitting_loss.loss_weights['contact_height']
        saved_cv = self.fitting_loss.loss_weights['contact_vel']
        init_state_vars = [self.trans, self.root_orient, self.latent_pose, self.trans_vel, self.joints_vel, self.root_orient_vel]
        init_motion_scale = 1.0
        obs_init = self._local_obs(observed_data, nsteps=n_init)
        closures3 = {}

        self._mark(None)
        last_phase_name, phase_iters = None, 0
        for i in range(num_iter[2]):
            tune_phase = self.stage3_tune_init_state and i < self.stage3_tune_init_freeze_start
            if self.stage3_tune_init_state and self.stage3_tune_init_freeze_start <= i < self.stage3_tune_init_freeze_end:
                motion_optim = optim_frozen
                for v in init_state_vars:
                    v.requires_grad = False
                if self.stage3_contact_refine_only:
                    self.fitting_loss.loss_weights['contact_height'] = 0.0
                    self.fitting_loss.loss_weights['contact_vel'] = 0.0
                init_motion_scale = float(self.seq_len) / n_init
            elif self.stage3_tune_init_state and i >= self.stage3_tune_init_freeze_end:
                motion_optim = optim_refine
                for v in init_state_vars:
                    v.requires_grad = True
                self.betas.requires_grad = True
                if self.optim_floor:
                    self.floor_plane.requires_grad = True
                if self.stage3_contact_refine_only:
                    self.fitting_loss.loss_weights['contact_height'] = saved_ch
                    self.fitting_loss.loss_weights['contact_vel'] = saved_cv
                init_motion_scale = float(self.seq_len) / n_init
            # one closure (and one captured graph) per phase: tune-init / frozen-init / refine
            phase = (tune_phase, motion_optim is optim_frozen, init_motion_scale,
                     self.fitting_loss.loss_weights['contact_height'], self.fitting_loss.loss_weights['contact_vel'])
            phase_name = 'stage3_tune_init' if tune_phase else ('stage3_frozen_init' if motion_optim is optim_frozen else 'stage3_refine')
            if last_phase_name is not None and phase_name != last_phase_name:
                self._mark(last_phase_name, phase_iters)
                phase_iters = 0
            last_phase_name, phase_iters = phase_name, phase_iters + 1
            if phase not in closures3:
                closures3[phase] = self.make_closure(
                    lambda tp=tune_phase, ims=init_motion_scale: self._stage3_objective(
                        obs_local, obs_init, prior_opt_params, tp, n_init, ims, og_overlap_w, has_overlap, fit_gender),
                    motion_params, None, short=tune_phase)
            motion_optim.step(closures3[phase])
            self._check_rollout_health()
            self._check_finite(motion_optim, phase_name, i)
            if self.iter_log is not None:          # (tools/lbfgs_phase_profile.py: a device synchronise per outer iteration)
                torch.cuda.synchronize()
                self.iter_log.append((phase_name, time.perf_counter(), self.closure_evals))
        if last_phase_name is not None:
            self._mark(last_phase_name, phase_iters)

        # ---- final roll-out and results ----------------------------------------------------------------
        with torch.no_grad():
            body_pose = self.latent2pose(self.latent_pose)
            if self.optim_floor:
                cam_smpl, _ = self.smpl_results(self.trans, self.root_orient, body_pose, self.betas)
                self.cam2prior_R, self.cam2prior_t, self.cam2prior_root_height = frames.compute_cam2prior(
                    self.floor_plane, self.trans[:, 0], ops.batch_rodrigues(self.root_orient[:, 0], _lib_override=self.fit_bm._lib),
                    cam_smpl['joints3d'][:, 0])
            rr, cam_rr = self.rollout_latent_motion(self.trans, self.root_orient, body_pose, self.betas, prior_opt_params,
                                                    self.latent_motion, fit_gender=fit_gender)
            body_pose = rr['pose_body']
            self.latent_pose = self.pose2latent(body_pose)
            self.trans, self.root_orient = cam_rr['trans'], cam_rr['root_orient']
            stage3, _ = self.smpl_results(self.trans, self.root_orient, body_pose, self.betas)
            stage3['prior_joints3d_rollout' if self.optim_floor else 'joints3d_rollout'] = rr['joints']
            if 'contacts' in rr:
                stage3['contacts'] = rr['contacts']
            if self.optim_floor:
                stage3['prior_trans'], stage3['prior_root_orient'] = rr['trans'], rr['root_orient']
        per_stage_outputs['stage3'] = stage3
        final = self.get_optim_result(body_pose)
        if 'contacts' in rr:
            final['contacts'] = rr['contacts']
        if stages_res_out is not None:
            self._save_stage(stages_res_out, 'stage3_results.npz', body_pose, contacts=rr.get('contacts'))
            if stage2_cam is not None:
                # the stage-2 result in the prior frame of the FINAL floor (motion_optimizer.py:651-674), read by viz_fitting_rgb
                with torch.no_grad():
                    smpl2, _ = self.smpl_results(stage2_cam['trans'], stage2_cam['root_orient'], stage2_cam['pose_body'], stage2_cam['betas'])
                    ar = np.arange(self.batch_size)
                    R2, t2, h2 = frames.compute_cam2prior(
                        self.floor_plane, stage2_cam['trans'][ar, self.init_fidx],
                        ops.batch_rodrigues(stage2_cam['root_orient'][ar, self.init_fidx], _lib_override=self.fit_bm._lib),
                        smpl2['joints3d'][ar, self.init_fidx])
                    pri2 = self.apply_cam2prior(stage2_cam, R2, t2, h2, stage2_cam['pose_body'], stage2_cam['betas'], self.init_fidx)
                self._save_dicts(stages_res_out, 'stage2_results_prior.npz', self.betas, pri2['trans'], pri2['root_orient'], stage2_cam['pose_body'])
        return final, per_stage_outputs

    def setup_stage3(self, data_fps=30):
        """Stage-3 initialisation (motion_optimizer.py:324-405): cam2prior from the current floor, latent motion from the
        posterior of the current SMPL sequence, initial velocities by finite differences, SMPL variables cut to frame 0.
        Returns (motion_params, prior_opt_params)."""
        with torch.no_grad():
            cur_body_pose = self.latent2pose(self.latent_pose)
            if self.optim_floor:
                init_smpl, _ = self.smpl_results(self.trans, self.root_orient, cur_body_pose, self.betas)
                ar = np.arange(self.batch_size)
                self.cam2prior_R, self.cam2prior_t, self.cam2prior_root_height = frames.compute_cam2prior(
                    self.floor_plane, self.trans[ar, self.init_fidx],
                    ops.batch_rodrigues(self.root_orient[ar, self.init_fidx], _lib_override=self.fit_bm._lib),
                    init_smpl['joints3d'][ar, self.init_fidx])
            self.latent_motion = self.infer_latent_motion(self.trans, self.root_orient, cur_body_pose, self.betas, data_fps).detach()
            vel_trans, vel_root = self.trans, self.root_orient
            if self.optim_floor:
                pd = self.apply_cam2prior({'trans': self.trans, 'root_orient': self.root_orient}, self.cam2prior_R, self.cam2prior_t,
                                          self.cam2prior_root_height, cur_body_pose, self.betas, self.init_fidx)
                vel_trans, vel_root = pd['trans'], pd['root_orient']
            tv, jv, rv = self.estimate_velocities(vel_trans, vel_root, cur_body_pose, self.betas, data_fps)
        # stage-3 variables back to back in one allocation; the variables of the frozen-init phase (latent motion, betas, floor)
        # come first so that both optimiser variable lists are contiguous (the fused L-BFGS binds them without copying)
        from .lbfgs import flat_arena
        src = [self.latent_motion, self.betas] + ([self.floor_plane] if self.optim_floor else []) + \
              [self.trans[:, :1], self.root_orient[:, :1], self.latent_pose[:, :1], tv[:, :1], jv[:, :1], rv[:, :1]]
        _, views = flat_arena([tuple(t.shape) for t in src], src[0].device)
        with torch.no_grad():
            for v, t in zip(views, src):
                v.copy_(t.detach())
        for v in views:
            v.requires_grad_(True)
        it = iter(views)
        self.latent_motion, self.betas = next(it), next(it)
        if self.optim_floor:
            self.floor_plane = next(it)
        self.trans, self.root_orient, self.latent_pose = next(it), next(it), next(it)
        self.trans_vel, self.joints_vel, self.root_orient_vel = next(it), next(it), next(it)
        prior_opt_params = [self.trans_vel, self.joints_vel, self.root_orient_vel]
        motion_params = views

        return motion_params, prior_opt_params

    def _stage1_objective(self, obs_local, has_overlap):
        """Stage-1 objective on this rank's sequences (motion_optimizer.py:241-252)."""
        body_pose = self.latent2pose(self._local(self.latent_pose))
        pred, _ = self.smpl_results(self._local(self.trans), self._local(self.root_orient), body_pose, self._local(self.betas))
        halo = self._halo(pred['verts3d'], self._local(self.betas), None,
                          has_overlap and self.fitting_loss.loss_weights['rgb_overlap_consist'] > 0.0)
        loss, stats = self.fitting_loss.root_fit(obs_local, pred, halo=halo)
        return self.fitting_loss.add_next_side(loss, 'root', pred, halo), stats

    def _stage2_objective(self, obs_local, has_overlap):
        """Stage-2 objective on this rank's sequences (motion_optimizer.py:291-304)."""
        lp = self._local(self.latent_pose)
        body_pose = self.latent2pose(lp)
        pred, _ = self.smpl_results(self._local(self.trans), self._local(self.root_orient), body_pose, self._local(self.betas))
        pred['latent_pose'] = lp
        pred['betas'] = self._local(self.betas)
        halo = self._halo(pred['verts3d'], pred['betas'], None,
                          has_overlap and self.fitting_loss.loss_weights['rgb_overlap_consist'] > 0.0)
        loss, stats = self.fitting_loss.smpl_fit(obs_local, pred, self.seq_len, halo=halo)
        return self.fitting_loss.add_next_side(loss, 'smpl', pred, halo), stats

    def smpl_results_moved(self, pred, trans, root_orient, new_trans, new_root_orient):
        '''
        The smpl_results() dictionary of the same pose_body / betas under another root trajectory: `pred` was evaluated with
        (trans, root_orient) [B,T,3]; every joint and vertex moves rigidly about the root joint (csrc/rigid.hip).
        '''
        from . import _lib as _libmod
        from .fit_kernels import RigidImage
        lib = self.fit_bm._lib
        B, T, _ = trans.size()
        jtr = pred['jtr']
        verts = pred['points3d'] if self.dense_smpl else pred['verts3d']
        f3 = lambda x: x.reshape(B * T, 3)
        j2, v2 = RigidImage.apply(lib if lib is not None else _libmod.get_lib(), jtr.reshape(B * T, -1, 3), verts.reshape(B * T, -1, 3),
                                  f3(root_orient), f3(trans), f3(new_root_orient), f3(new_trans))
        joints, verts = j2.reshape(B, T, -1, 3), v2.reshape(B, T, -1, 3)
        nj = len(SMPL_JOINTS)
        out = {'joints3d': joints[:, :, :nj], 'joints3d_extra': joints[:, :, nj:], 'faces': pred['faces'], 'jtr': joints}
        if self.dense_smpl:
            out['points3d'] = verts
            out['verts3d'] = verts.index_select(2, self._keypt_idx(verts.device))
        else:
            out['verts3d'] = verts
        return out

    def _pre_stage3_eager(self, latent_pose, trans, root_orient, betas, floor, trans_vel, joints_vel, root_orient_vel):
        from . import _lib as _libmod
        from .fit_kernels import FitPre
        lib = self.fit_bm._lib
        B = trans.size(0)
        cur_body_pose = self.latent2pose(latent_pose)
        cam_smpl, _ = self.smpl_results(trans, root_orient, cur_body_pose, betas)
        o = FitPre.apply(lib if lib is not None else _libmod.get_lib(), floor, trans.reshape(B, 3), root_orient.reshape(B, 3),
                         cur_body_pose.reshape(B, J_BODY * 3), cam_smpl['joints3d'].reshape(B, 22, 3), trans_vel.reshape(B, 3),
                         joints_vel.reshape(B, 22, 3), root_orient_vel.reshape(B, 3))
        return (cur_body_pose,) + tuple(o)

    def _stage3_nodes_config(self, ref):
        """cfg of the stage-3 composite nodes (humor_amd/stage3.py) when every piece they group is available, else None."""
        lib = self.fit_bm._lib
        if not (self.fused_stage3 and self.optim_floor and self.fused_pre and self.fused_post and self.rigid_cam_body and self.fitting_loss.fused
                and not self.dense_smpl and (ref.is_cuda or (lib is not None and lib.emulator)) and getattr(self.motion_prior, 'pred_contacts', False)):
            return None
        key = (ref.device, self.fused_vposer)
        cached = getattr(self, '_stage3_cfg', None)
        if cached is not None and cached[0] == key:        # (per evaluation: this sits between L-BFGS's host read and the first launch)
            return cached[1]
        fv = self._fused_vposer(ref)
        sm = self.fit_bm.parts_config(ref.device)
        cfg = None
        if fv is not None and sm is not None and fv.dec.out_dim == 2 * J_BODY * 3:
            if lib is None:
                from . import _lib as _libmod
                lib = _libmod.get_lib()
            cfg = dict(lib=lib, smpl=sm, vposer=fv.dec, _vposer_ws={})
        self._stage3_cfg = (key, cfg)
        return cfg

    def _stage3_objective_nodes(self, cfg, obs_local, obs_init, prior_opt_params, tune_phase, n_init, init_motion_scale, og_overlap_w,
                                has_overlap):
        """_stage3_objective as three composite autograd nodes + the fused loss (humor_amd/stage3.py): the same library calls in the same
        order, but no tensor with two readers crosses a node boundary, so autograd launches no accumulation kernels between them."""
        from .stage3 import Stage3Body, Stage3Head
        L = self._local
        trans, root_orient, betas, floor = L(self.trans), L(self.root_orient), L(self.betas), L(self.floor_plane)
        tv, jv, rv = (L(p) for p in prior_opt_params)
        B = trans.size(0)
        arena = getattr(self, '_arena', None) if self.shard is not None else None
        if arena is not None:
            # sharded closure: the head's adjoint writes the variables' gradients straight into this rank's rows of the gradient arena
            named = dict(latent_pose=self.latent_pose, trans=self.trans, root_orient=self.root_orient, betas=self.betas, floor=self.floor_plane,
                         trans_vel=prior_opt_params[0], joints_vel=prior_opt_params[1], root_orient_vel=prior_opt_params[2])
            shapes = dict(latent_pose=(B, -1), trans=(B, 3), root_orient=(B, 3), joints_vel=(B, 22, 3), trans_vel=(B, 3), root_orient_vel=(B, 3))
            place = {}
            for k, p_ in named.items():
                r = arena.rows_buffer(p_)
                if r is not None:
                    place[k] = r.reshape(shapes[k]) if k in shapes else r
            cfg = dict(cfg, grad_out=place)
        (pose0, past_in, trans_p, root_p, joints_p, c2p_R, c2p_t, root_h, floor_t, tv_t, jv_t, rv_t, betas_t) = Stage3Head.apply(
            cfg, L(self.latent_pose).reshape(B, -1), trans.reshape(B, 3), root_orient.reshape(B, 3), betas, floor, tv.reshape(B, 3),
            jv.reshape(B, 22, 3), rv.reshape(B, 3))
        if self.shard is None:
            self.cam2prior_R, self.cam2prior_t, self.cam2prior_root_height = c2p_R, c2p_t, root_h
        latent_motion = L(self.latent_motion)
        if tune_phase:
            latent_motion = latent_motion[:, :(n_init - 1)]
        res = self.motion_prior.roll_out(past_in.unsqueeze(1), None, latent_motion.size(1), z_seq=latent_motion, return_prior=self.cond_prior,
                                         return_world=True, return_z=True)
        world, prior_out, z_t = res if self.cond_prior else (res[0], None, res[1])
        (pri_jtr, pri_verts, cam_jtr, cam_verts, r_trans, r_root, r_pose, ro_joints, conf, contacts, _cam_trans, _cam_root, betas_t) = Stage3Body.apply(
            cfg, world, trans_p, root_p, pose0, joints_p, c2p_R, c2p_t, betas_t)
        nj = len(SMPL_JOINTS)
        faces = self.fit_bm.bm.faces_tensor
        pred = {'joints3d': pri_jtr[:, :, :nj], 'joints3d_extra': pri_jtr[:, :, nj:], 'jtr': pri_jtr, 'verts3d': pri_verts, 'faces': faces,
                'betas': betas_t, 'latent_motion': z_t, 'joints_vel': jv_t, 'trans_vel': tv_t, 'root_orient_vel': rv_t,
                'joints3d_rollout': ro_joints, 'contacts': contacts, 'contacts_conf': conf}
        cam_pred = {'joints3d': cam_jtr[:, :, :nj], 'joints3d_extra': cam_jtr[:, :, nj:], 'jtr': cam_jtr, 'verts3d': cam_verts, 'faces': faces,
                    'betas': betas_t, 'floor_plane': floor_t}
        if self.fitting_loss.loss_weights['pose_prior'] > 0.0:
            pred['latent_pose'] = cam_pred['latent_pose'] = self.pose2latent(r_pose)
        nsteps, obs = self.seq_len, obs_local
        if tune_phase:
            nsteps, obs = n_init, obs_init
            self.fitting_loss.loss_weights['rgb_overlap_consist'] = 0.0
        halo = self._halo(cam_verts, betas_t, floor_t, has_overlap and self.fitting_loss.loss_weights['rgb_overlap_consist'] > 0.0)
        loss, stats = self.fitting_loss.motion_fit(obs, pred, cam_pred, nsteps, cond_prior=prior_out, init_motion_scale=init_motion_scale, halo=halo)
        loss = self.fitting_loss.add_next_side(loss, 'motion', cam_pred, halo)
        if tune_phase:
            self.fitting_loss.loss_weights['rgb_overlap_consist'] = og_overlap_w
        return loss, stats

    def _stage3_objective(self, obs_local, obs_init, prior_opt_params, tune_phase, n_init, init_motion_scale, og_overlap_w,
                          has_overlap, fit_gender):
        """One stage-3 objective evaluation on this rank's sequences (motion_optimizer.py:514-605)."""
        cfg = self._stage3_nodes_config(self.trans)
        if cfg is not None:
            return self._stage3_objective_nodes(cfg, obs_local, obs_init, prior_opt_params, tune_phase, n_init, init_motion_scale, og_overlap_w,
                                                has_overlap)
        L = self._local
        trans, root_orient, betas = L(self.trans), L(self.root_orient), L(self.betas)
        floor = L(self.floor_plane) if self.optim_floor else None
        cam2prior, pre = None, None
        local_prior_params = [L(p) for p in prior_opt_params]
        lib = self.fit_bm._lib
        fused_pre = self.optim_floor and self.fused_pre and (trans.is_cuda or (lib is not None and lib.emulator))
        if fused_pre:
            # VPoser decode -> frame-0 SMPL -> ha_fit_pre (cam2prior, key frame in the prior frame, initial roll-out state; the prior-
            # frame joints are the rigid image of the camera-frame ones, so this is the only frame-0 SMPL evaluation).
            # (Replaying this ~60-launch segment as its own pair of hipGraphs -- torch.cuda.make_graphed_callables -- was measured:
            # the closure went from 5.3 to 6.5 ms; two extra graph launches per direction cost more than the launches they replace.)
            out = self._pre_stage3_eager(L(self.latent_pose), trans, root_orient, betas, floor, *local_prior_params)
            cur_body_pose, o = out[0], out[1:]
            B = trans.size(0)
            pre = {'past_in': o[0], 'trans': o[1].reshape(B, 1, 3), 'root_orient': o[2].reshape(B, 1, 3), 'joints': o[3].reshape(B, 1, 22, 3)}
            cam2prior = (o[4], o[5], o[6])
        else:
            cur_body_pose = self.latent2pose(L(self.latent_pose))
            if self.optim_floor:
                cam_smpl, _ = self.smpl_results(trans, root_orient, cur_body_pose, betas)
                cam2prior = frames.compute_cam2prior(floor, trans[:, 0], ops.batch_rodrigues(root_orient[:, 0], _lib_override=lib),
                                                     cam_smpl['joints3d'][:, 0])
        if self.optim_floor and self.shard is None:
            self.cam2prior_R, self.cam2prior_t, self.cam2prior_root_height = cam2prior
        latent_motion = L(self.latent_motion)
        if tune_phase:
            latent_motion = latent_motion[:, :(n_init - 1)]
        rr, cam_rr = self.rollout_latent_motion(trans, root_orient, cur_body_pose, betas, local_prior_params, latent_motion,
                                                return_prior=self.cond_prior, fit_gender=fit_gender, cam2prior=cam2prior, pre=pre)
        # the reference encodes the rolled-out poses with VPoser on every evaluation (motion_optimizer.py:571) but only the pose
        # prior reads the result, and no stage-3 configuration weights it: skipped when its weight is zero (same loss value)
        pose_prior_on = self.fitting_loss.loss_weights['pose_prior'] > 0.0
        cur_latent_pose = self.pose2latent(rr['pose_body']) if pose_prior_on else None
        pred, _ = self.smpl_results(rr['trans'], rr['root_orient'], rr['pose_body'], betas)
        pred.update(betas=betas, latent_motion=latent_motion, joints_vel=local_prior_params[1],
                    trans_vel=local_prior_params[0], root_orient_vel=local_prior_params[2], joints3d_rollout=rr['joints'])
        if pose_prior_on:
            pred['latent_pose'] = cur_latent_pose
        if 'contacts' in rr:
            pred['contacts'], pred['contacts_conf'] = rr['contacts'], rr['contacts_conf']
        cam_pred = pred
        if self.optim_floor:
            if self.rigid_cam_body and (trans.is_cuda or (lib is not None and lib.emulator)):
                cam_pred = self.smpl_results_moved(pred, rr['trans'], rr['root_orient'], cam_rr['trans'], cam_rr['root_orient'])
            else:
                cam_pred, _ = self.smpl_results(cam_rr['trans'], cam_rr['root_orient'], rr['pose_body'], betas)
            cam_pred.update(betas=betas, floor_plane=floor)
            if pose_prior_on:
                cam_pred['latent_pose'] = cur_latent_pose
        nsteps, obs = self.seq_len, obs_local
        if tune_phase:
            nsteps, obs = n_init, obs_init
            self.fitting_loss.loss_weights['rgb_overlap_consist'] = 0.0
        halo = self._halo(cam_pred['verts3d'], betas, floor,
                          has_overlap and self.fitting_loss.loss_weights['rgb_overlap_consist'] > 0.0)
        loss, stats = self.fitting_loss.motion_fit(obs, pred, cam_pred, nsteps, cond_prior=rr.get('cond_prior'),
                                                   init_motion_scale=init_motion_scale, halo=halo)
        loss = self.fitting_loss.add_next_side(loss, 'motion', cam_pred, halo)
        if tune_phase:
            self.fitting_loss.loss_weights['rgb_overlap_consist'] = og_overlap_w
        return loss, stats

    # ------------------------------------------------------------------------------------------------
    def _stage_snapshot(self, stages_res_out, fname):
        with torch.no_grad():
            body_pose = self.latent2pose(self.latent_pose)
            pred, _ = self.smpl_results(self.trans, self.root_orient, body_pose, self.betas)
        if stages_res_out is not None:
            self._save_stage(stages_res_out, fname, body_pose)
        return pred

    def _save_stage(self, stages_res_out, fname, body_pose, contacts=None):
        """Per-sequence stage results in the reference's npz layout (motion_optimizer.py:260-270)."""
        arr = lambda t: t.detach().cpu().numpy()
        betas, trans, ro, bp = arr(self.betas), arr(self.trans), arr(self.root_orient), arr(body_pose)
        for b, out_dir in enumerate(stages_res_out):
            d = dict(betas=betas[b], trans=trans[b], root_orient=ro[b], pose_body=bp[b])
            if contacts is not None:
                d['contacts'] = arr(contacts[b])
            if self.optim_floor:
                d['floor_plane'] = arr(self.floor_plane[b])
            np.savez(os.path.join(out_dir, fname), **d)

    def _save_dicts(self, stages_res_out, fname, betas, trans, root_orient, pose_body, contacts=None, floor=None):
        """One npz per sequence in the reference's layout: betas[NB], trans[T,3], root_orient[T,3], pose_body[T,63] (+ contacts, floor)."""
        arr = lambda t: t.detach().cpu().numpy()
        be, tr, ro, bp = arr(betas), arr(trans), arr(root_orient), arr(pose_body)
        co = arr(contacts) if contacts is not None else None
        fl = arr(floor) if floor is not None else None
        for b, out_dir in enumerate(stages_res_out):
            d = dict(betas=be[b], trans=tr[b], root_orient=ro[b], pose_body=bp[b])
            if co is not None:
                d['contacts'] = co[b]
            if fl is not None:
                d['floor_plane'] = fl[b]
            np.savez(os.path.join(out_dir, fname), **d)

    def apply_cam2prior(self, data_dict, R, t, root_height, body_pose, betas, key_frame_idx, inverse=False):
        '''Camera <-> prior frame for trans / root_orient (motion_optimizer.py:678-742).'''
        lib = self.fit_bm._lib
        out = {}
        root_orient, trans = data_dict['root_orient'], data_dict['trans']
        B, T, _ = root_orient.size()
        Rm = ops.batch_rodrigues(root_orient.reshape(-1, 3), _lib_override=lib).reshape(B, T, 3, 3)
        # 3x3 products as broadcast multiply + sum: a batched rocBLAS call costs ~50 us of host time and a 10-20 us kernel for
        # a few hundred tiny matrices (and two more of each in the backward pass)
        Rt = R.unsqueeze(1)
        Rl = Rt.transpose(3, 2) if inverse else Rt
        new_R = (Rl.unsqueeze(-1) * Rm.unsqueeze(-3)).sum(-2)
        out['root_orient'] = ops.rotation_matrix_to_angle_axis(new_R.reshape(-1, 3, 3), _lib_override=lib).reshape(B, T, 3)
        # key_frame_idx is the first frame on the fitting path (motion_optimizer.py:109: init_fidx = zeros)
        assert not np.any(np.asarray(key_frame_idx)), 'non-zero key frames are not supported'
        if inverse:
            off = trans[:, 0:1]
            tr = (Rl * (trans - off).unsqueeze(-2)).sum(-1) - t.unsqueeze(1)
        else:
            tr = (Rl * (trans + t.unsqueeze(1)).unsqueeze(-2)).sum(-1)
            smpl, _ = self.smpl_results(tr, out['root_orient'], body_pose, betas)
            cur_h = smpl['joints3d'][:, 0, 0, 2:3]
            dh = root_height - cur_h
            tr = tr + torch.cat([torch.zeros(B, 2, device=tr.device, dtype=tr.dtype), dh], dim=1).reshape(B, 1, 3)
        out['trans'] = tr
        return out

    def estimate_velocities(self, trans, root_orient, body_pose, betas, data_fps, smpl_results=None):
        B, T, _ = trans.size()
        h = 1.0 / data_fps
        if smpl_results is None:
            smpl_results, _ = self.smpl_results(trans, root_orient, body_pose, betas)
        trans_vel = frames.estimate_linear_velocity(trans, h)
        joints_vel = frames.estimate_linear_velocity(smpl_results['joints3d'], h)
        Rm = ops.batch_rodrigues(root_orient.reshape(-1, 3), _lib_override=self.fit_bm._lib).reshape(B, T, 3, 3)
        return trans_vel, joints_vel, frames.estimate_angular_velocity(Rm, h)

    def infer_latent_motion(self, trans, root_orient, body_pose, betas, data_fps, full_forward_pass=False):
        '''Posterior mean of z for every transition of the current SMPL sequence (motion_optimizer.py:802-874).'''
        B, T, _ = trans.size()
        lib = self.fit_bm._lib
        if self.optim_floor:
            pd = self.apply_cam2prior({'trans': trans, 'root_orient': root_orient}, self.cam2prior_R, self.cam2prior_t,
                                      self.cam2prior_root_height, body_pose, betas, self.init_fidx)
            trans, root_orient = pd['trans'], pd['root_orient']
        smpl, _ = self.smpl_results(trans, root_orient, body_pose, betas)
        tv, jv, rv = self.estimate_velocities(trans, root_orient, body_pose, betas, data_fps, smpl_results=smpl)
        seq = {'trans': trans, 'trans_vel': tv,
               'root_orient': ops.batch_rodrigues(root_orient.reshape(-1, 3), _lib_override=lib).reshape(B, T, 9),
               'root_orient_vel': rv,
               'pose_body': ops.batch_rodrigues(body_pose.reshape(-1, 3), _lib_override=lib).reshape(B, T, J_BODY * 9),
               'joints': smpl['joints3d'].reshape(B, T, -1), 'joints_vel': jv.reshape(B, T, -1)}
        _, post = self.motion_prior.infer_global_seq(seq, full_forward_pass=full_forward_pass)
        return post[0]

    def rollout_latent_motion(self, trans, root_orient, body_pose, betas, prior_opt_params, latent_motion, return_prior=False,
                              return_vel=False, fit_gender='neutral', use_mean=False, num_steps=-1, canonicalize_input=False,
                              cam2prior=None, pre=None):
        '''
        Initial SMPL state + latent sequence -> full SMPL sequence through the motion prior
        (motion_optimizer.py:876-1019).  Returns (prior-frame dict, camera-frame dict).
        '''
        if latent_motion is None:
            raise NotImplementedError('sampling roll-out is not on the fitting path')
        lib = self.fit_bm._lib
        B, Tm1 = trans.size(0), latent_motion.size(1)
        cam2prior = cam2prior if cam2prior is not None else ((self.cam2prior_R, self.cam2prior_t, self.cam2prior_root_height)
                                                              if self.optim_floor else None)
        trans_vel, joints_vel, root_orient_vel = prior_opt_params
        if pre is not None:
            # prior-frame key frame and initial state already assembled by ha_fit_pre (see _stage3_objective)
            trans, root_orient, joints, past_in = pre['trans'], pre['root_orient'], pre['joints'], pre['past_in']
        else:
            if self.optim_floor:
                pd = self.apply_cam2prior({'trans': trans, 'root_orient': root_orient}, cam2prior[0], cam2prior[1], cam2prior[2],
                                          body_pose, betas, self.init_fidx[:B])
                trans, root_orient = pd['trans'], pd['root_orient']
            smpl0, _ = self.smpl_results(trans, root_orient, body_pose, betas)      # one frame per sequence (B rows)
            joints = smpl0['joints3d']
            R_root = ops.batch_rodrigues(root_orient.reshape(-1, 3), _lib_override=lib).reshape(B, 9)
            R_body = ops.batch_rodrigues(body_pose.reshape(-1, 3), _lib_override=lib).reshape(B, J_BODY * 9)
            past_in = torch.cat([trans.reshape(B, 3), trans_vel.reshape(B, 3), R_root, root_orient_vel.reshape(B, 3), R_body,
                                 joints.reshape(B, -1), joints_vel.reshape(B, -1)], dim=1)
        fused_post = self.fused_post and not return_vel and (past_in.is_cuda or (lib is not None and lib.emulator)) and getattr(self.motion_prior, 'pred_contacts', False)
        if fused_post:
            # everything between the roll-out and the SMPL evaluations in ONE kernel per direction (csrc/fitpost.hip): R -> axis-angle,
            # frame 0 prepended, contact confidences / labels, and the camera-frame copy of the root trajectory
            from .fit_kernels import RolloutPost
            from . import _lib as _libmod
            res = self.motion_prior.roll_out(past_in.unsqueeze(1), None, Tm1, z_seq=latent_motion, return_prior=return_prior,
                                             canonicalize_input=canonicalize_input, return_world=True)
            world, prior_out = res if return_prior else (res, None)
            c2p_R = cam2prior[0] if self.optim_floor else None
            c2p_t = cam2prior[1] if self.optim_floor else None
            o = RolloutPost.apply(lib if lib is not None else _libmod.get_lib(), world, trans.reshape(B, 3), root_orient.reshape(B, 3),
                                  body_pose.reshape(B, J_BODY * 3), joints.reshape(B, 22, 3), c2p_R, c2p_t)
            out = {'trans': o[0], 'root_orient': o[1], 'pose_body': o[2], 'joints': o[3], 'contacts_conf': o[4], 'contacts': o[5]}
            if return_prior:
                out['cond_prior'] = prior_out
            cam = {'trans': o[6], 'root_orient': o[7]} if self.optim_floor else {'trans': o[0], 'root_orient': o[1]}
            cam['pose_body'] = out['pose_body']
            return out, cam
        res = self.motion_prior.roll_out(past_in.unsqueeze(1), None, Tm1, z_seq=latent_motion, return_prior=return_prior,
                                         canonicalize_input=canonicalize_input)
        pred, prior_out = res if return_prior else (res, None)
        aa_root = ops.rotation_matrix_to_angle_axis(pred['root_orient'].reshape(-1, 3, 3), _lib_override=lib).reshape(B, Tm1, 3)
        aa_body = ops.rotation_matrix_to_angle_axis(pred['pose_body'].reshape(-1, 3, 3), _lib_override=lib).reshape(B, Tm1, J_BODY * 3)
        out = {'trans': torch.cat([trans, pred['trans']], dim=1),
               'root_orient': torch.cat([root_orient, aa_root], dim=1),
               'pose_body': torch.cat([body_pose, aa_body], dim=1),
               'joints': torch.cat([joints, pred['joints'].reshape(B, Tm1, -1, 3)], dim=1)}
        if return_vel:
            out['trans_vel'] = torch.cat([trans_vel, pred['trans_vel']], dim=1)
            out['root_orient_vel'] = torch.cat([root_orient_vel, pred['root_orient_vel']], dim=1)
            out['joints_vel'] = torch.cat([joints_vel, pred['joints_vel'].reshape(B, Tm1, -1, 3)], dim=1)
        if return_prior:
            out['cond_prior'] = prior_out
        if 'contacts' in pred:
            conf9 = torch.sigmoid(pred['contacts'])
            lab9 = (conf9 > CONTACT_THRESH).to(torch.float)
            conf = torch.zeros((B, Tm1, len(SMPL_JOINTS)), device=conf9.device, dtype=conf9.dtype)
            lab = torch.zeros_like(conf)
            conf = conf.index_add(2, self._contact_idx, conf9)
            lab = lab.index_add(2, self._contact_idx, lab9)
            out['contacts_conf'] = torch.cat([conf[:, 0:1], conf], dim=1)
            out['contacts'] = torch.cat([lab[:, 0:1], lab], dim=1)
        cam = {}
        if self.optim_floor:
            cam = self.apply_cam2prior({'trans': out['trans'], 'root_orient': out['root_orient']}, cam2prior[0], cam2prior[1],
                                       cam2prior[2], out['pose_body'], betas, self.init_fidx[:B], inverse=True)
        else:
            cam['trans'], cam['root_orient'] = out['trans'], out['root_orient']
        cam['pose_body'] = out['pose_body']
        return out, cam

    def get_optim_result(self, body_pose=None):
        if body_pose is None:
            body_pose = self.latent2pose(self.latent_pose)
        d = lambda t: t.clone().detach()
        res = {'trans': d(self.trans), 'root_orient': d(self.root_orient), 'pose_body': d(body_pose), 'betas': d(self.betas),
               'latent_pose': d(self.latent_pose), 'latent_motion': d(self.latent_motion)}
        if self.optim_floor:
            res['floor_plane'] = d(frames.parse_floor_plane(self.floor_plane))
        return res

    def _fused_vposer(self, x):
        """The ha_mlp_* handles of the pose prior for tensors on a HIP device (or the emulator tier), else None."""
        if not self.fused_vposer:
            return None
        lib = self.fit_bm._lib
        if not (x.is_cuda or (lib is not None and lib.emulator)):
            return None
        if self._vposer_handle is None:
            from . import _lib as _libmod
            from .mlp import FusedVPoser
            try:
                self._vposer_handle = FusedVPoser(self.pose_prior, lib if lib is not None else _libmod.get_lib(), x.device.index or 0)
            except NotImplementedError as e:
                import warnings
                warnings.warn(f'humor_amd: pose prior evaluated through its own PyTorch modules ({e}); call pose_prior.eval() '
                              f'as run_fitting.py does to enable the fused VPoser kernels')
                self._vposer_handle = False      # not a VPoser v1.0-shaped module: evaluate it as given
        return self._vposer_handle or None

    def latent2pose(self, latent_pose):
        '''VPoser latent -> axis-angle body pose.  [B,T,D] -> [B,T,63]'''
        B, T, _ = latent_pose.size()
        fv = self._fused_vposer(latent_pose)
        if fv is not None:
            return fv.decode_aa(latent_pose.reshape(-1, self.latent_pose_dim)).reshape(B, T, J_BODY * 3)
        mats = self.pose_prior.decode(latent_pose.reshape(-1, self.latent_pose_dim), output_type='matrot')
        return ops.rotation_matrix_to_angle_axis(mats.reshape(B * T * J_BODY, 3, 3), _lib_override=self.fit_bm._lib).reshape(B, T, J_BODY * 3)

    def pose2latent(self, body_pose):
        B, T, _ = body_pose.size()
        fv = self._fused_vposer(body_pose)
        if fv is not None:
            return fv.encode_mean(body_pose.reshape(-1, J_BODY * 3)).reshape(B, T, self.latent_pose_dim)
        return self.pose_prior.encode(body_pose.reshape(-1, J_BODY * 3)).mean.reshape(B, T, self.latent_pose_dim)

    def _keypt_idx(self, device):
        if getattr(self, '_keypt_t', None) is None or self._keypt_t.device != device:
            self._keypt_t = torch.as_tensor(KEYPT_VERTS, dtype=torch.long, device=device)
        return self._keypt_t

    def smpl_results(self, trans, root_orient, body_pose, beta):
        '''
        SMPL forward for [B,T,.] parameters (T = 1 or any length): joints3d [B,T,22,3], joints3d_extra, verts3d [B,T,43,3].
        '''
        B, T, _ = trans.size()
        betas = beta.reshape(B, 1, self.num_betas).expand(B, T, self.num_betas).reshape(B * T, -1)
        body = self.fit_bm(pose_body=body_pose.reshape(B * T, -1), pose_hand=None, betas=betas,
                           root_orient=root_orient.reshape(B * T, -1), trans=trans.reshape(B * T, -1))
        joints = body.Jtr.reshape(B, T, -1, 3)
        nj = len(SMPL_JOINTS)
        verts = body.v.reshape(B, T, -1, 3)
        pred = {'joints3d': joints[:, :, :nj], 'joints3d_extra': joints[:, :, nj:], 'faces': body.f,
                'jtr': joints}          # 'jtr': the undivided joint tensor (what the fused loss kernel reads)
        if self.dense_smpl:
            pred['points3d'] = verts                                   # every vertex (chamfer term), key vertices gathered from it
            pred['verts3d'] = verts.index_select(2, self._keypt_idx(verts.device))
        else:
            pred['verts3d'] = verts
        return pred, body
