"""Multi-GPU data parallelism of the fitting closure (one process per GPU, torch.distributed; backend "nccl" = RCCL
over xGMI on the MI355X node, "gloo" in the CPU tests).  New functionality: the reference is single-GPU.

Design: *replicated optimiser, sharded closure*.  Every rank holds all optimisation variables (<= 0.4 MB at 32x60) and
runs the identical torch.optim.LBFGS; inside the closure a rank evaluates loss and gradient only for its contiguous
slice of sub-sequences, then ONE all-reduce(SUM) of the packed [flat gradient | loss] vector makes the result -- and
therefore every line-search decision -- identical on all ranks.  The overlap-consistency terms couple sequence b-1 and b;
when they live on different ranks BOTH ranks evaluate the pair from a forward-only all_gather of the boundary sequences'
overlapping frames (SURVEY.md 8(e) option B): the owner of b counts its value, the owner of b-1 adds only its own gradient
(the terms are squared differences), so gradients are exact, each pair is counted once and the backward pass has no collective.
=> 2 small collectives per closure evaluation (halo all_gather <= 90 KB, gradient all-reduce <= 0.4 MB), both latency-bound.
"""
import torch
import torch.distributed as dist


class Shard:
    """Contiguous partition of B sub-sequences over the ranks of `group`."""

    def __init__(self, B, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        base, rem = divmod(B, self.world)
        sizes = [base + (1 if r < rem else 0) for r in range(self.world)]
        self.b0 = sum(sizes[:self.rank])
        self.b1 = self.b0 + sizes[self.rank]
        self.B = B
        if min(sizes) < 1:
            raise ValueError(f'cannot shard {B} sequences over {self.world} ranks')

    def sl(self, x):
        return x[self.b0:self.b1]


def _all_gather(out, x, group):
    """dist.all_gather; the gloo backend (CPU tests, and two test ranks sharing one GPU) cannot gather device tensors, so
    those go through host memory.  RCCL ('nccl') gathers device tensors directly."""
    if x.is_cuda and dist.get_backend(group) == 'gloo':
        host = [torch.empty(x.shape, dtype=x.dtype) for _ in out]
        dist.all_gather(host, x.cpu(), group=group)
        for o, h in zip(out, host):
            o.copy_(h)
    else:
        dist.all_gather(out, x, group=group)


def all_gather_flat(x, group=None):
    """x [L] (no gradient) -> [world, L].  One collective: all_gather_into_tensor on RCCL, the list form through gloo."""
    world = dist.get_world_size(group)
    x = x.detach().contiguous()
    out = torch.empty(world, x.numel(), dtype=x.dtype, device=x.device)
    if dist.get_backend(group) == 'gloo':
        _all_gather(list(out.unbind(0)), x, group)
    else:
        dist.all_gather_into_tensor(out, x, group=group)
    return out


def allreduce_loss_and_grads(loss, params, group=None):
    """Sums loss and the .grad of every parameter over the ranks with a single packed all-reduce.
    Parameters whose grad is None (frozen this phase) are skipped consistently on all ranks."""
    live = [p for p in params if p.requires_grad]
    flat = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in live]
    packed = torch.cat(flat + [loss.detach().reshape(1)])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    o = 0
    for p in live:
        n = p.numel()
        g = packed[o:o + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        o += n
    return packed[-1].clone()


class GradArena:
    """Packed [gradients of the live parameters | loss] buffer of a sharded closure, persistent across evaluations.

    Without it an evaluation pays, per parameter, autograd's slice_backward (a full-size zero fill + a copy of the rank's rows), a `cat` into
    the packed vector and a copy back out of it after the all-reduce: ~35 small launches (0.09 ms of a 1.75 ms evaluation on one MI355X).
    With it: `begin()` zero-fills the arena once, `rows(p)` hands the closure this rank's rows of p through an autograd node whose backward
    drops the incoming gradient into the arena's rows (one copy) and returns the arena's full-size view, the all-reduce runs in place on
    the arena, and `p.grad` of every live parameter is left as a view of it (no copies out).  A parameter's rows are handed out ONCE per
    evaluation (later calls return the same tensor), so autograd sums its readers' gradients before the one backward call.
    The arena is overwritten by the next evaluation: callers that keep a gradient across evaluations clone it (the L-BFGS implementations do)."""

    def __init__(self, params, shard):
        self.shard = shard
        self.live = [p for p in params if p.requires_grad]
        self.sizes = [p.numel() for p in self.live]
        ref = self.live[0]
        self.buf = torch.zeros(sum(self.sizes) + 1, dtype=ref.dtype, device=ref.device)
        self.views, o = {}, 0
        for p, n in zip(self.live, self.sizes):
            self.views[id(p)] = self.buf[o:o + n].view_as(p)
            o += n
        self._rows = {}

    def matches(self, params):
        live = [p for p in params if p.requires_grad]
        return len(live) == len(self.live) and all(a is b for a, b in zip(live, self.live))

    def begin(self):
        self.buf.zero_()
        self._rows = {}

    def rows(self, p):
        """This rank's rows of parameter p (None: p is not one of the arena's parameters)."""
        view = self.views.get(id(p))
        if view is None:
            return None
        out = self._rows.get(id(p))
        if out is None:
            out = self._rows[id(p)] = _ShardRows.apply(p, self.shard.b0, self.shard.b1, view)
        return out

    def rows_buffer(self, p):
        """The arena's rows b0:b1 of parameter p (where a kernel may write dL/d(rows of p) directly), or None."""
        view = self.views.get(id(p))
        return None if view is None else view[self.shard.b0:self.shard.b1]

    def allreduce(self, loss, group=None):
        """Sums the arena (and the loss) over the ranks in place; leaves p.grad = the arena's view for every live parameter."""
        for p in self.live:
            view = self.views[id(p)]
            g = p.grad
            if g is None:
                continue                                   # no gradient reached p: its rows are still zero
            if g.data_ptr() != view.data_ptr():
                view.copy_(g)                              # (a gradient that did not come through rows())
        self.buf[-1:].copy_(loss.detach().reshape(1))
        dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=group)
        for p in self.live:
            p.grad = self.views[id(p)]
        return self.buf[-1].clone()


class _ShardRows(torch.autograd.Function):
    """x [B, ...] -> x[b0:b1]; backward: the incoming gradient goes into rows b0:b1 of `full` (a view of the GradArena, zero elsewhere) and
    `full` is returned -- what SliceBackward computes, without allocating and zero-filling a full-size tensor per evaluation."""

    @staticmethod
    def forward(ctx, x, b0, b1, full):
        ctx.b0, ctx.b1, ctx.full = b0, b1, full
        return x[b0:b1]

    @staticmethod
    def backward(ctx, g):
        rows = ctx.full[ctx.b0:ctx.b1]
        if g.data_ptr() != rows.data_ptr():
            rows.copy_(g)
        # a fresh view object: AccumulateGrad takes a gradient over without cloning it only when nobody else holds the tensor object
        return ctx.full.view_as(ctx.full), None, None, None
