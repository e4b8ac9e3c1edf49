"""Multi-GPU data parallelism of the fitting closure (one process per GPU, torch.distributed; backend "nccl" = RCCL
over xGMI on the MI355X node, "gloo" in the CPU tests).  New functionality: the reference is single-GPU.

Design: *replicated optimiser, sharded closure*.  Every rank holds all optimisation variables (<= 0.4 MB at 32x60) and
runs the identical torch.optim.LBFGS; inside the closure a rank evaluates loss and gradient only for its contiguous
slice of sub-sequences, then ONE all-reduce(SUM) of the packed [flat gradient | loss] vector makes the result -- and
therefore every line-search decision -- identical on all ranks.  The overlap-consistency terms couple sequence b-1 and b;
when b-1 lives on the previous rank its predicted tail is obtained with a differentiable all-gather (forward all_gather,
backward all-reduce of the gathered gradient), so gradients are exact and each pair is counted once.
=> 2 small collectives per closure forward (+1 in backward), all latency-bound (<= 170 KB).
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


class _AllGatherGrad(torch.autograd.Function):
    """x [L] -> [world, L]; backward all-reduces the gathered gradient and returns this rank's row."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        world = dist.get_world_size(group)
        out = [torch.empty_like(x) for _ in range(world)]
        _all_gather(out, x.contiguous(), group)
        return torch.stack(out, dim=0)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g[dist.get_rank(ctx.group)], None


def all_gather_with_grad(x, group=None):
    return _AllGatherGrad.apply(x, group)


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
