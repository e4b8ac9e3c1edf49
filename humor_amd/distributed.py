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
