"""Multi-GPU data parallelism for the train step: one process per GPU, the graph
batch sharded by rank, one SUM all-reduce of a flat gradient buffer (RCCL over
xGMI when the backend is "nccl"; "gloo" on CPU for tests).

Replaces the reference's only parallel strategy, single-process
`nn.DataParallel` over igraph lists (`/root/reference/util.py:557-573`,
`train.py:179-180,227-230`), which broadcasts all parameters every step and
reduce-adds gradients onto device 0.  The loss is a SUM over graphs
(`models.py:825,855,859`), so gradients are SUMMED, not averaged.
"""
import torch
import torch.distributed as dist

from .batching import shard_bounds


class GradSumReducer(object):
    """Keeps the gradients of the parameters the train step touches as views into
    ONE flat fp32 buffer, so the exchange step is a single all-reduce with no
    packing copies.  Parameters the step never touches (hv_unify; the backward
    encoder's when unidirectional) keep grad=None, as in the reference, so the
    optimizer skips them identically on every rank."""

    def __init__(self, model, live_names=None, sink=True):
        named = dict(model.named_parameters())
        if live_names is None:
            live_names = list(dict.fromkeys(list(model._enc_names) + list(model._dec_names)))
        self.params = [named[k] for k in live_names]
        self.names = list(live_names)
        # every slice starts on a 256-byte boundary (vector loads / stores of the optimizer kernel)
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 63) // 64 * 64
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=self.params[0].device)
        for p, o in zip(self.params, self.offsets):
            p.grad = self.flat[o:o + p.numel()].view_as(p)
        # backward() then adds straight into these views (no autograd accumulation kernels)
        if sink and hasattr(model, "attach_grad_sink") and self.flat.is_cuda:
            model.attach_grad_sink(True)

    def zero(self):
        self.flat.zero_()

    def all_reduce(self):
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        if self.flat.is_cuda and dist.get_backend() == "gloo":
            # rehearsal path (several ranks sharing one GPU): gloo reduces on the host
            host = self.flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            self.flat.copy_(host)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)  # RCCL over xGMI when backend == "nccl"

    def nbytes(self):
        return self.flat.numel() * 4


def shard(items, rank=None, world_size=None):
    """This rank's contiguous ceil-split chunk of `items` (list or GraphArrays)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_bounds(len(items), world_size, rank)
    return items[lo:hi]
