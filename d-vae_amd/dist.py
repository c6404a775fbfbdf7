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

from .batching import shard_bounds, balanced_shard_bounds


class GradSumReducer(object):
    """Keeps the gradients of the parameters the train step touches as views into
    ONE flat fp32 buffer, so the exchange step is a single all-reduce with no
    packing copies.  Parameters the step never touches (hv_unify; the backward
    encoder's when unidirectional) keep grad=None, as in the reference, so the
    optimizer skips them identically on every rank."""

    def __init__(self, model, live_names=None, sink=True):
        named = dict(model.named_parameters())
        if live_names is None:
            # decoder-only parameters first: their gradients are final when loss.backward() has run
            # the decoder, so that part of the buffer is reduced while the encoder's backward is
            # still computing (the rest, the encoder's and the shared gate/mapper, follows)
            enc = list(model._enc_names)
            early = [k for k in model._dec_names if k not in enc]
            live_names = early + list(dict.fromkeys(enc + list(model._dec_names)))
            live_names = list(dict.fromkeys(live_names))
            n_early_names = len(early)
        else:
            n_early_names = 0
        self.params = [named[k] for k in live_names]
        self.names = list(live_names)
        # every slice starts on a 256-byte boundary (vector loads / stores of the optimizer kernel)
        self.offsets = []
        off = 0
        self.n_early = 0
        for i, p in enumerate(self.params):
            if i == n_early_names:
                self.n_early = off
            self.offsets.append(off)
            off += (p.numel() + 63) // 64 * 64
        if n_early_names == len(self.params):
            self.n_early = off
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=self.params[0].device)
        for p, o in zip(self.params, self.offsets):
            p.grad = self.flat[o:o + p.numel()].view_as(p)
        self._early = None
        # overlap = True: the decoder-only part is all-reduced asynchronously from inside backward(), beside the
        # encoder's backward; False: one all-reduce of the whole buffer after backward() (bench.py --allreduce after)
        self.overlap = True
        # timing hook (bench.py): when set to a list, decoder_grads_ready() appends a timing event recorded on the
        # compute stream at the point where the decoder's gradients are final (= where the early part is launched)
        self.mark_decoder_done = None
        self.extra_params = []
        # True while the flat buffer is known to hold zeros (FlatAdam's kernel leaves it zeroed after a step and the
        # model clears the flag before it adds gradients): zero() is then free.  Code that writes into the .grad views
        # by hand between an optimizer step and zero() would have those writes wiped by zero() anyway.
        self.clean = False
        # backward() then adds straight into these views (no autograd accumulation kernels)
        if sink and hasattr(model, "attach_grad_sink") and self.flat.is_cuda:
            model.attach_grad_sink(True, self)

    def add_extra(self, params):
        """Parameters outside the flat buffer whose gradients must be summed over the ranks as well: the optional
        predictor head (train.py:165-172), whose MSE loss is a SUM over the rank's graphs (reduction='sum').
        Their .grad tensors are owned by autograd; all_reduce() exchanges them coalesced in one more small
        all-reduce."""
        have = {id(p) for p in self.extra_params} | {id(p) for p in self.params}
        self.extra_params += [p for p in params if id(p) not in have]

    def _reduce_extra(self):
        grads = [p.grad for p in self.extra_params if p.grad is not None]
        if not grads:
            return
        if len(grads) != len(self.extra_params):
            raise RuntimeError("some extra parameters have no gradient on this rank: the ranks would disagree on "
                               "what to exchange")
        flat = torch.cat([g.reshape(-1) for g in grads])
        if flat.is_cuda and dist.get_backend() == "gloo":
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat = host.to(flat.device)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        o = 0
        for g in grads:
            g.copy_(flat[o:o + g.numel()].view_as(g))
            o += g.numel()

    @staticmethod
    def _collective():
        """is there an exchange step?  (DVAE_FORCE_COLLECTIVE=1 runs it in a 1-rank group too, which
        is how the RCCL path is exercised on a one-GPU box)"""
        if not (dist.is_available() and dist.is_initialized()):
            return False
        import os
        return dist.get_world_size() > 1 or os.environ.get("DVAE_FORCE_COLLECTIVE") == "1"

    def decoder_grads_ready(self):
        """Called by the model when loss.backward() has added the decoder's gradients: starts the
        SUM all-reduce of the decoder-only part asynchronously (RCCL orders it after the work
        already enqueued on the compute stream) so that it overlaps the encoder's backward."""
        if self.mark_decoder_done is not None and self.flat.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.flat.device))
            self.mark_decoder_done.append(ev)
        if not self.overlap or self.n_early == 0 or self._early is not None or not self._collective():
            return
        if not self.flat.is_cuda or dist.get_backend() != "nccl":
            return
        self._early = dist.all_reduce(self.flat[:self.n_early], op=dist.ReduceOp.SUM, async_op=True)

    def zero(self):
        if self._early is not None:  # a step was abandoned between backward() and all_reduce()
            self._early.wait()
            self._early = None
        if self.clean and self.flat.is_cuda:
            return
        self.flat.zero_()

    def all_reduce(self):
        if not self._collective():
            return
        if self.extra_params:
            self._reduce_extra()
        if self.flat.is_cuda and dist.get_backend() == "gloo":
            # rehearsal path (several ranks sharing one GPU): gloo reduces on the host
            host = self.flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            self.flat.copy_(host)
            return
        if self._early is not None:
            self._early.wait()  # the compute stream waits for the early part
            self._early = None
            if self.n_early < self.numel:
                dist.all_reduce(self.flat[self.n_early:], op=dist.ReduceOp.SUM)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)  # RCCL over xGMI when backend == "nccl"

    def reduce_scalars(self, *values):
        """SUM over ranks of a few scalars in one small all-reduce: the (loss, recon, kld) of a step
        or an epoch, for logging only (every rank's loss is the sum over its own graphs).  Returns
        a float64 tensor on the values' device; a no-op copy without a process group."""
        t = torch.stack([torch.as_tensor(v).detach().double().reshape(()) for v in values])
        if self._collective():
            if t.is_cuda and dist.get_backend() == "gloo":
                host = t.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                t = host.to(t.device)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def nbytes(self):
        return self.flat.numel() * 4


def shard(items, rank=None, world_size=None, balance=False):
    """This rank's contiguous chunk of `items` (list or GraphArrays): the reference's ceil-split by
    graph count, or with balance=True (GraphArrays of variable-size graphs) the split that evens
    out the sum of n_g^2 per rank."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if balance:
        lo, hi = balanced_shard_bounds(items.n_vertices, world_size, rank)
    else:
        lo, hi = shard_bounds(len(items), world_size, rank)
    return items[lo:hi]
