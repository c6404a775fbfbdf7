"""Epoch-level harness around the GPU train step: the loops of the reference's experiment
driver that call the model (`/root/reference/train.py`), as plain functions with explicit
arguments instead of module globals and argparse state.

  train_epoch        train.py:209-262   shuffle, batch, zero_grad / encode / loss / backward / step
  test               train.py:282-330   NLL + reconstruction accuracy (encode_times x decode_times decodings)
  extract_latent     train.py:383-404   latent means of a data set
  prior_validity     train.py:333-380   validity / uniqueness / novelty of decodings from the prior
  save_checkpoint    train.py:705-712   model / optimizer / scheduler state dicts per epoch
  load_module_state  util.py:38-60      filtered (unknown keys dropped) state-dict load

Data sets come in the two forms of `data.load_ENAS_graphs` / `load_BN_graphs`: a list of
(graph, y) pairs (the reference's form) or a pre-packed (GraphArrays, y[]) pair.  Plots,
progress bars, latent-space figures and the Bayesian-optimisation scripts are not rebuilt.
"""
import os
import random

import numpy as np
import torch

from .batching import GraphArrays, graphs_to_arrays
from . import data as D


def _as_arrays(dataset, max_n):
    """(GraphArrays, y) whatever form the data set came in"""
    if isinstance(dataset, tuple) and isinstance(dataset[0], GraphArrays):
        return dataset[0], np.asarray(dataset[1], np.float64)
    graphs = [g for g, _ in dataset]
    ys = np.asarray([y for _, y in dataset], np.float64)
    return graphs_to_arrays(graphs, max_n), ys


def _batches(n_items, batch_size):
    return [(lo, min(lo + batch_size, n_items)) for lo in range(0, n_items, batch_size)]


def _packed_batches(model, ga, order, batch_size, prefetch, caps=None):
    """Yields (indices, packed DAGBatch).  With prefetch the next batches are packed (C++ packer,
    which releases the GIL, + the H2D copy of the index blob) by a background thread while the GPU
    works on the current one."""
    spans = _batches(len(order), batch_size)
    if not prefetch or len(spans) < 2:
        for lo, hi in spans:
            idx = order[lo:hi]
            yield idx, model._collate_fn(ga.select(idx), caps)
        return
    import queue
    import threading
    q = queue.Queue(maxsize=3)
    stop = threading.Event()
    dev = model.get_device()

    def put(item):
        """blocks while the queue is full, but gives up when the consumer has gone away"""
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return True
            except queue.Full:
                pass
        return False

    def work():
        try:
            if dev.type == "cuda":
                torch.cuda.set_device(dev)
            for lo, hi in spans:
                idx = order[lo:hi]
                if not put((idx, model._collate_fn(ga.select(idx), caps))):
                    return
            put(None)
        except BaseException as e:  # surfaces in the consumer
            put(e)

    t = threading.Thread(target=work, daemon=True)
    t.start()
    try:
        while True:
            item = q.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            yield item
    finally:
        # the consumer stopped (end of data, an exception in the training step, or the generator was closed):
        # release the worker and the batches it still holds
        stop.set()
        while True:
            try:
                q.get_nowait()
            except queue.Empty:
                break
        t.join()


class _StaticBatch(object):
    """What encode() / loss() need of a batch, with the index blob in a FIXED device buffer (the captured step reads it)"""
    is_dag_batch = True

    def __init__(self, like, blob):
        self.B, self.max_n, self.nvt, self.is_bn = like.B, like.max_n, like.nvt, like.is_bn
        self.counts = like.counts
        self.blob = blob
        self.device = like.device
        self.arrays = None

    def wait_ready(self):
        pass

    def __len__(self):
        return self.B


class GraphedTrainStep(object):
    """One train step (train.py:224-253: zero_grad -> encode -> loss -> backward -> Adam) captured ONCE as a HIP graph
    and replayed for every later batch: the host then pays two small copies and one graph launch per step instead of
    ~120 Python / launch calls (profiles/r03_host_cost_cfg1.txt), which is what bounds the reference's own recipe
    (batch 32) once the GPU side is fast.

    Needs a launch sequence that does not depend on the batch: batches packed shape-stable (batching.padding_caps ->
    model._collate_fn(G, caps=...)), an optimizer whose per-step scalars live on the device
    (optim.FlatAdam(capturable=True)) and the gradient sink (no autograd accumulation kernels).  The first `warmup`
    eligible batches run eagerly on the capture stream (they are ordinary train steps; the library creates its side
    streams there), the next one is captured, and from then on every eligible batch is a replay.  A batch that is not
    eligible -- other counts (the epoch's last, smaller batch; a batch above the caps), `eps` given to a step captured
    without it or the reverse -- runs the eager step, with the same result.

    Single process only: the gradient all-reduce is not captured (N > 1 ranks keep the eager step)."""

    def __init__(self, model, optimizer, beta=0.005, warmup=2):
        if not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedTrainStep needs optim.FlatAdam(..., capturable=True)")
        if optimizer.extra is not None:
            raise ValueError("GraphedTrainStep: extra parameters (predictor head) are stepped by torch's Adam; not captured")
        self.model, self.opt, self.beta = model, optimizer, beta
        self.warmup = int(warmup)
        self.n_warm = 0
        self.graph = None
        self.key = None
        self.stream = torch.cuda.Stream(device=model.get_device())
        self.replays = 0
        self.eager_steps = 0

    @staticmethod
    def _key(batch, eps):
        c = batch.counts
        return tuple(getattr(c, f) for f, _ in c._fields_) + (eps is not None,)

    def _eager(self, batch, eps):
        self.opt.zero_grad()
        mu, logvar = self.model.encode(batch)
        loss, recon, kld = self.model.loss(mu, logvar, batch, beta=self.beta, eps=eps)
        loss.backward()
        self.opt.step()
        self.eager_steps += 1
        return loss.detach(), recon.detach(), kld.detach()

    def _capture(self, batch, eps):
        dev = self.model.get_device()
        self.static_blob = torch.empty_like(batch.blob)
        self.static_eps = torch.empty(batch.B, self.model.nz, dtype=torch.float32, device=dev) if eps is not None else None
        self.static_batch = _StaticBatch(batch, self.static_blob)
        self.opt.sync_device_state()  # nothing inside the capture may have to touch lr / steps on the device
        self.graph = torch.cuda.CUDAGraph()
        cur = torch.cuda.current_stream(dev)
        self.stream.wait_stream(cur)
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.opt.zero_grad()
            mu, logvar = self.model.encode(self.static_batch)
            loss, recon, kld = self.model.loss(mu, logvar, self.static_batch, beta=self.beta, eps=self.static_eps)
            loss.backward()
            self.opt.step()
            self.out = (loss.detach(), recon.detach(), kld.detach())
        # (capturing executed nothing: Adam's host-side step count moved, the device's has not; the replay that
        # follows IS that step)
        self.key = self._key(batch, eps)

    def __call__(self, batch, eps=None):
        """-> (loss, recon, kld) of the step, 0-dim device tensors (replays: the same three static tensors every time:
        read or accumulate them before the next call)"""
        ok = (getattr(batch, "padded", False) and self.model.training and self.model._grad_sink and
              not self.opt.reducer._collective())
        if not ok or (self.key is not None and self._key(batch, eps) != self.key):
            return self._eager(batch, eps)
        if self.graph is None:
            if self.n_warm < self.warmup:
                self.n_warm += 1
                cur = torch.cuda.current_stream(self.model.get_device())
                self.stream.wait_stream(cur)
                with torch.cuda.stream(self.stream):
                    out = self._eager(batch, eps)
                cur.wait_stream(self.stream)
                return out
            self._capture(batch, eps)
            first = True
        else:
            first = False
        batch.wait_ready()
        self.static_blob.copy_(batch.blob, non_blocking=True)
        if eps is not None:
            self.static_eps.copy_(eps, non_blocking=True)
        if not first:
            self.opt.sync_device_state()  # a scheduler may have changed lr
        self.graph.replay()
        if not first:
            self.opt.replayed()
        self.replays += 1
        return self.out


def train_epoch(model, optimizer, train_data, batch_size=32, rng=random, predictor=False, reducer=None,
                on_batch=None, prefetch=False, graphed=None, caps=None):
    """One pass over the training set (train.py:209-262).  Returns the SUMS over the epoch of
    (loss, recon, kld[, pred]) exactly as the reference accumulates them.  `rng` shuffles the
    data (the reference shuffles its list in place with the global `random`); `reducer`
    (dist.GradSumReducer) makes it data parallel: every rank passes its own shard and gets the
    whole-job sums back.  `prefetch=True` packs the next batches in a background thread; off by
    default: the C++ packer takes ~20 us for 32 graphs, ~0.3 ms for 1024, the host runs ahead of the
    GPU anyway, and the second thread costs the enqueueing one more than it saves (epoch at batch
    32: 29.0 k DAGs/s without, 25.9 k with; at batch 1024: 238 k / 235 k).
    `graphed` (a GraphedTrainStep for this model and optimizer) + `caps` (batching.padding_caps of the data set):
    the batches are packed shape-stable and the step is replayed from one captured HIP graph; batches that do not
    fit (the last, smaller one) run the eager step inside it."""
    model.train()
    ga, ys = _as_arrays(train_data, model.max_n)
    order = list(range(len(ga)))
    rng.shuffle(order)
    dev = model.get_device()
    tot = torch.zeros(4, dtype=torch.float64, device=dev)
    done = 0
    if graphed is not None and (predictor or reducer is not None):
        raise ValueError("train_epoch: the captured step covers neither the predictor head nor the all-reduce")
    for idx, batch in _packed_batches(model, ga, order, batch_size, prefetch, caps):
        if graphed is not None:
            l3 = graphed(batch)
            tot[:3] += torch.stack(l3).double()
            done += len(idx)
            if on_batch is not None:
                on_batch(done, len(order))
            continue
        optimizer.zero_grad()
        mu, logvar = model.encode(batch)
        loss, recon, kld = model.loss(mu, logvar, batch)
        if not predictor:
            # (loss, recon, kld) are views of the library's one 3-float output: ONE accumulation launch per step instead
            # of stack + cast + add (the host is the bound at batch 32: every torch call around the step is ~8 us of it)
            out3 = loss._base
            loss.backward()
            if reducer is not None:
                reducer.all_reduce()
            optimizer.step()
            if out3 is not None and out3.numel() == 3:
                tot[:3] += out3.detach()
            else:
                tot[:3] += torch.stack([loss.detach(), recon.detach(), kld.detach()]).double()
            done += len(idx)
            if on_batch is not None:
                on_batch(done, len(order))
            continue
        pred = torch.zeros((), device=dev)
        if predictor:
            y = torch.as_tensor(ys[idx], dtype=torch.float32, device=dev).unsqueeze(1)
            pred = model.mseloss(model.predictor(mu), y)
            loss = loss + pred
        loss.backward()
        if reducer is not None:
            reducer.all_reduce()
        optimizer.step()
        tot += torch.stack([loss.detach(), recon.detach(), kld.detach(), pred.detach()]).double()  # no host sync
        done += len(idx)
        if on_batch is not None:
            on_batch(done, len(order))
    if reducer is not None:
        tot = reducer.reduce_scalars(*tot.unbind(0))  # whole-job sums on every rank: one 4-float all-reduce per epoch
    t = tot.tolist()
    return (t[0], t[1], t[2], t[3]) if predictor else (t[0], t[1], t[2])


@torch.no_grad()
def extract_latent(model, dataset, infer_batch_size=128):
    """train.py:383-404: (Z [N, nz] latent means, Y [N])"""
    model.eval()
    ga, ys = _as_arrays(dataset, model.max_n)
    Z = []
    for lo, hi in _batches(len(ga), infer_batch_size):
        mu, _ = model.encode(ga[lo:hi])
        Z.append(mu)
    return torch.cat(Z, 0).cpu().numpy(), ys


def _same_graph_count(a, b):
    """number of positions where the two GraphArrays hold the same DAG (is_same_DAG, util.py:466-476:
    same size, same vertex types, same predecessor sets)"""
    same = (a.n_vertices == b.n_vertices)
    live = np.arange(a.max_n)[None, :] < a.n_vertices[:, None]  # entries past a graph's size carry no meaning
    same &= ((a.types == b.types) | ~live).all(1)
    same &= ((a.pred_mask == b.pred_mask) | ~live).all(1)
    return int(same.sum())


@torch.no_grad()
def test(model, test_data, infer_batch_size=128, encode_times=10, decode_times=10, predictor=False):
    """train.py:282-330: average NLL of the test graphs and reconstruction accuracy = fraction of
    exact reconstructions over encode_times samples of z x decode_times stochastic decodings."""
    model.eval()
    ga, ys = _as_arrays(test_data, model.max_n)
    dev = model.get_device()
    nll_sum, pred_sum, n_perfect = 0.0, 0.0, 0
    for lo, hi in _batches(len(ga), infer_batch_size):
        g = ga[lo:hi]
        batch = model._collate_fn(g)
        mu, logvar = model.encode(batch)
        _, nll, _ = model.loss(mu, logvar, batch)
        nll_sum += nll.item()
        if predictor:
            y = torch.as_tensor(ys[lo:hi], dtype=torch.float32, device=dev).unsqueeze(1)
            pred_sum += model.mseloss(model.predictor(mu), y).item()
        for _ in range(encode_times):
            z = model.reparameterize(mu, logvar)
            for _ in range(decode_times):
                n_perfect += _same_graph_count(g, model.decode_arrays(z))
    n = len(ga)
    acc = n_perfect / float(n * encode_times * decode_times)
    if predictor:
        return nll_sum / n, acc, (pred_sum / n) ** 0.5
    return nll_sum / n, acc


@torch.no_grad()
def prior_validity(model, train_data, data_type="ENAS", n_latent_points=1000, decode_times=10, infer_batch_size=128,
                   scale_to_train_range=False):
    """train.py:333-380: decode points drawn from the prior; returns (valid, unique, novel) ratios."""
    model.eval()
    dev = model.get_device()
    z_mean = z_std = None
    if scale_to_train_range:
        Z, _ = extract_latent(model, train_data, infer_batch_size)
        z_mean = torch.as_tensor(Z.mean(0), dtype=torch.float32, device=dev)
        z_std = torch.as_tensor(Z.std(0), dtype=torch.float32, device=dev)
    valid_fn = D.is_valid_ENAS if data_type == "ENAS" else (lambda g, s, e: D.is_valid_BN(g, s, e, nvt=model.nvt))
    G_valid = []
    for lo, hi in _batches(n_latent_points, infer_batch_size):
        z = torch.randn(hi - lo, model.nz, device=dev)
        if scale_to_train_range:
            z = z * z_std + z_mean
        for _ in range(decode_times):
            for g in model.decode(z):
                if valid_fn(g, model.START_TYPE, model.END_TYPE):
                    G_valid.append(g)
    r_valid = len(G_valid) / float(n_latent_points * decode_times)
    strs = [D.decode_igraph_to_ENAS(g) for g in G_valid]
    r_unique = len(set(strs)) / float(len(strs)) if strs else 0.0
    ga, _ = _as_arrays(train_data, model.max_n)
    r_novel = 1.0 - D.ratio_same_DAG(ga.to_graphs(), G_valid)
    return r_valid, r_unique, r_novel


def save_checkpoint(res_dir, epoch, model, optimizer=None, scheduler=None):
    """train.py:705-712 file names: {model,optimizer,scheduler}_checkpoint<epoch>.pth"""
    os.makedirs(res_dir, exist_ok=True)
    torch.save(model.state_dict(), os.path.join(res_dir, "model_checkpoint%d.pth" % epoch))
    if optimizer is not None:
        torch.save(optimizer.state_dict(), os.path.join(res_dir, "optimizer_checkpoint%d.pth" % epoch))
    if scheduler is not None:
        torch.save(scheduler.state_dict(), os.path.join(res_dir, "scheduler_checkpoint%d.pth" % epoch))


def load_module_state(module, state_name, map_location=None):
    """util.py:38-60: load a saved state dict, dropping keys the module does not have (for a
    model; optimizers and schedulers load their dict whole).  Tensors only: weights_only load."""
    state = torch.load(state_name, map_location=map_location, weights_only=True)
    if isinstance(module, torch.nn.Module):
        own = module.state_dict()
        state = {k: v for k, v in state.items() if k in own}
        module.load_state_dict(state, strict=False)
        if hasattr(module, "_pw"):
            module._pw = None  # weights changed: repack
    else:
        module.load_state_dict(state)
