"""DAG batching: graphs -> packed index tables on the device.

Replaces the reference's batching, which is a Python list of igraph copies
(`/root/reference/models.py:631-632`), the per-vertex filter / gather / zero-pad
of `models.py:637-668`, and `custom_DataParallel.scatter` (`util.py:557-573`).

Input graphs are duck-typed (python-igraph is not required): any object with
`vcount()`, `vs['type']` and `get_adjlist(mode)` / `predecessors(v)`; vertex ids
must be a topological order (`models.py:677`).  Graphs can also be given as plain
arrays (`GraphArrays`), which is what a pre-packed dataset keeps.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

IGRAPH_IN = 2  # igraph.IN


class GraphArrays(object):
    """A set of DAGs as arrays: types [N, max_n] int32, pred_mask [N, max_n] uint64
    (bit u of pred_mask[g, v] <=> edge u -> v), n_vertices [N] int32."""

    def __init__(self, types, pred_mask, n_vertices, max_n):
        self.types = np.ascontiguousarray(types, dtype=np.int32)
        self.pred_mask = np.ascontiguousarray(pred_mask, dtype=np.uint64)
        self.n_vertices = np.ascontiguousarray(n_vertices, dtype=np.int32)
        self.max_n = int(max_n)
        if self.types.shape != self.pred_mask.shape or self.types.shape[1] != self.max_n:
            raise ValueError("types / pred_mask must both be [N, max_n]")

    def __len__(self):
        return self.types.shape[0]

    def select(self, index):
        index = np.asarray(index)
        return GraphArrays(self.types[index], self.pred_mask[index], self.n_vertices[index], self.max_n)

    def __getitem__(self, s):
        if isinstance(s, slice):
            return GraphArrays(self.types[s], self.pred_mask[s], self.n_vertices[s], self.max_n)
        return self.select(s)

    def to_dense(self):
        """-> types int64 [N, n], adj uint8 [N, n, n], n_g int64 [N] (oracle format)"""
        N, n = self.types.shape
        adj = np.zeros((N, n, n), np.uint8)
        for u in range(n):
            adj[:, u, :] = ((self.pred_mask >> np.uint64(u)) & np.uint64(1)).astype(np.uint8)
        return self.types.astype(np.int64), adj, self.n_vertices.astype(np.int64)

    def to_graphs(self):
        """-> list of graph.DAG (the igraph-like container decode() also returns)"""
        from .graph import DAG
        return [DAG.from_arrays(self.types[i], self.pred_mask[i], self.n_vertices[i]) for i in range(len(self))]

    @staticmethod
    def from_dense(types, adj, n_g):
        types = np.asarray(types)
        adj = np.asarray(adj)
        N, n = types.shape
        mask = np.zeros((N, n), np.uint64)
        for u in range(n):
            mask |= (adj[:, u, :].astype(np.uint64) << np.uint64(u))
        return GraphArrays(types, mask, n_g, n)


def graphs_to_arrays(graphs, max_n):
    """igraph-like objects -> GraphArrays.  Raises ValueError on graphs that do
    not fit (too many vertices, vertex ids not a topological order)."""
    if isinstance(graphs, GraphArrays):
        if graphs.max_n != max_n:
            raise ValueError("GraphArrays packed for max_n=%d, model has max_n=%d" % (graphs.max_n, max_n))
        return graphs
    if max_n > 64:
        raise ValueError("max_n > 64 is not supported (predecessor bitmask is 64 bits)")
    N = len(graphs)
    types = np.zeros((N, max_n), np.int32)
    mask = np.zeros((N, max_n), np.uint64)
    n_g = np.zeros((N,), np.int32)
    for i, g in enumerate(graphs):
        k = g.vcount()
        if k > max_n:
            raise ValueError("graph %d has %d vertices, max_n is %d" % (i, k, max_n))
        if k < 1:
            raise ValueError("graph %d is empty" % i)
        n_g[i] = k
        types[i, :k] = g.vs["type"]
        if hasattr(g, "get_adjlist"):
            adjl = g.get_adjlist(IGRAPH_IN)
        else:
            adjl = [g.predecessors(v) for v in range(k)]
        for v, preds in enumerate(adjl):
            m = 0
            for u in preds:
                if u >= v:
                    raise ValueError("graph %d: edge %d -> %d violates the topological vertex order" % (i, u, v))
                m |= (1 << int(u))
            mask[i, v] = m
    return GraphArrays(types, mask, n_g, max_n)


_COPY_STREAMS = {}


def _copy_stream(device):
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _COPY_STREAMS:
        _COPY_STREAMS[key] = torch.cuda.Stream(device=device)
    return _COPY_STREAMS[key]


def padding_caps(arrays, batch_size, nvt, sigmas=6.0):
    """Caps for shape-stable (padded) batches of `batch_size` graphs drawn from the data set `arrays`
    (include/dvae_hip.h: dvae_batch_export_padded): every such batch then has the same counts, blob layout and launch
    sequence, which is what replaying one captured train step needs (harness.GraphedTrainStep).
    The counts that cost COMPUTE when padded -- partial-state rows PT and EB-table rows NB -- are capped
    statistically: batch sums of per-graph contributions, mean + `sigmas` standard deviations of a random batch
    (clipped to the worst case), so padding adds a few per cent of rows; a batch above a cap falls back to the plain
    export (and the eager step).  E, PL and CH only size index arrays / empty chunks: worst case.
    Returns None when the data set does not qualify (graphs of different sizes)."""
    n = arrays.max_n
    if len(arrays) == 0 or not np.all(arrays.n_vertices == n):
        return None
    m = arrays.pred_mask
    indeg = np.zeros(m.shape, np.int64)
    for u in range(n):
        indeg += ((m >> np.uint64(u)) & np.uint64(1)).astype(np.int64)
    part = np.maximum(indeg - 1, 0).sum(1)                                   # partial rows per graph
    full = ((indeg[:, 1:] >= 1) & ((m[:, 1:] & np.uint64(1)) == 0)).sum(1)   # slot rows the EB table reads
    B = int(batch_size)

    def cap(x, worst):
        v = B * float(x.mean()) + sigmas * (B ** 0.5) * float(x.std()) + 1.0
        return int(min(np.ceil(v), B * worst))

    pairs = n * (n - 1) // 2
    chunk = 64 if n > 16 else 16
    PT = nvt + cap(part, (n - 1) * (n - 2) // 2)
    NB = nvt + cap(part + full, (n - 1) * (n - 2) // 2 + (n - 1))
    caps = _lib.BatchCaps()
    caps.E = B * pairs
    caps.PT = PT
    caps.PL = B * sum(v * (v - 1) // 2 for v in range(1, n))
    caps.NB = NB
    caps.CH = NB + nvt + (B * pairs + chunk - 1) // chunk
    return caps


class DAGBatch(object):
    """A packed batch: host arrays + the device index blob the kernels read.
    `caps` (padding_caps): export shape-stable -- `self.padded` tells whether the batch fitted the caps."""

    def __init__(self, arrays, nvt, start_type, is_bn, device, caps=None):
        lib = _lib.load()
        self.arrays = arrays
        self.B = len(arrays)
        self.max_n = arrays.max_n
        self.nvt = int(nvt)
        self.is_bn = bool(is_bn)
        if self.B == 0:
            raise ValueError("empty batch")
        counts = _lib.BatchCounts()
        handle = C.c_void_p()
        _lib.check(lib.dvae_batch_create(self.B, self.max_n, self.nvt, int(start_type), int(self.is_bn),
                                         _lib.ptr(arrays.types), _lib.ptr(arrays.pred_mask),
                                         _lib.ptr(arrays.n_vertices), C.byref(handle), C.byref(counts)),
                   "dvae_batch_create")
        self.padded = False
        try:
            self.device = torch.device(device)
            on_gpu = self.device.type == "cuda"
            host = None
            if caps is not None:
                # upper bound of the padded blob: the plain blob + the padding of the five capped counts (each array
                # that grows is one of 4 index arrays of E, 3 of PL, 3 of PT, 1 of NB, 2 of CH entries; + alignment)
                pc = _lib.BatchCounts()
                grow = (4 * max(0, caps.E - counts.E) + 3 * max(0, caps.PL - counts.PL) + 3 * max(0, caps.PT - counts.PT) +
                        max(0, caps.NB - counts.NB) + 2 * max(0, caps.CH - counts.CH) + 256)
                host = torch.empty(int(counts.total_ints) + grow, dtype=torch.int32, pin_memory=on_gpu)
                rc = lib.dvae_batch_export_padded(handle, C.byref(caps), host.data_ptr(), host.numel(), C.byref(pc))
                if rc == 0:
                    counts = pc
                    host = host[:int(pc.total_ints)]
                    self.padded = True
                else:
                    host = None  # above a cap (or graphs of different sizes): plain export
            if host is None:
                # pinned staging: the copy below is then a true asynchronous DMA instead of a staged pageable copy
                host = torch.empty(int(counts.total_ints), dtype=torch.int32, pin_memory=on_gpu)
                _lib.check(lib.dvae_batch_export(handle, host.data_ptr(), counts.total_ints), "dvae_batch_export")
        finally:
            lib.dvae_batch_destroy(handle)
        self.counts = counts
        self.blob_host = host
        self.blob = None
        self.ready = None
        self._used_on = set()  # streams the caching allocator has been told about (wait_ready)
        if on_gpu:
            # The copy runs on a copy stream of its own, so a batch packed ahead of time (harness prefetch thread)
            # does not queue behind the compute already enqueued; consumers wait on `ready` (wait_ready()).
            with torch.cuda.device(self.device):
                cs = _copy_stream(self.device)
                with torch.cuda.stream(cs):
                    self.blob = host.to(self.device, non_blocking=True)
                    self.ready = torch.cuda.Event()
                    self.ready.record(cs)

    def wait_ready(self):
        """Called by every consumer before it enqueues kernels that read the blob: orders the current stream behind
        the blob's host-to-device copy (until the copy is known to be done), and tells the caching allocator that the
        blob is in use on this stream.  The blob was allocated under the COPY stream: without `record_stream` its
        memory goes back to that stream's pool the moment the batch object dies, and the next batch's copy may
        overwrite it while kernels enqueued for this batch -- the host runs steps ahead of the GPU -- still read
        their indices from it.  (The library's own side streams are joined into the caller's stream before each
        call returns, so the caller's stream covers them.)"""
        if self.blob is not None and self.blob.is_cuda:
            cur = torch.cuda.current_stream(self.device)
            if cur.cuda_stream not in self._used_on:
                self.blob.record_stream(cur)
                self._used_on.add(cur.cuda_stream)
        if self.ready is not None:
            if self.ready.query():
                self.ready = None
            else:
                torch.cuda.current_stream(self.device).wait_event(self.ready)

    def __len__(self):
        return self.B

    def edges(self):
        return int(self.counts.E)


def shard_bounds(n_items, world_size, rank):
    """Contiguous ceil-split of a batch over ranks, the rule of
    custom_DataParallel.scatter (util.py:566-572)."""
    chunk = -(-n_items // world_size)
    lo = min(rank * chunk, n_items)
    return lo, min(lo + chunk, n_items)


def balanced_shard_bounds(n_vertices, world_size, rank):
    """Contiguous split of a variable-size batch over ranks that evens out the decoder's work instead
    of the graph count: a graph of n vertices costs n(n-1)/2 teacher-forced pairs (models.py:828-857),
    so the cut points are where the running sum of n_g^2 crosses k/world_size of the total.  Same
    contract as shard_bounds (contiguous, ordered, covers the batch exactly once), so the global
    eps rows [lo, hi) still belong to rank `rank`; for equal-size graphs it is the even split."""
    w = np.asarray(n_vertices, np.float64) ** 2
    n_items = int(w.shape[0])
    if n_items == 0:
        return 0, 0
    csum = np.cumsum(w)
    cuts = [0]
    for k in range(1, world_size):
        # first index whose prefix reaches k/world of the total, never before the previous cut
        c = int(np.searchsorted(csum, csum[-1] * k / world_size, side="left")) + 1
        # put the boundary graph on the lighter side
        if c - 1 > cuts[-1] and abs(csum[c - 2] - csum[-1] * k / world_size) <= abs(csum[c - 1] - csum[-1] * k / world_size):
            c -= 1
        cuts.append(min(max(c, cuts[-1]), n_items))
    cuts.append(n_items)
    return cuts[rank], cuts[rank + 1]
