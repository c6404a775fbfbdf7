"""CPU ORACLE for the D-VAE train-step hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  The product package (d-vae_amd/) never does; the product path fails
loudly when its HIP library is missing.

It is a plain-PyTorch (CPU, fp32 or fp64, autograd) restatement of the
reference algorithm in /root/reference/models.py:

  * `encode`            follows DVAE.encode            models.py:720-731
                        (DVAE_BN.encode                models.py:1068-1080)
  * `_propagate`        follows _propagate_to/_from    models.py:634-685
                        (BN variant                    models.py:1030-1066)
  * `loss`              follows DVAE.loss              models.py:806-860
  * `reparameterize`    follows DVAE.reparameterize    models.py:733-740
  * `_edge_score`       follows _get_edge_score        models.py:742-745 (BN 1082-1085)
  * `_graph_state_*`    follow _get_graph_state        models.py:704-718 (BN 1006-1027)
  * `decode`            follows DVAE.decode            models.py:747-804 (the random draws
                        of np.random.choice :766 and torch.rand_like :783 injected as uniforms)

Two evaluation modes that must agree (tests check it):

  mode="dense"     level-batched over the whole batch with [B, n, ...] tensors
                   and masks; padding semantics of SURVEY.md section 3.3.
  mode="refstyle"  per-vertex / per-graph Python loops that mirror the
                   reference op for op (one GRUCell call per _update_v call,
                   list-of-[1,hs] states, torch.cat of padded predecessor
                   stacks).  This is the "port" timed as cpu_baseline.

Parity pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4).  The oracle is pinned against outputs of the reference
itself, generated in the build container by oracle/gen_golden.py (imports
/root/reference/models.py with oracle/igraph_standin on sys.path) and committed
under tests/golden/.  tests/test_oracle_golden.py re-checks them.

Inputs are plain arrays, not igraph objects:
  types [B, n] int64  (vertex type id; entries at v >= n_g are ignored)
  adj   [B, n, n] bool/0-1, adj[b, u, v] = 1 iff edge u -> v (u < v)
  n_g   [B] int64     (vertex count of each graph, <= n)
  sd    dict name -> tensor in the reference state_dict layout
"""
from collections import namedtuple

import torch
import torch.nn.functional as F

Cfg = namedtuple("Cfg", "max_n nvt START_TYPE END_TYPE hs nz bidir vid is_bn")


def make_cfg(max_n, nvt, START_TYPE=0, END_TYPE=1, hs=501, nz=56, bidir=False, vid=True, is_bn=False):
    return Cfg(max_n, nvt, START_TYPE, END_TYPE, hs, nz, bool(bidir), bool(vid) and not is_bn, bool(is_bn))


# ----------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------
def _gru(sd, name, x_onehot, h):
    """torch.nn.GRUCell semantics (gate order r, z, n) used at models.py:670."""
    w_ih, w_hh = sd[name + ".weight_ih"], sd[name + ".weight_hh"]
    b_ih, b_hh = sd[name + ".bias_ih"], sd[name + ".bias_hh"]
    gi = x_onehot @ w_ih.t() + b_ih
    gh = h @ w_hh.t() + b_hh
    H = h.shape[-1]
    r = torch.sigmoid(gi[..., :H] + gh[..., :H])
    z = torch.sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
    n = torch.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
    return (1 - z) * n + z * h


def _linear(sd, name, x):
    w = sd[name + ".weight"]
    b = sd.get(name + ".bias")
    y = x @ w.t()
    return y if b is None else y + b


def _gated(sd, direction, s):
    """_gated, models.py:628-629: sigmoid(gate(s)) * mapper(s); mapper has no bias."""
    g = torch.sigmoid(_linear(sd, "gate_%s.0" % direction, s))
    m = _linear(sd, "mapper_%s.0" % direction, s)
    return g * m


def _one_hot(idx, length, dtype):
    return F.one_hot(idx, length).to(dtype)


def _msg_input(cfg, h, vidx, types_v, dtype):
    """What the gate/mapper see for a source vertex: [h ; onehot(vid)] for DVAE
    with vid (models.py:655-658), h alone without vid, onehot(type) for DVAE_BN
    (models.py:1044-1050)."""
    if cfg.is_bn:
        return _one_hot(types_v, cfg.nvt, dtype)
    if cfg.vid:
        vid = torch.zeros(h.shape[0], cfg.max_n, dtype=dtype)
        vid[:, vidx] = 1
        return torch.cat([h, vid], 1)
    return h


def reparameterize(mu, logvar, eps, training=True, eps_scale=0.01):
    """models.py:733-740 with the normal draw `eps` injected by the caller."""
    if training:
        std = (0.5 * logvar).exp()
        return eps * eps_scale * std + mu
    return mu


# ----------------------------------------------------------------------------
# dense (level-batched) mode
# ----------------------------------------------------------------------------
def _chain_dense(cfg, sd, cell, direction, types, adjf, n_g, h_first=None):
    """One topologically ordered pass (models.py:675-685).  direction 'forward'
    walks v = 0..n-1 and sums over predecessors; 'backward' walks v = n-1..0 and
    sums over successors.  Returns list H[v] of [B, hs] (rows with v >= n_g hold
    unused values) and list of messages G[v]."""
    B, n = types.shape
    dtype = adjf.dtype
    H = [None] * n
    G = [None] * n
    order = range(n) if direction == "forward" else range(n - 1, -1, -1)
    for v in order:
        x = _one_hot(types[:, v], cfg.nvt, dtype)
        agg = torch.zeros(B, cfg.hs, dtype=dtype)
        if direction == "forward":
            for u in range(v):
                agg = agg + adjf[:, u, v, None] * G[u]
        else:
            for w in range(v + 1, n):
                agg = agg + adjf[:, v, w, None] * G[w]
        if h_first is not None and v == order[0]:
            agg = h_first
        H[v] = _gru(sd, cell, x, agg)
        G[v] = _gated(sd, direction, _msg_input(cfg, H[v], v, types[:, v], dtype))
    return H, G


def _pick(Hlist, idx):
    """rows Hlist[idx[b]][b]"""
    st = torch.stack(Hlist, 1)  # [B, n, hs]
    return st[torch.arange(st.shape[0]), idx]


def encode_dense(cfg, sd, types, adj, n_g):
    dtype = sd["fc1.weight"].dtype
    adjf = adj.to(dtype)
    types = types.clone()
    B, n = types.shape
    ar = torch.arange(n)[None, :]
    types[ar >= n_g[:, None]] = 0  # ignored slots
    Hf, _ = _chain_dense(cfg, sd, "grue_forward", "forward", types, adjf, n_g)
    Hb = None
    if cfg.bidir:
        Hb, _ = _chain_dense(cfg, sd, "grue_backward", "backward", types, adjf, n_g)
    if cfg.is_bn:
        # models.py:1006-1027 with start=1, end_offset=1 (called from :1077)
        mask = ((ar >= 1) & (ar < (n_g[:, None] - 1))).to(dtype)[:, :, None]
        Hg = (torch.stack(Hf, 1) * mask).sum(1)
        if cfg.bidir:
            Hgb = (torch.stack(Hb, 1) * mask).sum(1)
            Hg = _linear(sd, "hg_unify.0", torch.cat([Hg, Hgb], 1))
    else:
        # models.py:704-718
        Hg = _pick(Hf, n_g - 1)
        if cfg.bidir:
            Hg = _linear(sd, "hg_unify.0", torch.cat([Hg, Hb[0]], 1))
    return _linear(sd, "fc1", Hg), _linear(sd, "fc2", Hg)


def _edge_score(cfg, sd, Hvi, Hc, H0):
    if cfg.is_bn:
        x = torch.cat([Hvi, Hc, H0], 1)
    else:
        x = torch.cat([Hvi, Hc], 1)
    h = torch.relu(_linear(sd, "add_edge.0", x))
    return torch.sigmoid(_linear(sd, "add_edge.2", h))


def loss_dense(cfg, sd, mu, logvar, types, adj, n_g, eps, beta=0.005, training=True):
    dtype = mu.dtype
    adjf = adj.to(dtype)
    types = types.clone()
    B, n = types.shape
    ar = torch.arange(n)[None, :]
    types[ar >= n_g[:, None]] = 0
    z = reparameterize(mu, logvar, eps, training)
    H0 = torch.tanh(_linear(sd, "fc3", z))
    Hd = [None] * n
    Gd = [None] * n
    x0 = _one_hot(types[:, 0], cfg.nvt, dtype)
    Hd[0] = _gru(sd, "grud", x0, H0)
    Gd[0] = _gated(sd, "forward", _msg_input(cfg, Hd[0], 0, types[:, 0], dtype))
    res = torch.zeros((), dtype=dtype)
    bidx = torch.arange(B)
    for v in range(1, n):
        exists = (v < n_g)
        # vertex-type log-likelihood, models.py:820-826
        tgt = torch.where(exists, types[:, v], torch.full_like(types[:, v], cfg.START_TYPE))
        if cfg.is_bn:
            # graph state while decoding = sum of all states so far (models.py:1006-1027, start=0)
            upto = torch.clamp(n_g, max=v)
            mask = (ar[:, :v] < upto[:, None]).to(dtype)[:, :, None]
            Hg = (torch.stack(Hd[:v], 1) * mask).sum(1)
        else:
            Hg = _pick(Hd[:v], torch.clamp(n_g, max=v) - 1)
        logits = _linear(sd, "add_vertex.2", torch.relu(_linear(sd, "add_vertex.0", Hg)))
        res = res + F.log_softmax(logits, 1)[bidx, tgt].sum()
        # edges, models.py:832-856
        x = _one_hot(types[:, v], cfg.nvt, dtype)
        S = torch.zeros(B, cfg.hs, dtype=dtype)
        Hcur = _gru(sd, "grud", x, S)
        ex = exists.to(dtype)[:, None]
        scores = [None] * v
        for vi in range(v - 1, -1, -1):
            Hvi = Hd[vi] * (vi < n_g).to(dtype)[:, None]
            scores[vi] = _edge_score(cfg, sd, Hvi, Hcur * ex, H0)
            S = S + adjf[:, vi, v, None] * Gd[vi]
            Hcur = _gru(sd, "grud", x, S)
        Hd[v] = Hcur
        Gd[v] = _gated(sd, "forward", _msg_input(cfg, Hd[v], v, types[:, v], dtype))
        scores = torch.cat(scores, 1)
        truth = adjf[:, :v, v] * ex
        res = res - F.binary_cross_entropy(scores, truth, reduction="sum")
    res = -res
    kld = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())
    return res + beta * kld, res, kld


# ----------------------------------------------------------------------------
# reference-style (per-vertex, per-graph) mode -- the timed CPU "port"
# ----------------------------------------------------------------------------
class _G(object):
    """Tiny adjacency container playing the role of the igraph object that the
    reference hangs its per-vertex state on (models.py:671-672)."""
    __slots__ = ("types", "pred", "succ", "Hf", "Hb")

    def __init__(self):
        self.types, self.pred, self.succ, self.Hf, self.Hb = [], [], [], [], []

    def vcount(self):
        return len(self.types)

    def add_vertex(self, t):
        self.types.append(int(t))
        self.pred.append([])
        self.succ.append([])
        self.Hf.append(None)
        self.Hb.append(None)

    def add_edge(self, u, v):
        self.pred[v].append(u)
        self.pred[v].sort()
        self.succ[u].append(v)
        self.succ[u].sort()


def graphs_from_arrays(types, adj, n_g):
    out = []
    for b in range(types.shape[0]):
        g = _G()
        k = int(n_g[b])
        for v in range(k):
            g.add_vertex(int(types[b, v]))
        nz = torch.nonzero(adj[b, :k, :k])
        for u, v in nz.tolist():
            g.add_edge(u, v)
        out.append(g)
    return out


def _propagate_to(cfg, sd, G, v, cell, H=None, reverse=False):
    """models.py:634-673 / 1030-1066."""
    dtype = sd["fc1.weight"].dtype
    G = [g for g in G if g.vcount() > v]
    if len(G) == 0:
        return None
    X = _one_hot(torch.tensor([g.types[v] for g in G]), cfg.nvt, dtype)
    direction = "backward" if reverse else "forward"
    if H is None:
        nbrs = [(g.succ[v] if reverse else g.pred[v]) for g in G]
        max_n_pred = max(len(x) for x in nbrs)
        if max_n_pred == 0:
            H = torch.zeros(len(G), cfg.hs, dtype=dtype)
        else:
            width = cfg.nvt if cfg.is_bn else (cfg.hs + cfg.max_n if cfg.vid else cfg.hs)
            rows = []
            for g, nb in zip(G, nbrs):
                per = []
                for u in nb:
                    if cfg.is_bn:
                        per.append(_one_hot(torch.tensor([g.types[u]]), cfg.nvt, dtype))
                    else:
                        hu = (g.Hb if reverse else g.Hf)[u]
                        if cfg.vid:
                            hu = torch.cat([hu, _one_hot(torch.tensor([u]), cfg.max_n, dtype)], 1)
                        per.append(hu)
                per.append(torch.zeros(max_n_pred - len(nb), width, dtype=dtype))
                rows.append(torch.cat(per, 0).unsqueeze(0))
            Hp = torch.cat(rows, 0)
            H = _gated(sd, direction, Hp).sum(1)
    Hv = _gru(sd, cell, X, H)
    for i, g in enumerate(G):
        if reverse:
            g.Hb[v] = Hv[i:i + 1]
        else:
            g.Hf[v] = Hv[i:i + 1]
    return Hv


def encode_refstyle(cfg, sd, types, adj, n_g):
    dtype = sd["fc1.weight"].dtype
    G = graphs_from_arrays(types, adj, n_g)
    B = len(G)
    _propagate_to(cfg, sd, G, 0, "grue_forward", torch.zeros(B, cfg.hs, dtype=dtype))
    for v in range(1, cfg.max_n):
        _propagate_to(cfg, sd, G, v, "grue_forward")
    if cfg.bidir:
        v0 = cfg.max_n - 1
        full = [g for g in G if g.vcount() > v0]
        _propagate_to(cfg, sd, G, v0, "grue_backward", torch.zeros(len(full), cfg.hs, dtype=dtype), reverse=True)
        for v in range(v0 - 1, -1, -1):
            _propagate_to(cfg, sd, G, v, "grue_backward", reverse=True)
    if cfg.is_bn:
        Hg = torch.cat([torch.cat(g.Hf[1:g.vcount() - 1], 0).sum(0, keepdim=True) for g in G], 0)
        if cfg.bidir:
            Hgb = torch.cat([torch.cat(g.Hb[1:g.vcount() - 1], 0).sum(0, keepdim=True) for g in G], 0)
            Hg = _linear(sd, "hg_unify.0", torch.cat([Hg, Hgb], 1))
    else:
        Hg = torch.cat([g.Hf[g.vcount() - 1] for g in G], 0)
        if cfg.bidir:
            Hg = _linear(sd, "hg_unify.0", torch.cat([Hg, torch.cat([g.Hb[0] for g in G], 0)], 1))
    return _linear(sd, "fc1", Hg), _linear(sd, "fc2", Hg)


def loss_refstyle(cfg, sd, mu, logvar, types, adj, n_g, eps, beta=0.005, training=True):
    dtype = mu.dtype
    Gt = graphs_from_arrays(types, adj, n_g)
    B = len(Gt)
    z = reparameterize(mu, logvar, eps, training)
    H0 = torch.tanh(_linear(sd, "fc3", z))
    G = [_G() for _ in range(B)]
    for g in G:
        g.add_vertex(cfg.START_TYPE)
    _propagate_to(cfg, sd, G, 0, "grud", H0)

    def vstate(v):
        return torch.cat([g.Hf[v] if v < g.vcount() else torch.zeros(1, cfg.hs, dtype=dtype) for g in G], 0)

    res = 0
    for v in range(1, cfg.max_n):
        tt = [gt.types[v] if v < gt.vcount() else cfg.START_TYPE for gt in Gt]
        if cfg.is_bn:
            Hg = torch.cat([torch.cat(g.Hf[:g.vcount()], 0).sum(0, keepdim=True) for g in G], 0)
        else:
            Hg = torch.cat([g.Hf[g.vcount() - 1] for g in G], 0)
        logits = _linear(sd, "add_vertex.2", torch.relu(_linear(sd, "add_vertex.0", Hg)))
        res = res + F.log_softmax(logits, 1)[torch.arange(B), torch.tensor(tt)].sum()
        for i, g in enumerate(G):
            if tt[i] != cfg.START_TYPE:
                g.add_vertex(tt[i])
        _propagate_to(cfg, sd, G, v, "grud")
        te = [gt.pred[v] if v < gt.vcount() else [] for gt in Gt]
        scores = []
        for vi in range(v - 1, -1, -1):
            scores.append(_edge_score(cfg, sd, vstate(vi), vstate(v), H0))
            for i, g in enumerate(G):
                if vi in te[i]:
                    g.add_edge(vi, v)
            _propagate_to(cfg, sd, G, v, "grud")
        scores = torch.cat(scores[::-1], 1)
        truth = torch.zeros_like(scores)
        for i, x in enumerate(te):
            for u in x:
                truth[i, u] = 1.0
        res = res - F.binary_cross_entropy(scores, truth, reduction="sum")
    res = -res
    kld = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())
    return res + beta * kld, res, kld


# ----------------------------------------------------------------------------
# generation (models.py:747-804)
# ----------------------------------------------------------------------------
def choice_from_uniform(probs, u):
    """np.random.choice(range(len(p)), p=p) for ONE uniform draw u in [0, 1): numpy's legacy
    generator builds the float64 CDF, normalises it by its last entry and takes
    cdf.searchsorted(u, side='right')."""
    import numpy as np
    cdf = np.asarray(probs, dtype=np.float64).cumsum()
    cdf /= cdf[-1]
    return int(min(cdf.searchsorted(float(u), side="right"), len(cdf) - 1))


def decode(cfg, sd, z, stochastic=True, u_type=None, u_edge=None):
    """DVAE.decode / DVAE_BN's inherited decode, graph by graph (decode has no cross-graph
    coupling).  stochastic: vertex type = inverse-CDF sample with u_type[b, idx] (what
    np.random.choice does with its uniform), edge vi -> idx iff u_edge[b, idx, vi] < score
    (torch.rand_like(score) < score).  Returns types [B, n] int64, adj [B, n, n] uint8, n_g [B]."""
    import numpy as np
    B, n, dtype = z.shape[0], cfg.max_n, z.dtype
    H0 = torch.tanh(_linear(sd, "fc3", z))
    types_out = np.zeros((B, n), np.int64)
    adj_out = np.zeros((B, n, n), np.uint8)
    ng_out = np.zeros((B,), np.int64)

    def agg(H, types, preds):
        if not preds:
            return torch.zeros(1, cfg.hs, dtype=dtype)
        tot = 0
        for u in preds:
            tot = tot + _gated(sd, "forward", _msg_input(cfg, H[u], u, torch.tensor([types[u]]), dtype))
        return tot

    for b in range(B):
        h0 = H0[b:b + 1]
        types = [cfg.START_TYPE]
        preds = [[]]
        H = [_gru(sd, "grud", _one_hot(torch.tensor([cfg.START_TYPE]), cfg.nvt, dtype), h0)]
        for idx in range(1, n):
            if idx == n - 1:
                t = cfg.END_TYPE
            else:
                Hg = torch.cat(H, 0).sum(0, keepdim=True) if cfg.is_bn else H[-1]
                scores = _linear(sd, "add_vertex.2", torch.relu(_linear(sd, "add_vertex.0", Hg)))
                if stochastic:
                    probs = F.softmax(scores, 1)[0].detach().numpy()
                    t = choice_from_uniform(probs, u_type[b, idx])
                else:
                    t = int(torch.argmax(scores, 1).item())
            types.append(t)
            preds.append([])
            x = _one_hot(torch.tensor([t]), cfg.nvt, dtype)
            Hcur = _gru(sd, "grud", x, agg(H, types, preds[idx]))
            for vi in range(idx - 1, -1, -1):
                if t == cfg.END_TYPE:
                    # connect every loose end (out-degree 0) to the end vertex, then stop (:786-791)
                    has_out = set(u for p in preds for u in p)
                    preds[idx] = [u for u in range(idx) if u not in has_out]
                    Hcur = _gru(sd, "grud", x, agg(H, types, preds[idx]))
                    break
                score = _edge_score(cfg, sd, H[vi], Hcur, h0)[0, 0].item()
                take = (float(u_edge[b, idx, vi]) < score) if stochastic else (score > 0.5)
                if take:
                    preds[idx].append(vi)
                Hcur = _gru(sd, "grud", x, agg(H, types, preds[idx]))
            H.append(Hcur)
            if t == cfg.END_TYPE:
                break
        k = len(types)
        ng_out[b] = k
        types_out[b, :k] = types
        for v, pl in enumerate(preds):
            for u in pl:
                adj_out[b, u, v] = 1
    return types_out, adj_out, ng_out


# ----------------------------------------------------------------------------
# public entry points
# ----------------------------------------------------------------------------
def encode(cfg, sd, types, adj, n_g, mode="dense"):
    fn = encode_dense if mode == "dense" else encode_refstyle
    return fn(cfg, sd, types, adj, n_g)


def loss(cfg, sd, mu, logvar, types, adj, n_g, eps, beta=0.005, training=True, mode="dense"):
    fn = loss_dense if mode == "dense" else loss_refstyle
    return fn(cfg, sd, mu, logvar, types, adj, n_g, eps, beta, training)


def train_step_grads(cfg, sd, types, adj, n_g, eps, beta=0.005, mode="dense"):
    """encode -> loss -> backward.  Returns (loss, recon, kld, mu, logvar, grads)
    where grads maps parameter name -> gradient tensor, or None for parameters
    the step never touches ('dead parameters')."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    mu, logvar = encode(cfg, leaf, types, adj, n_g, mode)
    total, res, kld = loss(cfg, leaf, mu, logvar, types, adj, n_g, eps, beta, True, mode)
    names = list(leaf.keys())
    gs = torch.autograd.grad(total, [leaf[k] for k in names], allow_unused=True)
    grads = {k: g for k, g in zip(names, gs)}
    return total.detach(), res.detach(), kld.detach(), mu.detach(), logvar.detach(), grads
