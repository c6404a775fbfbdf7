"""Row codecs and synthetic DAG generators for the data sets the reference
trains on (formats: `/root/reference/util.py:128-144` ENAS, `:255-275` Bayesian
networks).  `data/asia_200k.txt` and `data/final_structures12.txt` are absent
from the reference checkout, so BASELINE configs 3-5 use the generators below
(specified in SURVEY.md section 8d)."""
import ast
import random

import numpy as np

from .batching import GraphArrays


def parse_row(row):
    if isinstance(row, str):
        row = ast.literal_eval(row.strip())
    if isinstance(row, tuple):
        row = row[0]
    return row


def enas_rows_to_arrays(rows, max_n=None):
    """ENAS rows [[t0],[t1,e],[t2,e,e],...]: vertex i+1 has type t_i + 2, a
    mandatory edge i -> i+1 and optional edges j -> i+1 (j < i); vertex 0 is the
    input (type 0), the last vertex the output (type 1) fed by the last layer."""
    rows = [parse_row(r) for r in rows]
    if max_n is None:
        max_n = max(len(r) for r in rows) + 2
    N = len(rows)
    types = np.zeros((N, max_n), np.int32)
    mask = np.zeros((N, max_n), np.uint64)
    n_g = np.zeros((N,), np.int32)
    for g, row in enumerate(rows):
        k = len(row)
        if k + 2 > max_n:
            raise ValueError("row %d has %d layers, max_n is %d" % (g, k, max_n))
        n_g[g] = k + 2
        for i, node in enumerate(row):
            types[g, i + 1] = node[0] + 2
            m = 1 << i
            for j, e in enumerate(node[1:]):
                if e == 1:
                    m |= 1 << j
            mask[g, i + 1] = m
        types[g, k + 1] = 1
        mask[g, k + 1] = 1 << k
    return GraphArrays(types, mask, n_g, max_n)


def bn_rows_to_arrays(rows, max_n=None):
    """Bayesian-network rows [[v0],[v1,e],...]: vertex i+1 has type v_i + 2 and an
    edge from vertex j+1 for every set flag; a node with no parent hangs off the
    input vertex 0; every loose end feeds the output vertex."""
    rows = [parse_row(r) for r in rows]
    if max_n is None:
        max_n = max(len(r) for r in rows) + 2
    N = len(rows)
    types = np.zeros((N, max_n), np.int32)
    mask = np.zeros((N, max_n), np.uint64)
    n_g = np.zeros((N,), np.int32)
    for g, row in enumerate(rows):
        k = len(row)
        n_g[g] = k + 2
        has_out = [False] * (k + 2)
        for i, node in enumerate(row):
            types[g, i + 1] = node[0] + 2
            m = 0
            if sum(node[1:]) == 0:
                m = 1
                has_out[0] = True
            else:
                for j, e in enumerate(node[1:]):
                    if e == 1:
                        m |= 1 << (j + 1)
                        has_out[j + 1] = True
            mask[g, i + 1] = m
        types[g, k + 1] = 1
        m = 0
        for j in range(k + 1):
            if not has_out[j]:
                m |= 1 << j
        mask[g, k + 1] = m
    return GraphArrays(types, mask, n_g, max_n)


def synth_enas_rows(count, n_layers=6, n_types=6, p_edge=0.30, seed=0, p_rule=None):
    """ENAS-shaped rows: types uniform, optional edges Bernoulli(p_edge)
    (0.30 = the ENAS-6 empirical density).  p_rule(i) overrides the per-layer
    probability (cfg 5 uses min(0.30, 3 / i))."""
    rng = random.Random(seed)
    rows = []
    for _ in range(count):
        row = []
        for i in range(n_layers):
            p = p_rule(i) if p_rule else p_edge
            row.append([rng.randrange(n_types)] + [1 if rng.random() < p else 0 for _ in range(i)])
        rows.append(row)
    return rows


def synth_bn_rows(count, n_vars=8, p_edge=0.25, seed=0):
    """asia-shaped rows: the variables in a random order, edge j -> i (j < i in row
    order) with probability p_edge."""
    rng = random.Random(seed)
    rows = []
    for _ in range(count):
        order = list(range(n_vars))
        rng.shuffle(order)
        rows.append([[t] + [1 if rng.random() < p_edge else 0 for _ in range(i)] for i, t in enumerate(order)])
    return rows


# ---------------------------------------------------------------------------------------------
# Data-set loaders and row <-> graph codecs (reference `util.py:64-296`), igraph-free.
# Rows are parsed with ast.literal_eval (the reference eval()s them).
# ---------------------------------------------------------------------------------------------
import os
from types import SimpleNamespace

from .graph import DAG, IN as _IN


def _one_graph(rows_to_arrays, row):
    """a single row through the array codec above -> (DAG, n_vertices)"""
    ga = rows_to_arrays([row])
    n = int(ga.n_vertices[0])
    return DAG.from_arrays(ga.types[0], ga.pred_mask[0], n), n


def decode_ENAS_to_igraph(row):
    """ENAS row -> (DAG, n_vertices), the interface of `util.py:128-144`; built by `enas_rows_to_arrays`
    (vertex ids are a topological order)."""
    return _one_graph(enas_rows_to_arrays, row)


def decode_BN_to_igraph(row):
    """Bayesian-network row -> (DAG, n_vertices), the interface of `util.py:255-275`; built by `bn_rows_to_arrays`."""
    return _one_graph(bn_rows_to_arrays, row)


def _masks(g):
    """(types, predecessor bitmasks) of a DAG, vertex by vertex"""
    types = [int(t) for t in g.vs["type"]]
    masks = [sum(1 << u for u in preds) for preds in g.get_adjlist(_IN)]
    return types, masks


def flat_ENAS_to_nested(row, n_nodes):
    """'4 0 1 3 1 0 ...' -> [[4], [0, 1], [3, 1, 0], ...] (layer i takes 1 + i numbers; `util.py:147-158`)."""
    flat = [int(x) for x in row.split()] if isinstance(row, str) else list(row)
    layers, pos = [], 0
    for width in range(1, n_nodes + 1):
        if pos >= len(flat):
            break
        layers.append(flat[pos:pos + width])
        pos += width
    return layers


def decode_igraph_to_ENAS(g):
    """graph -> flattened ENAS string: per inner vertex v its type - 2, then one flag per vertex u < v - 1 (the edge
    v - 1 -> v is implied); the inverse of `enas_rows_to_arrays` (`util.py:161-173`)."""
    types, masks = _masks(g)
    out = []
    for v in range(1, len(types) - 1):
        out.append(types[v] - 2)
        out.extend((masks[v] >> u) & 1 for u in range(v - 1))
    return " ".join(str(x) for x in out)


def decode_igraph_to_BN_adj(g):
    """BN graph -> flattened variable-by-variable adjacency string, variables in the order of their ids (vertex
    type - 2), entry [a][b] = edge a -> b (`util.py:278-284`)."""
    types, masks = _masks(g)
    inner = sorted(range(1, len(types) - 1), key=lambda v: types[v])  # stable: equal types keep vertex order
    return " ".join(str((masks[b] >> a) & 1) for a in inner for b in inner)


def adjstr_to_BN(row, n_vars=8):
    """flattened adjacency string -> BN row format: variable i with one flag per earlier variable (`util.py:287-296`)."""
    bits = [int(x) for x in row.split()]
    if len(bits) != n_vars * n_vars:
        raise ValueError("expected %d adjacency entries, found %d" % (n_vars * n_vars, len(bits)))
    return [[i] + [bits[j * n_vars + i] for j in range(i)] for i in range(n_vars)]


def _load_rows(path, with_y, burn_in):
    rows, ys = [], []
    with open(path, "r") as f:
        for i, line in enumerate(f):
            if i < burn_in:
                continue
            line = line.strip()
            if not line:
                continue
            val = ast.literal_eval(line)
            if with_y:
                row, y = val
            else:
                row, y = val, 0.0
            rows.append(row)
            ys.append(float(y))
    return rows, ys


def _split(items, ys, rand_seed):
    """the reference's 90/10 split: random.Random(seed).shuffle of the (g, y) list"""
    order = list(range(len(items)))
    random.Random(rand_seed).shuffle(order)
    cut = int(len(order) * 0.9)
    return order[:cut], order[cut:]


def _finish(rows, ys, decode, to_arrays, fmt, max_n, rand_seed):
    tr, te = _split(rows, ys, rand_seed)
    if fmt == "igraph":
        graphs = [decode(r)[0] for r in rows]
        return [(graphs[i], ys[i]) for i in tr], [(graphs[i], ys[i]) for i in te]
    if fmt == "arrays":
        ga = to_arrays(rows, max_n)
        y = np.asarray(ys, np.float64)
        return (ga.select(tr), y[tr]), (ga.select(te), y[te])
    raise ValueError("fmt must be 'igraph' (DAG objects) or 'arrays' (pre-packed GraphArrays)")


def load_ENAS_graphs(name, n_types=6, fmt="igraph", rand_seed=0, with_y=True, burn_in=1000, data_dir="data"):
    """`util.py:64-93`: reads data/<name>.txt (rows `[[t0],[t1,e],...], y`), skips `burn_in` rows,
    returns (train, test, graph_args) with the reference's shuffle and 90/10 split.
    fmt='igraph': lists of (DAG, y);  fmt='arrays': (GraphArrays, y[]) pairs, the pre-packed form
    the models take directly."""
    rows, ys = _load_rows(os.path.join(data_dir, "%s.txt" % name), with_y, burn_in)
    max_n = max(len(r) for r in rows) + 2
    graph_args = SimpleNamespace(num_vertex_type=n_types + 2, max_n=max_n, START_TYPE=0, END_TYPE=1)
    train, test = _finish(rows, ys, decode_ENAS_to_igraph, enas_rows_to_arrays, fmt, max_n, rand_seed)
    return train, test, graph_args


def load_BN_graphs(name, n_types=8, fmt="igraph", rand_seed=0, with_y=True, data_dir="data"):
    """`util.py:190-223`: Bayesian-network rows, no burn-in, all graphs of one size."""
    rows, ys = _load_rows(os.path.join(data_dir, "%s.txt" % name), with_y, 0)
    sizes = set(len(r) for r in rows)
    if len(sizes) != 1:
        raise ValueError("all Bayesian networks must have the same number of nodes (found %s)" % sorted(sizes))
    max_n = sizes.pop() + 2
    graph_args = SimpleNamespace(num_class=1, num_vertex_type=n_types + 2, max_n=max_n, START_TYPE=0, END_TYPE=1)
    train, test = _finish(rows, ys, decode_BN_to_igraph, bn_rows_to_arrays, fmt, max_n, rand_seed)
    return train, test, graph_args


# ---------------------------------------------------------------------------------------------
# Validity / novelty predicates on decoded graphs (reference `util.py:466-539`)
# ---------------------------------------------------------------------------------------------
def is_same_DAG(g0, g1):
    """same vertex types and predecessor sets, vertex by vertex (no isomorphism test)"""
    return g0.vcount() == g1.vcount() and _masks(g0) == _masks(g1)


def ratio_same_DAG(G0, G1):
    """fraction of G1 that also occurs in G0"""
    def key(g):
        return (tuple(g.vs["type"]), tuple(tuple(sorted(p)) for p in g.get_adjlist(_IN)))
    known = set(key(g) for g in G0)
    return sum(1 for g in G1 if key(g) in known) / max(len(G1), 1)


def is_valid_DAG(g, START_TYPE=0, END_TYPE=1):
    """acyclic, exactly one START and one END vertex, START the only vertex without predecessors and END the only
    one without successors"""
    types, masks = _masks(g)
    has_succ = 0
    for m in masks:
        has_succ |= m
    for v, t in enumerate(types):
        if masks[v] == 0 and t != START_TYPE:
            return False
        if not (has_succ >> v) & 1 and t != END_TYPE:
            return False
    return g.is_dag() and types.count(START_TYPE) == 1 and types.count(END_TYPE) == 1


def is_valid_ENAS(g, START_TYPE=0, END_TYPE=1):
    """a valid DAG whose layers form a chain 0 -> 1 -> ... -> n-2 and whose output has that chain's end as its only
    predecessor"""
    if not is_valid_DAG(g, START_TYPE, END_TYPE):
        return False
    _, masks = _masks(g)
    n = len(masks)
    return all((masks[v + 1] >> v) & 1 for v in range(n - 2)) and bin(masks[n - 1]).count("1") == 1


def is_valid_BN(g, START_TYPE=0, END_TYPE=1, nvt=10):
    """acyclic, one START, one END, and every one of the nvt vertex types exactly once"""
    types = [int(t) for t in g.vs["type"]]
    return (g.is_dag() and types.count(START_TYPE) == 1 and types.count(END_TYPE) == 1 and
            len(set(types)) == nvt and len(types) == nvt)
