"""Row-format -> graph helpers for the oracle side (TEST INFRASTRUCTURE ONLY).

Restates the two text codecs of the reference so that golden vectors can be
generated from the same rows the reference trains on:

  decode_ENAS_to_igraph   /root/reference/util.py:128-144
  decode_BN_to_igraph     /root/reference/util.py:255-275

`igraph` here is whatever module of that name is importable: the real
python-igraph if present, otherwise oracle/igraph_standin/igraph.py (the
caller puts it on sys.path).
"""
import ast

import numpy as np


def _igraph():
    import igraph  # noqa: resolved through sys.path by the caller
    return igraph


def parse_row(row):
    if isinstance(row, str):
        row = ast.literal_eval(row.strip())
    if isinstance(row, tuple):  # "(row, y)" lines of data/*.txt
        row = row[0]
    return row


def decode_ENAS_to_igraph(row):
    igraph = _igraph()
    row = parse_row(row)
    n = len(row)
    g = igraph.Graph(directed=True)
    g.add_vertices(n + 2)
    g.vs[0]["type"] = 0
    for i, node in enumerate(row):
        g.vs[i + 1]["type"] = node[0] + 2
        g.add_edge(i, i + 1)
        for j, e in enumerate(node[1:]):
            if e == 1:
                g.add_edge(j, i + 1)
    g.vs[n + 1]["type"] = 1
    g.add_edge(n, n + 1)
    return g, n + 2


def decode_BN_to_igraph(row):
    igraph = _igraph()
    row = parse_row(row)
    n = len(row)
    g = igraph.Graph(directed=True)
    g.add_vertices(n + 2)
    g.vs[0]["type"] = 0
    for i, node in enumerate(row):
        g.vs[i + 1]["type"] = node[0] + 2
        if sum(node[1:]) == 0:
            g.add_edge(0, i + 1)
        else:
            for j, e in enumerate(node[1:]):
                if e == 1:
                    g.add_edge(j + 1, i + 1)
    g.vs[n + 1]["type"] = 1
    for j in [v.index for v in g.vs.select(_outdegree_eq=0) if v.index != n + 1]:
        g.add_edge(j, n + 1)
    return g, n + 2


def igraphs_to_arrays(graphs, max_n):
    """-> types [B, max_n] int64, adj [B, max_n, max_n] uint8, n_g [B] int64"""
    igraph = _igraph()
    B = len(graphs)
    types = np.zeros((B, max_n), np.int64)
    adj = np.zeros((B, max_n, max_n), np.uint8)
    n_g = np.zeros((B,), np.int64)
    for b, g in enumerate(graphs):
        k = g.vcount()
        n_g[b] = k
        types[b, :k] = g.vs["type"]
        for v, preds in enumerate(g.get_adjlist(igraph.IN)):
            for u in preds:
                adj[b, u, v] = 1
    return types, adj, n_g


def arrays_to_igraphs(types, adj, n_g):
    igraph = _igraph()
    out = []
    for b in range(len(n_g)):
        k = int(n_g[b])
        g = igraph.Graph(directed=True)
        g.add_vertices(k)
        for v in range(k):
            g.vs[v]["type"] = int(types[b][v])
        for u in range(k):
            for v in range(k):
                if adj[b][u][v]:
                    g.add_edge(u, v)
        out.append(g)
    return out
