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
