"""Graph container, row codecs, loaders and validity predicates (reference util.py:64-296,
466-539) -- CPU only.  The reference's own round-trip example rows (util.py:175-187) are the
fixtures; the packer input built from a DAG must equal the one built straight from the row."""
import os
import random

import numpy as np
import pytest

from _dvae_pkg import dvae_amd
from dvae_amd import data as D
from dvae_amd.graph import DAG, IN, OUT

# the two rows the reference keeps as its (commented-out) format test, util.py:177-178
ROW8 = "[[4], [0, 1], [3, 1, 0], [3, 0, 1, 1], [1, 1, 1, 1, 1], [2, 1, 1, 0, 1, 1], [5, 1, 1, 1, 1, 1, 0], [2, 0, 0, 1, 0, 0, 1, 0]]"
ROW12 = ("[[2], [2, 0], [4, 0, 0], [0, 1, 0, 0], [2, 1, 0, 0, 1], [3, 1, 0, 0, 0, 0], [5, 0, 0, 0, 0, 1, 0], "
         "[4, 0, 0, 0, 0, 0, 0, 0], [4, 1, 0, 0, 1, 0, 0, 0, 0], [3, 0, 1, 1, 0, 0, 1, 0, 0, 0], "
         "[5, 0, 0, 0, 1, 0, 1, 1, 0, 0, 1], [5, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0]]")
# the example of util.py:288-289
BN_ADJ = "0 1 0 0 0 1 0 0 0 0 0 0 1 1 1 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 1 0 0 0 0 0 0 1 1 0 0 0 0 0 0 0 1 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0"
BN_ROW = [[0], [1, 1], [2, 0, 0], [3, 0, 0, 0], [4, 0, 1, 0, 0], [5, 1, 1, 0, 0, 0], [6, 0, 1, 0, 0, 1, 0], [7, 0, 0, 0, 1, 1, 1, 0]]


def test_dag_container_surface():
    g = DAG(directed=True)
    g.add_vertex(type=0)
    g.add_vertices(3)
    g.vs[1]["type"] = 3
    g.vs[2]["type"] = 2
    g.vs[3]["type"] = 1
    g.add_edge(0, 1)
    g.add_edges([(0, 2), (1, 3), (2, 3)])
    assert g.vcount() == 4 and g.ecount() == 4
    assert g.vs["type"] == [0, 3, 2, 1]
    assert g.predecessors(3) == [1, 2] and g.successors(0) == [1, 2]
    assert g.get_adjlist(IN) == [[], [0], [0], [1, 2]] and g.get_adjlist(OUT) == [[1, 2], [3], [3], []]
    assert g.neighbors(1, "in") == [0] and g.neighbors(1, "all") == [0, 3]
    assert [v.index for v in g.vs.select(_outdegree_eq=0)] == [3]
    assert g.vs[3].indegree() == 2 and g.vs[0].outdegree() == 2
    assert g.is_dag() and g.topological_sorting() == [0, 1, 2, 3]
    assert g.are_connected(2, 3) and not g.are_connected(3, 2)
    assert g.get_adjacency().data[0] == [0, 1, 1, 0]
    h = g.copy()
    h.add_edge(3, 0)
    assert not h.is_dag() and g.is_dag() and g.ecount() == 4
    g.vs["H"] = [1, 2, 3, 4]
    del g.vs["H"]
    with pytest.raises(KeyError):
        g.vs["H"]


@pytest.mark.parametrize("row,n", [(ROW8, 8), (ROW12, 12)])
def test_enas_round_trip(row, n):
    g, nv = D.decode_ENAS_to_igraph(row)
    assert nv == n + 2 and g.vcount() == n + 2
    assert g.vs[0]["type"] == 0 and g.vs[n + 1]["type"] == 1
    s = D.decode_igraph_to_ENAS(g)
    nested = D.flat_ENAS_to_nested(s, n)
    assert nested == D.parse_row(row)
    assert D.is_valid_DAG(g) and D.is_valid_ENAS(g)
    # packer input from the graph == packer input straight from the row
    a = dvae_amd.graphs_to_arrays([g], n + 2)
    b = D.enas_rows_to_arrays([row], n + 2)
    assert np.array_equal(a.types, b.types) and np.array_equal(a.pred_mask, b.pred_mask)
    assert dvae_amd.data.is_same_DAG(g, a.to_graphs()[0])


def test_bn_codecs():
    assert D.adjstr_to_BN(BN_ADJ) == BN_ROW
    g, nv = D.decode_BN_to_igraph(BN_ROW)
    assert nv == 10 and D.is_valid_BN(g, nvt=10)
    assert D.decode_igraph_to_BN_adj(g) == BN_ADJ
    # node 0 has no parent -> hangs off the input vertex; loose ends feed the output vertex
    assert g.predecessors(1) == [0]
    assert all(g.vs[v].outdegree() > 0 for v in range(9))
    a = dvae_amd.graphs_to_arrays([g], 10)
    b = D.bn_rows_to_arrays([BN_ROW], 10)
    assert np.array_equal(a.types, b.types) and np.array_equal(a.pred_mask, b.pred_mask)


def test_validity_predicates():
    g, _ = D.decode_ENAS_to_igraph(ROW8)
    bad = g.copy()
    bad.add_vertex(type=5)  # dangling vertex: no input, no output
    assert not D.is_valid_DAG(bad)
    skip = DAG()
    skip.add_vertices(4)
    skip.vs["type"] = [0, 2, 3, 1]
    skip.add_edges([(0, 1), (0, 2), (1, 3), (2, 3)])
    assert D.is_valid_DAG(skip) and not D.is_valid_ENAS(skip)  # 1 -> 2 is missing
    assert D.ratio_same_DAG([g], [g.copy(), skip]) == 0.5


def test_loaders_split_and_burn_in(tmp_path):
    rows = D.synth_enas_rows(60, 6, 6, 0.3, seed=3)
    with open(tmp_path / "enas.txt", "w") as f:
        for i, r in enumerate(rows):
            f.write("%s, %r\n" % (r, 0.5 + i))
    train, test, ga = D.load_ENAS_graphs("enas", n_types=6, burn_in=10, data_dir=str(tmp_path))
    assert ga.max_n == 8 and ga.num_vertex_type == 8 and ga.START_TYPE == 0 and ga.END_TYPE == 1
    assert len(train) == 45 and len(test) == 5
    # the reference's rule: random.Random(0).shuffle of the post-burn-in list, first 90 % train
    ref = list(range(50))
    random.Random(0).shuffle(ref)
    ys = [y for _, y in train] + [y for _, y in test]
    assert ys == [0.5 + 10 + i for i in ref]
    g0, y0 = train[0]
    assert D.is_same_DAG(g0, D.decode_ENAS_to_igraph(rows[10 + ref[0]])[0])
    (tr_a, tr_y), (te_a, te_y), _ = D.load_ENAS_graphs("enas", fmt="arrays", burn_in=10, data_dir=str(tmp_path))
    assert len(tr_a) == 45 and np.allclose(tr_y, [y for _, y in train])
    assert all(D.is_same_DAG(a, b[0]) for a, b in zip(tr_a.to_graphs(), train))
    # Bayesian networks: no burn-in, one size
    bn = D.synth_bn_rows(20, 8, 0.25, seed=1)
    with open(tmp_path / "bn.txt", "w") as f:
        for r in bn:
            f.write("%s, %r\n" % (r, -1.0))
    train, test, ga = D.load_BN_graphs("bn", n_types=8, data_dir=str(tmp_path))
    assert ga.max_n == 10 and ga.num_vertex_type == 10 and len(train) == 18 and len(test) == 2
    assert all(D.is_valid_BN(g, nvt=10) for g, _ in train)


@pytest.mark.skipif(not os.path.exists("/root/reference/data/final_structures6.txt"),
                    reason="the reference's data file only exists in the build container")
def test_loader_reads_the_reference_file():
    train, test, ga = D.load_ENAS_graphs("final_structures6", fmt="arrays", data_dir="/root/reference/data")
    n = len(train[0]) + len(test[0])
    assert ga.max_n == 8 and ga.num_vertex_type == 8 and n == 19020  # 20020 rows - burn_in 1000
    assert len(train[0]) == int(n * 0.9)
    assert (train[0].n_vertices == 8).all()
