"""Host-side logic without a GPU: the C-ABI library loads and exports every
declared symbol, the DAG packer's tables reproduce the oracle's loss when
interpreted on the CPU, malformed graphs are rejected, module layout matches the
reference's state_dict."""
import os
import re

import numpy as np
import pytest
import torch

from _dvae_pkg import dvae_amd
from common import golden_files, load_case, cfg_of, make_model, state_dict_of
import blob_interp as BI
import dvae_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = golden_files("*tiny*")


def test_library_exports_every_declared_symbol():
    lib = dvae_amd._lib.load()
    header = open(os.path.join(ROOT, "include", "dvae_hip.h")).read()
    declared = set(re.findall(r"\b(dvae_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(dvae_amd._lib.SIGNATURES.keys())
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.dvae_version() >= 100


def test_compute_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    d, cls, ctor = load_case(TINY[0])
    model = make_model(cls, ctor)
    ga = dvae_amd.GraphArrays.from_dense(d["types"], d["adj"], d["n_g"])
    with pytest.raises(dvae_amd._lib.DvaeHipError):
        model.encode(ga)


@pytest.mark.parametrize("path", TINY, ids=[os.path.basename(p)[:-4] for p in TINY])
def test_packer_tables_reproduce_reference_loss(path):
    d, cls, ctor = load_case(path)
    cfg = cfg_of(cls, ctor)
    sd = state_dict_of(d, torch.float64)
    ga = dvae_amd.GraphArrays.from_dense(d["types"], d["adj"], d["n_g"])
    b = dvae_amd.DAGBatch(ga, ctor["nvt"], 0, cls == "DVAE_BN", "cpu")
    mu, lv, (L, R, K) = BI.forward(cfg, sd, b, torch.from_numpy(d["eps"]).double())
    assert abs(L.item() - d["loss3"][0]) < 1e-6 * d["loss3"][0]
    assert (mu.float() - torch.from_numpy(d["mu"])).abs().max() < 1e-6


def test_packer_counts_and_dedup():
    d, cls, ctor = load_case(golden_files("dvae_cfg1_uni")[0])
    ga = dvae_amd.GraphArrays.from_dense(d["types"], d["adj"], d["n_g"])
    b = dvae_amd.DAGBatch(ga, 8, 0, False, "cpu")
    c = b.counts
    B, n = 32, 8
    E = int(d["adj"].sum())
    assert (c.B, c.n, c.nB, c.E, c.NP) == (B, n, B * n, E, B * n * (n - 1) // 2)
    # distinct partial rows: one per type + (indeg - 1) per vertex with indeg >= 2
    indeg = d["adj"].astype(np.int64).sum(1)
    assert c.PT == 8 + int(np.maximum(indeg - 1, 0).sum())
    assert c.R == B * (n - 1) and c.NCE == B * (n - 1)
    t = BI.unpack_blob(b)
    # every pair points at valid table rows; targets equal the adjacency
    assert t["pair_a"].max() <= c.NA and t["pair_h"].max() <= c.NB
    assert int(t["pair_t"].sum()) == E
    # chunk tables cover every pair exactly once
    for key in ("ca", "ch"):
        assert sorted(t[key + "_pairs"].tolist()) == list(range(c.NP))


def test_padded_export_is_shape_stable_and_inert():
    """dvae_batch_export_padded (include/dvae_hip.h): batches of the same (B, max_n) packed under the same caps have
    IDENTICAL counts (hence blob layout, workspace sizes and launch sequence: what replaying one captured train step
    needs), and the padding entries change nothing: the loss evaluated through the padded tables is the reference's."""
    d, cls, ctor = load_case(golden_files("dvae_tiny_bi_full")[0])
    cfg = cfg_of(cls, ctor)
    sd = state_dict_of(d, torch.float64)
    ga = dvae_amd.GraphArrays.from_dense(d["types"], d["adj"], d["n_g"])
    caps = dvae_amd.batching.padding_caps(ga, len(ga), ctor["nvt"])
    assert caps is not None
    plain = dvae_amd.DAGBatch(ga, ctor["nvt"], 0, False, "cpu")
    padded = dvae_amd.DAGBatch(ga, ctor["nvt"], 0, False, "cpu", caps=caps)
    assert padded.padded and not plain.padded
    c, q = plain.counts, padded.counts
    assert (q.E, q.PT, q.PL, q.NB, q.CH) == (caps.E, caps.PT, caps.PL, caps.NB, caps.CH)
    assert q.PT >= c.PT and q.NB >= c.NB and q.CH >= c.CH + (q.NB - c.NB)
    for f in ("B", "n", "T", "nB", "NA", "NP", "CA", "CC", "R", "RL", "NCE", "HGF", "HGB", "level_major"):
        assert getattr(c, f) == getattr(q, f), f
    mu, lv, (L, R, K) = BI.forward(cfg, sd, padded, torch.from_numpy(d["eps"]).double())
    assert abs(L.item() - d["loss3"][0]) < 1e-6 * d["loss3"][0]
    t = BI.unpack_blob(padded)
    # padding rows of the EB table: gathered from the first type row, referenced by no pair; the zero row is NB_cap
    assert (t["rowsB"][c.NB:] == c.nB).all() and t["pair_h"].max() < c.NB
    # every padding row has an empty backward chunk of its own (its gradient row is written whole: zeros)
    rows_of_empty = set(int(r) & 0x7fffffff for r, a, b in zip(t["ch_row"], t["ch_off"][:-1], t["ch_off"][1:]) if a == b)
    assert set(range(c.NB, q.NB)) <= rows_of_empty
    assert sorted(t["ch_pairs"].tolist()) == list(range(c.NP))
    # another batch of the same shape, other edges: the same counts, field for field
    rng = np.random.default_rng(0)
    adj2 = np.triu((rng.random(d["adj"].shape) < 0.35), 1).astype(np.uint8)
    for v in range(1, adj2.shape[1]):
        adj2[:, v - 1, v] = 1
    other = dvae_amd.DAGBatch(dvae_amd.GraphArrays.from_dense(d["types"], adj2, d["n_g"]), ctor["nvt"], 0, False, "cpu", caps=caps)
    if other.padded:
        for f, _ in q._fields_:
            assert getattr(other.counts, f) == getattr(q, f), f
        assert other.blob_host.numel() == padded.blob_host.numel()
    # a batch above a cap falls back to the plain export
    tight = dvae_amd._lib.BatchCaps()
    tight.E, tight.PT, tight.PL, tight.NB, tight.CH = caps.E, c.PT - 1, caps.PL, caps.NB, caps.CH
    assert not dvae_amd.DAGBatch(ga, ctor["nvt"], 0, False, "cpu", caps=tight).padded
    # graphs of different sizes do not qualify
    d2, cls2, ctor2 = load_case(golden_files("dvae_tiny_bi_var")[0])
    ga2 = dvae_amd.GraphArrays.from_dense(d2["types"], d2["adj"], d2["n_g"])
    assert dvae_amd.batching.padding_caps(ga2, len(ga2), ctor2["nvt"]) is None
    assert not dvae_amd.DAGBatch(ga2, ctor2["nvt"], 0, False, "cpu", caps=caps).padded


def test_packer_level_major_flag():
    """full-size batches (the reference's data sets) are flagged level-major: EA row i and vertex-type row i are
    slot i, which lets the loss forward run those products level by level; ragged or BN batches are not"""
    d, cls, ctor = load_case(golden_files("dvae_cfg1_uni")[0])
    ga = dvae_amd.GraphArrays.from_dense(d["types"], d["adj"], d["n_g"])
    b = dvae_amd.DAGBatch(ga, 8, 0, False, "cpu")
    assert b.counts.level_major == 1
    t = BI.unpack_blob(b)
    assert t["rowsA"].tolist() == list(range(b.counts.NA)) and b.counts.NA == 32 * 7
    assert t["vs_idx"].tolist() == list(range(b.counts.R))
    d, cls, ctor = load_case(golden_files("dvae_tiny_bi_var")[0])
    ga = dvae_amd.GraphArrays.from_dense(d["types"], d["adj"], d["n_g"])
    assert dvae_amd.DAGBatch(ga, ctor["nvt"], 0, False, "cpu").counts.level_major == 0
    d, cls, ctor = load_case(golden_files("bn_tiny_uni_full")[0])
    ga = dvae_amd.GraphArrays.from_dense(d["types"], d["adj"], d["n_g"])
    assert dvae_amd.DAGBatch(ga, ctor["nvt"], 0, True, "cpu").counts.level_major == 0


def test_graph_objects_and_arrays_agree():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle", "igraph_standin"))
    import graph_rows
    d, cls, ctor = load_case(golden_files("dvae_tiny_uni_var")[0])
    graphs = [graph_rows.decode_ENAS_to_igraph(r)[0] for r in d["rows"].tolist()]
    ga = dvae_amd.graphs_to_arrays(graphs, 8)
    types, adj, n_g = ga.to_dense()
    assert (n_g == d["n_g"]).all() and (adj == d["adj"]).all()
    for b, k in enumerate(n_g):
        assert (types[b, :k] == d["types"][b, :k]).all()


def test_malformed_graphs_are_rejected():
    types = np.zeros((1, 4), np.int32)
    types[0] = [0, 2, 3, 1]
    ok = np.array([[0, 1, 3, 4]], np.uint64)
    n_g = np.array([4], np.int32)
    dvae_amd.DAGBatch(dvae_amd.GraphArrays(types, ok, n_g, 4), 5, 0, False, "cpu")
    with pytest.raises(dvae_amd._lib.DvaeHipError, match="topological"):
        bad = ok.copy(); bad[0, 1] = 0b100  # edge 2 -> 1
        dvae_amd.DAGBatch(dvae_amd.GraphArrays(types, bad, n_g, 4), 5, 0, False, "cpu")
    with pytest.raises(dvae_amd._lib.DvaeHipError, match="type"):
        t2 = types.copy(); t2[0, 2] = 7
        dvae_amd.DAGBatch(dvae_amd.GraphArrays(t2, ok, n_g, 4), 5, 0, False, "cpu")
    with pytest.raises(dvae_amd._lib.DvaeHipError, match="vertices"):
        dvae_amd.DAGBatch(dvae_amd.GraphArrays(types, ok, np.array([5], np.int32), 4), 5, 0, False, "cpu")
    with pytest.raises(ValueError):
        dvae_amd.DAGBatch(dvae_amd.GraphArrays(types[:0], ok[:0], n_g[:0], 4), 5, 0, False, "cpu")


def test_state_dict_layout_matches_reference():
    for path in golden_files("*cfg*") + golden_files("*tiny_bi_full"):
        d, cls, ctor = load_case(path)
        model = make_model(cls, ctor, int(d["seed"]))
        assert list(model.state_dict().keys()) == d["param_names"].tolist()
        for k, w in model.state_dict().items():
            s = d["psum::" + k]
            assert abs(w.double().sum().item() - s[0]) < 1e-9, k


def test_shard_bounds_is_ceil_split():
    # util.py:566-572: contiguous chunks of ceil(B / ndev)
    assert [dvae_amd.shard_bounds(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [dvae_amd.shard_bounds(8, 2, r) for r in range(2)] == [(0, 4), (4, 8)]
    assert dvae_amd.shard_bounds(2, 4, 3) == (2, 2)


def test_balanced_shard_bounds_partitions_and_evens_out_pair_work():
    rng = np.random.RandomState(3)
    for world in (1, 2, 4, 8):
        for n_items in (0, 1, 5, 64, 1000):
            nv = rng.randint(3, 65, size=n_items)
            spans = [dvae_amd.balanced_shard_bounds(nv, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n_items
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))       # contiguous, ordered, no overlap
            assert all(lo <= hi for lo, hi in spans)
    # equal sizes: the even split, i.e. the reference's rule
    assert [dvae_amd.balanced_shard_bounds([8] * 8, 2, r) for r in range(2)] == [(0, 4), (4, 8)]
    # skewed batch: small graphs first, large last; the heaviest rank's share of sum n^2 drops vs the ceil split
    nv = np.concatenate([np.full(96, 6), np.full(32, 60)])
    w = nv.astype(np.float64) ** 2

    def worst(bounds):
        return max(w[lo:hi].sum() for lo, hi in bounds) / (w.sum() / 4)
    ceil = worst([dvae_amd.shard_bounds(len(nv), 4, r) for r in range(4)])
    bal = worst([dvae_amd.balanced_shard_bounds(nv, 4, r) for r in range(4)])
    assert bal < 1.1 < ceil


def test_ctypes_struct_mirrors_match_the_header_layout(tmp_path):
    """include/dvae_hip.h compiles as plain C (gcc, the language a foreign-function binding sees), and
    every struct d-vae_amd/_lib.py mirrors with ctypes has the header's size and field offsets."""
    import ctypes
    import subprocess
    L = dvae_amd._lib
    structs = {"dvae_model_desc": L.ModelDesc, "dvae_batch_counts": L.BatchCounts, "dvae_gemm_desc": L.GemmDesc,
               "dvae_level_desc": L.LevelDesc, "dvae_batch_caps": L.BatchCaps}
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "dvae_hip.h"', "int main(void) {"]
    for cname, cls in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ["return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)],
                   check=True, capture_output=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(out[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(out["%s.%s" % (cname, fname)]) == getattr(cls, fname).offset, (cname, fname)


def test_bench_self_launch_plumbing(tmp_path):
    """`python bench.py --gpus N` without a launcher starts its N ranks itself (the parent never touches the
    GPU) and relays rank 0's one JSON line; under a launcher (WORLD_SIZE set) or with one GPU it does not."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    assert bench.launch_plan(["--gpus", "1"], {}) == []
    assert bench.launch_plan(["--gpus", "4"], {"WORLD_SIZE": "4", "RANK": "2"}) == []
    plan = bench.launch_plan(["--gpus", "3", "--steps", "7", "--scaling", "strong"], {"KEEP": "1"}, port=23456)
    assert len(plan) == 3
    for r, (cmd, env) in enumerate(plan):
        assert cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "3", "--steps", "7", "--scaling", "strong"]
        assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"]) == (str(r), str(r), "3")
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "23456" and env["KEEP"] == "1"
    # the relay: rank 0's JSON line comes out once; a failing rank ends the job with its code
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\n"
                      "r = int(os.environ['RANK'])\n"
                      "if os.environ.get('FAIL') == str(r): sys.exit(7)\n"
                      "if os.environ.get('FAIL'): time.sleep(30)\n"
                      "print('noise'); print('{\"rank\": %d}' % r)\n")
    import subprocess
    drv = ("import sys; sys.path.insert(0, %r); import bench, os\n"
           "plan = [([sys.executable, %r], dict(os.environ, RANK=str(r))) for r in range(3)]\n"
           "sys.exit(bench.self_launch(plan))\n" % (root, str(script)))
    ok = subprocess.run([sys.executable, "-c", drv], capture_output=True, text=True, timeout=60)
    assert ok.returncode == 0 and ok.stdout.strip() == '{"rank": 0}'
    bad = subprocess.run([sys.executable, "-c", drv], capture_output=True, text=True, timeout=60, env=dict(os.environ, FAIL="1"))
    assert bad.returncode == 7 and "{" not in bad.stdout
