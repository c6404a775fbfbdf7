"""Generate golden vectors by running the REFERENCE itself (build container only).

Run as:   PYTHONDONTWRITEBYTECODE=1 python -B oracle/gen_golden.py

Imports /root/reference/models.py unmodified (oracle/igraph_standin on
sys.path stands in for python-igraph, which is not installable here) and
writes small .npz fixtures (data only: inputs, weights, expected outputs and
gradients) under tests/golden/.  Nothing of the reference's source travels.
/root/reference does not exist on the GPU box, so this script is never run
there; the fixtures it wrote are what the tests read.

The normal draw inside reference `reparameterize` (models.py:737,
torch.randn_like on the CPU default generator) is pinned by calling
torch.manual_seed(EPS_SEED) right before `loss()`; replaying the same seed
with torch.randn(B, nz) gives the identical `eps`, which is stored.
"""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "igraph_standin"))
sys.path.insert(0, HERE)
sys.path.append("/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import graph_rows  # noqa: E402
import models as ref_models  # noqa: E402  (the reference)

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
EPS_SEED = 7


def enas_rows(lo, hi):
    rows = []
    with open("/root/reference/data/final_structures6.txt") as f:
        for i, line in enumerate(f):
            if lo <= i < hi:
                rows.append(graph_rows.parse_row(line))
            if i >= hi:
                break
    return rows


def truncate_enas(row, k):
    """first k layers of an ENAS row: still a valid ENAS row (k+2 vertices)."""
    return [list(node) for node in row[:k]]


def synth_bn_rows(count, nvars, p, seed):
    """asia-shaped Bayesian-network rows (SURVEY.md 8d cfg 3): variables in a
    random order, edge j -> i (j < i in row order) with probability p."""
    rng = random.Random(seed)
    rows = []
    for _ in range(count):
        order = list(range(nvars))
        rng.shuffle(order)
        row = []
        for i, t in enumerate(order):
            row.append([t] + [1 if rng.random() < p else 0 for _ in range(i)])
        rows.append(row)
    return rows


def run_case(name, cls, ctor, graphs, rows, full_tensors, seed=1, slice_n=16):
    torch.manual_seed(seed)
    model = getattr(ref_models, cls)(**ctor)
    model.train()
    max_n = ctor["max_n"]
    types, adj, n_g = graph_rows.igraphs_to_arrays(graphs, max_n)
    B = len(graphs)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}

    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    opt.zero_grad()
    batch = model._collate_fn(graphs)
    mu, logvar = model.encode(batch)
    torch.manual_seed(EPS_SEED)
    loss, recon, kld = model.loss(mu, logvar, batch)
    loss.backward()
    torch.manual_seed(EPS_SEED)
    eps = torch.randn(B, ctor["nz"])

    out = {
        "cls": np.array(cls), "ctor_keys": np.array(list(ctor.keys())),
        "ctor_vals": np.array([int(v) for v in ctor.values()], np.int64),
        "seed": np.array(seed), "rows": np.array([repr(r) for r in rows]),
        "types": types, "adj": adj, "n_g": n_g, "eps": eps.numpy(),
        "mu": mu.detach().numpy(), "logvar": logvar.detach().numpy(),
        "loss3": np.array([loss.item(), recon.item(), kld.item()], np.float64),
    }
    none_grads = []
    names = []
    for k, p in model.named_parameters():
        names.append(k)
        w = sd0[k]
        out["psum::" + k] = np.array([w.double().sum().item(), w.double().abs().sum().item()])
        if p.grad is None:
            none_grads.append(k)
            continue
        g = p.grad.detach()
        out["gnorm::" + k] = np.array([g.double().norm().item(), g.double().sum().item()])
        if full_tensors:
            out["grad::" + k] = g.numpy()
        else:
            out["gslice::" + k] = g.reshape(-1)[:slice_n].numpy().copy()
    if full_tensors:
        for k, w in sd0.items():
            out["sd::" + k] = w.numpy()
    out["param_names"] = np.array(names)
    out["none_grads"] = np.array(none_grads if none_grads else [""])
    # one Adam step: parameter-delta checksum
    opt.step()
    dsum = 0.0
    for k, p in model.named_parameters():
        dsum += (p.detach().double() - sd0[k].double()).abs().sum().item()
    out["adam_delta_abs_sum"] = np.array(dsum)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("%-28s B=%d loss=%.6f recon=%.6f kld=%.6f none_grads=%d" % (
        name, B, loss.item(), recon.item(), kld.item(), len(none_grads)))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    rows8 = enas_rows(1000, 1008)
    g8 = [graph_rows.decode_ENAS_to_igraph(r)[0] for r in rows8]
    # variable-size ENAS batch: k layers, k in 1..6  (k+2 vertices <= 8)
    ks = [6, 3, 1, 5, 2, 4, 6, 2]
    rows_var = [truncate_enas(r, k) for r, k in zip(rows8, ks)]
    g_var = [graph_rows.decode_ENAS_to_igraph(r)[0] for r in rows_var]
    tiny = dict(max_n=8, nvt=8, START_TYPE=0, END_TYPE=1, hs=20, nz=6)
    for bidir in (0, 1):
        tag = "bi" if bidir else "uni"
        run_case("dvae_tiny_%s_full" % tag, "DVAE", dict(tiny, bidirectional=bidir), g8, rows8, True)
        run_case("dvae_tiny_%s_var" % tag, "DVAE", dict(tiny, bidirectional=bidir), g_var, rows_var, True)
    run_case("dvae_tiny_novid_full", "DVAE", dict(tiny, bidirectional=1, vid=0), g8, rows8, True)

    # Bayesian networks (asia shape: 8 variables -> 10 vertices, 10 types)
    bn_rows = synth_bn_rows(6, 8, 0.25, 0)
    g_bn = [graph_rows.decode_BN_to_igraph(r)[0] for r in bn_rows]
    # variable-size BN batch: sub-networks over the first k variables of each row
    kb = [8, 3, 5, 8, 2, 6]
    bn_rows_var = [[list(nd) for nd in r[:k]] for r, k in zip(bn_rows, kb)]
    g_bn_var = [graph_rows.decode_BN_to_igraph(r)[0] for r in bn_rows_var]
    tiny_bn = dict(max_n=10, nvt=10, START_TYPE=0, END_TYPE=1, hs=20, nz=6)
    for bidir in (0, 1):
        tag = "bi" if bidir else "uni"
        run_case("bn_tiny_%s_full" % tag, "DVAE_BN", dict(tiny_bn, bidirectional=bidir), g_bn, bn_rows, True)
        run_case("bn_tiny_%s_var" % tag, "DVAE_BN", dict(tiny_bn, bidirectional=bidir), g_bn_var, bn_rows_var, True)

    # BASELINE config 1 shape: hs=501, nz=56, file lines 1001..1032, seed 1
    rows32 = enas_rows(1000, 1032)
    g32 = [graph_rows.decode_ENAS_to_igraph(r)[0] for r in rows32]
    big = dict(max_n=8, nvt=8, START_TYPE=0, END_TYPE=1, hs=501, nz=56)
    run_case("dvae_cfg1_uni", "DVAE", dict(big, bidirectional=0), g32, rows32, False)
    run_case("dvae_cfg1_bi", "DVAE", dict(big, bidirectional=1), g32, rows32, False)
    bn32 = synth_bn_rows(32, 8, 0.25, 0)
    gbn32 = [graph_rows.decode_BN_to_igraph(r)[0] for r in bn32]
    bigbn = dict(max_n=10, nvt=10, START_TYPE=0, END_TYPE=1, hs=501, nz=56)
    run_case("bn_cfg3_uni", "DVAE_BN", dict(bigbn, bidirectional=0), gbn32, bn32, False)


if __name__ == "__main__":
    main()
