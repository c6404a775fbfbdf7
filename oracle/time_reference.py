"""Times the REFERENCE's own train step on the host CPU (build container only) next to the
oracle's reference-style port, on BASELINE config 1 (DVAE unidirectional, hs=501, nz=56, 32 ENAS-6
graphs: file lines 1001-1032 of data/final_structures6.txt), so that the `cpu_baseline` that
bench.py reports on the GPU box (kind "port") can be related to the reference itself.

Run as:   PYTHONDONTWRITEBYTECODE=1 python -B oracle/time_reference.py [threads]
"""
import os
import statistics
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "igraph_standin"))
sys.path.insert(0, HERE)
sys.path.append("/root/reference")

import torch  # noqa: E402

import graph_rows  # noqa: E402
import dvae_oracle as O  # noqa: E402
import models as ref_models  # noqa: E402  (the reference)


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    torch.set_num_threads(threads)
    rows = []
    with open("/root/reference/data/final_structures6.txt") as f:
        for i, line in enumerate(f):
            if 1000 <= i < 1032:
                rows.append(graph_rows.parse_row(line))
    graphs = [graph_rows.decode_ENAS_to_igraph(r)[0] for r in rows]
    for bidir in (False, True):
        torch.manual_seed(1)
        model = ref_models.DVAE(8, 8, 0, 1, hs=501, nz=56, bidirectional=bidir)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-4)

        def ref_step():
            opt.zero_grad()
            g = model._collate_fn(graphs)
            mu, logvar = model.encode(g)
            loss, _, _ = model.loss(mu, logvar, g)
            loss.backward()
            opt.step()

        types, adj, n_g = (torch.from_numpy(x) for x in graph_rows.igraphs_to_arrays(graphs, 8))
        cfg = O.make_cfg(8, 8, 0, 1, 501, 56, bidir, True, False)
        params = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
        popt = torch.optim.Adam(list(params.values()), lr=1e-4)
        eps = torch.randn(32, 56)

        def port_step():
            popt.zero_grad()
            mu, lv = O.encode(cfg, params, types, adj, n_g, mode="refstyle")
            loss, _, _ = O.loss(cfg, params, mu, lv, types, adj, n_g, eps, mode="refstyle")
            loss.backward()
            popt.step()

        out = []
        for name, fn in (("reference", ref_step), ("oracle port", port_step)):
            for _ in range(2):
                fn()
            ts = []
            for _ in range(10):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            med = statistics.median(ts)
            out.append(med)
            print("%-12s bidirectional=%-5s %d threads: median %.3f s per 32-graph step = %.1f DAGs/s"
                  % (name, bidir, threads, med, 32 / med))
        print("  port / reference speed ratio: %.2f" % (out[0] / out[1]))


main()
