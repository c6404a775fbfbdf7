"""Golden vectors for the BASELINE cfg 4 and cfg 5 SHAPES, produced by running the REFERENCE itself
(build container only; same mechanism and file format as oracle/gen_golden.py, which it reuses).

Run as:   PYTHONDONTWRITEBYTECODE=1 python -B oracle/gen_golden_cfg45.py

  dvae_cfg4_bi   DVAE(14, 8, 0, 1, hs=501, nz=56, bidirectional)   16 synthetic ENAS-12 rows
                 (SURVEY.md 8d cfg 4: 12 layers, 6 op types, optional edges Bernoulli(0.30), Random(0))
  dvae_cfg5_bi   DVAE(64, 18, 0, 1, hs=1024, nz=56, bidirectional)  4 synthetic 62-layer rows
                 (cfg 5: 16 op types, edge j -> i with probability min(0.30, 3 / i), Random(0)),
                 two of them truncated (ragged batch: 64, 32, 64 and 13 vertices)

Weights are not stored at these sizes: the tests re-derive them from the seed through the product
module (same construction order as the reference) and check them against the stored checksums.
/root/reference does not exist on the GPU box: this script never runs there.
"""
import random

import gen_golden as G  # imports the reference (read-only) and the igraph stand-in
import graph_rows


def synth_enas_rows(count, n_layers, n_types, p_edge, seed, p_rule=None):
    """the generator of d-vae_amd/data.py: synth_enas_rows, restated so that this script depends on the
    reference and the oracle only"""
    rng = random.Random(seed)
    rows = []
    for _ in range(count):
        row = []
        for i in range(n_layers):
            p = p_rule(i) if p_rule else p_edge
            row.append([rng.randrange(n_types)] + [1 if rng.random() < p else 0 for _ in range(i)])
        rows.append(row)
    return rows


def main():
    G.torch.set_num_threads(8)
    rows12 = synth_enas_rows(16, 12, 6, 0.30, 0)
    g12 = [graph_rows.decode_ENAS_to_igraph(r)[0] for r in rows12]
    cfg4 = dict(max_n=14, nvt=8, START_TYPE=0, END_TYPE=1, hs=501, nz=56, bidirectional=1)
    G.run_case("dvae_cfg4_bi", "DVAE", cfg4, g12, rows12, False)

    rows62 = synth_enas_rows(4, 62, 16, 0.30, 0, lambda i: min(0.30, 3.0 / i) if i > 0 else 0.0)
    ks = [62, 30, 62, 11]
    rows62 = [G.truncate_enas(r, k) for r, k in zip(rows62, ks)]
    g62 = [graph_rows.decode_ENAS_to_igraph(r)[0] for r in rows62]
    cfg5 = dict(max_n=64, nvt=18, START_TYPE=0, END_TYPE=1, hs=1024, nz=56, bidirectional=1)
    G.run_case("dvae_cfg5_bi", "DVAE", cfg5, g62, rows62, False)


if __name__ == "__main__":
    main()
