#!/usr/bin/env python3
"""Throughput of decode() (generation): graphs per second on the GPU for a
batch of latent points, next to the oracle's graph-by-graph CPU restatement on a small sample.
usage: python tools/decode_bench.py [B] [cfg2|cfg3|cfg4]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402
import dvae_oracle as O  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    name = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
    cfg = bench.CONFIGS[name]
    model = bench.make_model(cfg).cuda().eval()
    z = torch.randn(B, cfg["nz"], device="cuda")
    for stochastic in (True, False):
        for _ in range(3):
            ga = model.decode_arrays(z, stochastic)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        iters = 10
        for _ in range(iters):
            ga = model.decode_arrays(z, stochastic)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        print("%s %s decode, B=%d, stochastic=%s: %.2f ms per batch, %.0f graphs/s (mean size %.2f vertices)"
              % (name, cfg["cls"], B, stochastic, dt * 1e3, B / dt, float(np.mean(ga.n_vertices))))
    # CPU: the oracle's restatement of the reference loop, 16 graphs
    ocfg = O.make_cfg(cfg["max_n"], cfg["nvt"], 0, 1, cfg["hs"], cfg["nz"], cfg["bidir"], True, cfg["cls"] == "DVAE_BN")
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    n = cfg["max_n"]
    rng = np.random.default_rng(0)
    t0 = time.perf_counter()
    with torch.no_grad():
        O.decode(ocfg, sd, z[:16].cpu(), True, rng.random((16, n)), rng.random((16, n, n)))
    dt = time.perf_counter() - t0
    print("oracle (CPU, %d threads) decode of 16 graphs: %.1f graphs/s" % (torch.get_num_threads(), 16 / dt))


main()
