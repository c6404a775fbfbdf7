#!/usr/bin/env python3
"""End-to-end epoch throughput INCLUDING batching: harness.train_epoch over a synthetic data set
of ENAS-6 shape (the reference's set has 19 020 graphs; 17 118 train), shuffled every epoch and
packed batch by batch on the host while the GPU trains (what bench.py excludes by pre-packing).
usage: python tools/epoch_bench.py [n_graphs] [batch] [cfg]"""
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from _dvae_pkg import dvae_amd  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 17118
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    cfg = bench.CONFIGS[sys.argv[3] if len(sys.argv) > 3 else "cfg2"]
    ga = bench.make_graphs(cfg, n, 0)
    y = [0.0] * n
    model = bench.make_model(cfg).cuda()
    opt = dvae_amd.optim.FlatAdam(model, lr=1e-4)
    rng = random.Random(0)
    for prefetch in (False, True):
        dvae_amd.harness.train_epoch(model, opt, (ga[:4 * bs], y[:4 * bs]), bs, rng, prefetch=prefetch)  # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tl, rl, kl = dvae_amd.harness.train_epoch(model, opt, (ga, y), bs, rng, prefetch=prefetch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("epoch of %d graphs, batch %d, prefetch=%s: %.3f s, %.0f DAGs/s (loss/graph %.3f)"
              % (n, bs, prefetch, dt, n / dt, tl / n))


main()
