#!/usr/bin/env python3
"""Looks for host/GPU ordering bugs in the training loop: the same epochs (same seed, same shuffles) once as the
harness runs them -- asynchronously, the host steps ahead of the GPU, batches packed by the prefetch thread and
dropped after use -- and once with a device synchronisation after every step (nothing can overlap, every batch
outlives its kernels).  Floating-point atomics make the two runs differ in the last bits only; a lifetime or
stream-ordering bug shows as a different loss trajectory (or a fault).
usage: python tools/race_check.py [cfg1] [batch] [n_graphs] [epochs]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from _dvae_pkg import dvae_amd  # noqa: E402


def run(cfg, bs, n, epochs, sync):
    ga = bench.make_graphs(cfg, n, 0)
    model = bench.make_model(cfg).cuda()
    opt = dvae_amd.optim.FlatAdam(model, lr=1e-4)
    rng = random.Random(0)
    torch.manual_seed(0)
    out = []
    hook = (lambda done, total: torch.cuda.synchronize()) if sync else None
    for _ in range(epochs):
        out.append(dvae_amd.harness.train_epoch(model, opt, (ga, [0.0] * n), bs, rng, prefetch=not sync, on_batch=hook))
    torch.cuda.synchronize()
    return out


def main():
    cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg1"]
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    epochs = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    a = run(cfg, bs, n, epochs, sync=False)
    b = run(cfg, bs, n, epochs, sync=True)
    worst = 0.0
    for ea, eb in zip(a, b):
        for x, y in zip(ea, eb):
            worst = max(worst, abs(x - y) / max(abs(y), 1e-9))
    print("async", [round(e[0] / n, 4) for e in a], "sync", [round(e[0] / n, 4) for e in b], "worst rel diff %.2e" % worst)
    if worst > 5e-3:
        sys.exit("MISMATCH between the asynchronous and the synchronised run")
    print("race check ok")


main()
