#!/usr/bin/env python3
"""End-to-end run of the reference's ENAS recipe on the GPU path (README.md:37 of the reference:
DVAE --bidirectional, hs=501, nz=56, batch 32, lr 1e-4, ReduceLROnPlateau) on SYNTHETIC ENAS-6
graphs (the real data file cannot travel to the GPU box): per-epoch losses, then test NLL,
reconstruction accuracy and prior validity / uniqueness / novelty, as train.py reports them.
usage: python tools/train_demo.py [epochs] [n_graphs] [batch]"""
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _dvae_pkg import dvae_amd  # noqa: E402
from dvae_amd import data as D, harness as H  # noqa: E402


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 19020
    bs = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    rows = D.synth_enas_rows(n, 6, 6, 0.30, seed=0)
    order = list(range(n))
    random.Random(0).shuffle(order)
    ga = D.enas_rows_to_arrays(rows, 8)
    cut = int(n * 0.9)
    train = (ga.select(order[:cut]), [0.0] * cut)
    test = (ga.select(order[cut:]), [0.0] * (n - cut))
    torch.manual_seed(1)
    model = dvae_amd.DVAE(8, 8, 0, 1, hs=501, nz=56, bidirectional=True).cuda()
    opt = dvae_amd.optim.FlatAdam(model, lr=1e-4)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, "min", factor=0.1, patience=10)
    rng = random.Random(1)
    t0 = time.perf_counter()
    for epoch in range(1, epochs + 1):
        tl, rl, kl = H.train_epoch(model, opt, train, bs, rng)
        sched.step(tl)
        if epoch == 1 or epoch % 5 == 0 or epoch == epochs:
            print("epoch %3d  loss %.3f  recon %.3f  kld %.3f  (per graph; %.1f s elapsed)"
                  % (epoch, tl / cut, rl / cut, kl / cut, time.perf_counter() - t0), flush=True)
    t1 = time.perf_counter()
    nll, acc = H.test(model, (test[0][:500], test[1][:500]), 128, 10, 10)
    print("test (500 graphs, 10 x 10 decodings): NLL %.3f  reconstruction accuracy %.4f  (%.1f s)"
          % (nll, acc, time.perf_counter() - t1), flush=True)
    rv, ru, rn = H.prior_validity(model, train, "ENAS", 1000, 10, 128, scale_to_train_range=True)
    print("prior: valid %.4f  unique %.4f  novel %.4f" % (rv, ru, rn))


main()
