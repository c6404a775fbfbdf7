#!/usr/bin/env python3
"""End-to-end run of the reference's ENAS recipe on the GPU path (README.md:37 of the reference:
DVAE --bidirectional, hs=501, nz=56, batch 32, lr 1e-4, ReduceLROnPlateau): per-epoch losses, then test NLL,
reconstruction accuracy and prior validity / uniqueness / novelty, as train.py reports them.
Data: SYNTHETIC ENAS-6 graphs by default, or -- with `--data <name>` -- the reference's own rows
(tests/golden/final_structures6.txt, a data fixture: the 20 020 rows of the reference's data/final_structures6.txt)
through data.load_ENAS_graphs: burn-in 1000, random.Random(0) shuffle, 90/10 split (util.py:64-93).
The reference holds no result fixture for such a run: its numbers are "parity unpinned".
usage: python tools/train_demo.py [epochs] [n_graphs] [batch] [--data final_structures6]"""
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
    argv = list(sys.argv[1:])
    data_name = None
    if "--data" in argv:
        i = argv.index("--data")
        data_name = argv[i + 1]
        del argv[i:i + 2]
    epochs = int(argv[0]) if len(argv) > 0 else 30
    n = int(argv[1]) if len(argv) > 1 else 19020
    bs = int(argv[2]) if len(argv) > 2 else 32
    if data_name:
        train, test, ga_args = D.load_ENAS_graphs(data_name, n_types=6, fmt="arrays", rand_seed=0, with_y=True, burn_in=1000,
                                                  data_dir=os.path.join(ROOT, "tests", "golden"))
        assert ga_args.max_n == 8 and ga_args.num_vertex_type == 8
        cut = len(train[0])
        print("data: %s, %d train / %d test graphs (burn-in 1000, Random(0) 90/10 split)" % (data_name, cut, len(test[0])), flush=True)
    else:
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
