#!/usr/bin/env python3
"""Training-curve sanity of the two arithmetic modes: the same model, data order and noise draws
trained for a few hundred steps with fp32 contractions and with bf16 MFMA inputs; prints the loss
per graph every 25 steps.  (bf16 is the separately reported reduced-precision mode.)
usage: python tools/converge.py [steps] [cfg]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from _dvae_pkg import dvae_amd  # noqa: E402


def run(cfg, mode, steps, batches, eps):
    model = bench.make_model(cfg, mfma_dtype=mode).cuda().train()
    opt = dvae_amd.optim.FlatAdam(model, lr=1e-4)
    B = cfg["B"]
    out = []
    for i in range(steps):
        b = batches[i % len(batches)]
        opt.zero_grad()
        mu, lv = model.encode(b)
        loss, _, _ = model.loss(mu, lv, b, eps=eps[i % len(eps)])
        loss.backward()
        opt.step()
        if i % 25 == 0 or i == steps - 1:
            out.append((i, loss.item() / B))
    return out


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    cfg = dict(bench.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "cfg2"])
    B = cfg["B"]
    probe = bench.make_model(cfg).cuda()
    batches = [probe._collate_fn(bench.make_graphs(cfg, B, seed=i)) for i in range(8)]
    g = torch.Generator(device="cuda").manual_seed(7)
    eps = [torch.randn(B, cfg["nz"], device="cuda", generator=g) for _ in range(8)]
    a = run(cfg, "fp32", steps, batches, eps)
    b = run(cfg, "bf16", steps, batches, eps)
    print("step   loss/graph fp32   loss/graph bf16   rel diff")
    for (i, x), (_, y) in zip(a, b):
        print("%4d   %14.4f   %14.4f   %8.1e" % (i, x, y, abs(x - y) / abs(x)))


main()
