#!/usr/bin/env python3
"""Host cost of enqueuing one train step, separated from the GPU's time: the stream is first blocked behind
~100 ms of unrelated work (large torch.mm products), then K steps are enqueued and the host clock is read
BEFORE any synchronisation; the GPU time of the same K steps is read from events afterwards.
usage: python tools/host_cost.py [cfg1] [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from _dvae_pkg import dvae_amd  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    cfg = bench.CONFIGS[name]
    dev = torch.device("cuda", 0)
    model = bench.make_model(cfg).to(dev).train()
    reducer = dvae_amd.dist.GradSumReducer(model)
    opt = dvae_amd.optim.FlatAdam(model, lr=1e-4, reducer=reducer)
    B = cfg["B"]
    batches = [model._collate_fn(bench.make_graphs(cfg, B, seed=i)) for i in range(4)]
    eps = [torch.randn(B, cfg["nz"], device=dev) for _ in range(4)]

    def step(i):
        b = batches[i % 4]
        reducer.zero()
        mu, logvar = model.encode(b)
        loss, _, _ = model.loss(mu, logvar, b, eps=eps[i % 4])
        loss.backward()
        reducer.all_reduce()
        opt.step()

    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    big = torch.randn(8192, 8192, device=dev)
    out = torch.empty_like(big)
    for rep in range(3):
        for _ in range(12):
            torch.mm(big, big, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        for i in range(K):
            step(i)
        t1 = time.perf_counter()
        e1.record()
        torch.cuda.synchronize()
        print("host enqueue %.3f ms/step   GPU (back to back, no host in the way) %.3f ms/step" %
              ((t1 - t0) * 1e3 / K, e0.elapsed_time(e1) / K))


main()
