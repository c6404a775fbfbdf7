#!/usr/bin/env python3
"""GPU time of each phase of the train step (HIP events on the compute stream, no profiler):
encode fwd | loss fwd | loss bwd | encode bwd (+ grad accumulation) | optimizer.
usage: python tools/phase_times.py [cfg2] [steps] [f32|bf16]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from _dvae_pkg import dvae_amd  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    cfg = bench.CONFIGS[name]
    dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
    dev = torch.device("cuda", 0)
    model = bench.make_model(cfg, mfma_dtype="bf16" if dtype == "bf16" else "fp32").to(dev).train()
    reducer = dvae_amd.dist.GradSumReducer(model)
    opt = dvae_amd.optim.FlatAdam(model, lr=1e-4, reducer=reducer)
    B = cfg["B"]
    nb = min(4, cfg.get("n_batches", 4))
    batches = [model._collate_fn(bench.make_graphs(cfg, B, seed=i)) for i in range(nb)]
    eps = [torch.randn(B, cfg["nz"], device=dev) for _ in range(nb)]
    names = ["encode fwd", "loss fwd", "loss bwd", "encode bwd", "optimizer"]
    tot = [0.0] * 5
    for it in range(steps + 5):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        b = batches[it % nb]
        reducer.zero()
        ev[0].record()
        mu, logvar = model.encode(b)
        ev[1].record()
        mu.register_hook(lambda g, e=ev[3]: (e.record(), g)[1])
        loss, _, _ = model.loss(mu, logvar, b, eps=eps[it % nb])
        ev[2].record()
        loss.backward()
        ev[4].record()
        reducer.all_reduce()
        opt.step()
        ev[5].record()
        torch.cuda.synchronize()
        if it >= 5:
            for k in range(5):
                tot[k] += ev[k].elapsed_time(ev[k + 1])
    s = sum(tot) / steps
    for k in range(5):
        print("%-12s %8.3f ms" % (names[k], tot[k] / steps))
    print("%-12s %8.3f ms  (with a device sync per step: the host cannot run ahead)" % ("sum", s))


main()
