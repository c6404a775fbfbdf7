#!/usr/bin/env python3
"""Where the HOST's time goes while it enqueues batch-32 train steps: cProfile over 300 steps (cfg 1).
usage: python tools/host_profile.py [cfg1] [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from _dvae_pkg import dvae_amd  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 300
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


for i in range(20):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("plain: %.3f ms per step (host loop, queue may fill)" % ((t1 - t0) * 1e3 / K))
pr = cProfile.Profile()
pr.enable()
for i in range(K):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)

# ---- second view: wall time inside each C-ABI entry point with an EMPTY queue (device synchronised before every step),
# i.e. the pure enqueue cost of each library call, and what is left for Python / autograd around them
import collections  # noqa: E402

lib = dvae_amd._lib.load()
acc = collections.defaultdict(float)
cnt = collections.defaultdict(int)


class Timed:
    def __init__(self, name, fn):
        self.name, self.fn = name, fn

    def __call__(self, *a):
        t = time.perf_counter()
        r = self.fn(*a)
        acc[self.name] += time.perf_counter() - t
        cnt[self.name] += 1
        return r


for nm in list(dvae_amd._lib.SIGNATURES):
    if nm.startswith("dvae_") and nm not in ("dvae_last_error",):
        setattr(lib, nm, Timed(nm, getattr(lib, nm)))
tot = 0.0
for i in range(K):
    torch.cuda.synchronize()
    t = time.perf_counter()
    step(i)
    tot += time.perf_counter() - t
torch.cuda.synchronize()
print("empty queue: %.3f ms per step on the host" % (tot * 1e3 / K))
inside = 0.0
for nm, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-32s %5.1f calls/step  %8.1f us/step" % (nm, cnt[nm] / K, v * 1e6 / K))
    inside += v
print("  inside the library %.1f us/step, Python + autograd + torch around it %.1f us/step" % (inside * 1e6 / K, (tot - inside) * 1e6 / K))
