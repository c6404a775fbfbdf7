#!/usr/bin/env python3
"""Stage times of the persistent small-batch chains' forward GRU phases (diagnostic build exp_lib/libdvae_tprof.so,
tools/tiny_ablate.sh): a few cfg 1 train steps, then the per-phase means of workgroup 0's stage stamps.
usage: python tools/tiny_prof.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from _dvae_pkg import dvae_amd  # noqa: E402

dvae_amd._lib.LIB_PATH = os.path.join(ROOT, "exp_lib", "libdvae_tprof.so")
lib = dvae_amd._lib.load()
cfg = bench.CONFIGS["cfg1"]
model = bench.make_model(cfg).cuda().train()
opt = dvae_amd.optim.FlatAdam(model, lr=1e-4)
batches = [model._collate_fn(bench.make_graphs(cfg, 32, seed=i)) for i in range(4)]
eps = [torch.randn(32, cfg["nz"], device="cuda") for _ in range(4)]


def step(i):
    b = batches[i % 4]
    opt.zero_grad()
    mu, lv = model.encode(b)
    loss, _, _ = model.loss(mu, lv, b, eps=eps[i % 4])
    loss.backward()
    opt.step()


for i in range(10):
    step(i)
out = (C.c_ulonglong * 16)()
fn = lib.dvae_debug_tiny_prof
fn.restype = C.c_int
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
fn(out, 1)
for i in range(50):
    step(i)
fn(out, 0)
n = max(1, out[5])
names = ["prefetch issue", "wait", "A loads + MFMA + slice exchange", "tail arithmetic + store issue", "drain + signal"]
print("forward GRU phases of workgroup 0: %d phases" % n)
for k, name in enumerate(names):
    print("  %-34s %7.3f us per phase" % (name, out[k] * 0.01 / n))
print("  %-34s %7.3f us per phase" % ("sum", sum(out[:5]) * 0.01 / n))
n = max(1, out[13])
print("forward gate / mapper phases of workgroup 0: %d phases" % n)
for k, name in enumerate(names):
    print("  %-34s %7.3f us per phase" % (name, out[8 + k] * 0.01 / n))
print("  %-34s %7.3f us per phase" % ("sum", sum(out[8:13]) * 0.01 / n))
