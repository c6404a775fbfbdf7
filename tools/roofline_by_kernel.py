#!/usr/bin/env python3
"""Per-kernel-family share of the cfg 2 train step and fraction of the fp32 MFMA peak.

Two inputs, both produced on the GPU box by tools/profile_round.sh:
  * the rocprofv3 --kernel-trace --stats summary of `bench.py` (profiles/rNN_bench_cfg2_kernel_stats.csv): calls and
    total duration per kernel over the profiled run;
  * the launch log of ONE train step under DVAE_GEMM_LOG=1 (gpurun_out/prof/launch_log_cfg2.txt; this script makes
    it when called with --log on a GPU box): one line per matrix-unit launch with its kernel family and M, N, K.
FLOP per family = sum of 2 M N K over the family's launches of one step (padded sizes, what the kernel executes);
time per family = its share of the profiled kernel time; achieved = FLOP / time of the launches alone (kernel
durations as rocprof reports them, launches on two streams overlap, so the shares add up to more than the step).

usage: python tools/roofline_by_kernel.py --log out.txt            (GPU box: write the launch log of one step)
       python tools/roofline_by_kernel.py r02 [steps_in_profile]   (anywhere: profiles/r02_* -> profiles/r02_roofline_by_kernel.json)"""
import collections
import csv
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 157.3


def write_log(path):
    code = r"""
import os, sys, torch
sys.path.insert(0, %r)
import bench
from _dvae_pkg import dvae_amd
cfg = bench.CONFIGS["cfg2"]
dev = torch.device("cuda", 0)
model = bench.make_model(cfg).to(dev).train()
reducer = dvae_amd.dist.GradSumReducer(model)
opt = dvae_amd.optim.FlatAdam(model, lr=1e-4, reducer=reducer)
b = model._collate_fn(bench.make_graphs(cfg, cfg["B"], seed=0))
eps = torch.randn(cfg["B"], cfg["nz"], device=dev)
def step():
    reducer.zero()
    mu, logvar = model.encode(b)
    loss, _, _ = model.loss(mu, logvar, b, eps=eps)
    loss.backward()
    reducer.all_reduce()
    opt.step()
step(); torch.cuda.synchronize()
sys.stderr.write("STEP BEGIN\n"); sys.stderr.flush()
step(); torch.cuda.synchronize()
sys.stderr.write("STEP END\n"); sys.stderr.flush()
""" % ROOT
    env = dict(os.environ, DVAE_GEMM_LOG="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    if r.returncode != 0:
        sys.exit(r.stderr[-2000:])
    txt = r.stderr
    body = txt[txt.index("STEP BEGIN"):txt.index("STEP END")]
    open(path, "w").write("\n".join(l for l in body.splitlines() if l.startswith(("GEMMSEL", "CHAINSEL"))) + "\n")
    print("wrote", path)


def family_of_gemm(tile, akm, bkm):
    bm, bn = tile.split("x")
    wm, wn = ("1", "4") if tile == "32x128" else ("2", "2")
    return "gemm_f32_kernel<%s, %s, %s, %s, %s, %s" % (bm, bn, wm, wn, "true" if akm else "false", "true" if bkm else "false")


def main():
    if sys.argv[1] == "--log":
        return write_log(sys.argv[2])
    r = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 13  # bench.py --steps 10 --warmup 3 in tools/profile_round.sh
    log = os.path.join(ROOT, "profiles", "%s_launch_log_cfg2.txt" % r)
    flop = collections.Counter()
    launches = collections.Counter()
    for l in open(log):
        f = l.split()
        if f[0] == "GEMMSEL":
            M, N, K = int(f[2]), int(f[3]), int(f[4])
            fam = family_of_gemm(f[1], f[5] == "akm=1", f[6] == "bkm=1")
        else:
            fam = l[len("CHAINSEL "):l.index(">") + 1]
            M, N, K = (int(x) for x in l[l.index(">") + 1:].split())
        flop[fam] += 2.0 * M * N * K
        launches[fam] += 1
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "%s_bench_cfg2_kernel_stats.csv" % r))))
    # kernels the profiled command ALSO runs on their own (bench.py's roofline legs): their average duration alone, from
    # the rocprofv3 stats of tools/chain_bench.py (profiles/rNN_chain_level_alone_kernel_stats.csv), to take those
    # launches out of the in-step figures
    alone = {}
    ap = os.path.join(ROOT, "profiles", "%s_chain_level_alone_kernel_stats.csv" % r)
    if os.path.exists(ap):
        for x in csv.DictReader(open(ap)):
            alone[x["Name"]] = float(x["AverageNs"])
    total_ns = sum(float(x["TotalDurationNs"]) for x in rows)
    fams = collections.OrderedDict()
    for x in rows:
        name = x["Name"].replace("void dvae::", "").replace("dvae::", "")
        key = None
        for fam in flop:
            if name.startswith(fam):
                key = fam
        if key is None:
            key = re.sub(r"\(.*", "", name)
        e = fams.setdefault(key, dict(ns=0.0, calls=0))
        ns, calls = float(x["TotalDurationNs"]), int(x["Calls"])
        if key in flop and x["Name"] in alone and calls > steps * launches[key]:
            extra = calls - steps * launches[key]  # the roofline leg's launches of this kernel
            ns -= extra * alone[x["Name"]]
            calls -= extra
        e["ns"] += ns
        e["calls"] += calls
    out = dict(workload="cfg2 fp32, bench.py train step", steps_in_profile=steps, peak_tflops=PEAK,
               kernel_ms_per_step=total_ns / steps / 1e6, families=[])
    for k, e in sorted(fams.items(), key=lambda kv: -kv[1]["ns"]):
        ms = e["ns"] / steps / 1e6
        if k in flop and abs(e["calls"] / steps - launches[k]) > 0.01:
            # the profiled command also times the dominant product on its own (bench.py's roofline leg: 33 launches of
            # this family): price the step's launches at the family's average duration
            ms = e["ns"] / e["calls"] * launches[k] / 1e6
        d = dict(kernel=k, launches_per_step=round(e["calls"] / steps, 1), ms_per_step=round(ms, 4),
                 share_of_kernel_time=round(ms / (total_ns / steps / 1e6), 4))
        if k in flop:
            d["gflop_per_step"] = round(flop[k] / 1e9, 2)
            d["achieved_tflops"] = round(flop[k] / (ms * 1e-3) / 1e12, 1)
            d["frac_of_peak"] = round(d["achieved_tflops"] / PEAK, 3)
            d["launches_in_log"] = launches[k]
        if d["share_of_kernel_time"] >= 0.004 or k in flop:
            out["families"].append(d)
    path = os.path.join(ROOT, "profiles", "%s_roofline_by_kernel.json" % r)
    json.dump(out, open(path, "w"), indent=1)
    for d in out["families"][:14]:
        print(d)
    print("wrote", path)


main()
