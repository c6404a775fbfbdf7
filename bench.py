#!/usr/bin/env python3
"""Benchmark of the D-VAE train step on MI355X (driver contract: see README).

  python bench.py --gpus N --steps K --warmup W [--config cfg2]

A "step" is one full train step over one batch of synthetic DAGs held on the GPU:
zero_grad -> encode -> loss -> backward -> [gradient SUM all-reduce] -> Adam.
Batches are packed before the timed region (pre-packed data set; packing time is
printed separately as `pack_ms_per_batch`).  N > 1: launched by torch.distributed.run,
one rank per GPU, per-GPU batch fixed (weak scaling), no collective other than
the one gradient all-reduce.  Rank 0 prints ONE JSON line.

The JSON line carries
  roofline      the dominant kernel (fp32 MFMA GEMM) timed live with HIP events on
                the stream it runs on, on the largest product of the step
  step_roofline whole-step algorithmic FLOPs (6 * F_alg) / step time
  cpu_baseline  the oracle's reference-style CPU port timed on the host cores
                (rank 0, N = 1 only, bounded sample)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from _dvae_pkg import dvae_amd  # noqa: E402

PEAK_F32_TFLOPS = 157.3    # MI355X fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak
# Memory-side traffic of one launch of the dominant GEMM on its cfg-2 shape (9251 x 2048 x 512),
# from rocprofv3 PMC passes (profiles/README.md): FETCH_SIZE 58717 KB, doubled per the gfx950
# correction of MI355X_MICROARCH.md (section HBM), + WRITE_SIZE 74037 KB (profiles/r01_gemm_pmc.json).  Infinity-Cache hits are
# counted by these counters, so this is traffic behind L2, an upper bound on HBM bytes.
MEASURED_TRAFFIC = {(9251, 2048, 512): (2 * 58717 + 74037) * 1024}

CONFIGS = {
    # name: (class, ctor kwargs, per-GPU batch, generator)
    "cfg1": dict(cls="DVAE", max_n=8, nvt=8, hs=501, nz=56, bidir=False, B=32, kind="enas", layers=6, ntypes=6),
    "cfg2": dict(cls="DVAE", max_n=8, nvt=8, hs=501, nz=56, bidir=True, B=1024, kind="enas", layers=6, ntypes=6),
    "cfg3": dict(cls="DVAE_BN", max_n=10, nvt=10, hs=501, nz=56, bidir=False, B=4096, kind="bn", layers=8, ntypes=8),
    "cfg4": dict(cls="DVAE", max_n=14, nvt=8, hs=501, nz=56, bidir=True, B=1024, kind="enas", layers=12, ntypes=6),
    # 64-vertex synthetic DAGs: layer i takes an optional edge from j < i with probability min(0.30, 3 / i)
    "cfg5": dict(cls="DVAE", max_n=64, nvt=18, hs=1024, nz=56, bidir=True, B=2048, kind="enas", layers=62, ntypes=16,
                 sparse=True, n_batches=2),
}


def make_graphs(cfg, count, seed):
    if cfg["kind"] == "enas":
        p_rule = (lambda i: min(0.30, 3.0 / i) if i > 0 else 0.0) if cfg.get("sparse") else None
        rows = dvae_amd.data.synth_enas_rows(count, cfg["layers"], cfg["ntypes"], 0.30, seed, p_rule)
        return dvae_amd.data.enas_rows_to_arrays(rows, cfg["max_n"])
    rows = dvae_amd.data.synth_bn_rows(count, cfg["layers"], 0.25, seed)
    return dvae_amd.data.bn_rows_to_arrays(rows, cfg["max_n"])


def make_model(cfg, seed=1, mfma_dtype="fp32"):
    torch.manual_seed(seed)
    if cfg["cls"] == "DVAE":
        return dvae_amd.DVAE(cfg["max_n"], cfg["nvt"], 0, 1, cfg["hs"], cfg["nz"], cfg["bidir"], mfma_dtype=mfma_dtype)
    return dvae_amd.DVAE_BN(cfg["max_n"], cfg["nvt"], 0, 1, cfg["hs"], cfg["nz"], cfg["bidir"], mfma_dtype=mfma_dtype)


def f_alg_macs(cfg, ga):
    """De-duplicated forward MACs per DAG (SURVEY.md section 8d formula)."""
    n, T, H, nz = cfg["max_n"], cfg["nvt"], cfg["hs"], cfg["nz"]
    dirs = 2 if cfg["bidir"] else 1
    _, adj, _ = ga.to_dense()
    E = adj.sum() / len(ga)
    od0 = adj[:, 0, :].sum() / len(ga)
    gru = (dirs * n + n + E) * 3 * H * H
    if cfg["cls"] == "DVAE_BN":
        gm = 0.0
        edge = (2 * n - 1 + E - od0) * H * H + n * (n - 1) / 2 * H
    else:
        gm = (dirs + 1) * (n - 1) * 2 * H * H
        edge = ((n - 1) + (n - 1 + E - od0)) * 4 * H * H + n * (n - 1) / 2 * 4 * H
    vert = (n - 1) * (2 * H * H + 2 * H * T)
    head = 3 * H * nz + (2 * H * H if cfg["bidir"] else 0)
    return gru + gm + vert + edge + head, float(E), float(od0)


def time_dominant_gemm(cfg, batch, iters=30, bf16=False):
    """HIP-event timing of the fp32 MFMA GEMM on the largest product of the step:
    the edge predictor's first layer over all distinct 'current vertex' states
    (EB = ST[rowsB] W1b^T: M = NB rows, N = HEp, K = Hp)."""
    lib = dvae_amd._lib.load()
    Hp = (cfg["hs"] + 31) // 32 * 32
    HEp = Hp if cfg["cls"] == "DVAE_BN" else 4 * Hp
    M = int(batch.counts.NB)
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    A = torch.randn(M, Hp, device=dev, generator=g)
    W = torch.randn(HEp, Hp, device=dev, generator=g) * 0.05
    Cm = torch.empty(M, HEp, device=dev)
    d = dvae_amd._lib.GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), W.data_ptr(), Cm.data_ptr()
    d.lda, d.ldb, d.ldc = Hp, Hp, HEp
    d.M, d.N, d.K = M, HEp, Hp
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    fn = lib.dvae_gemm_bf16 if bf16 else lib.dvae_gemm_f32
    for _ in range(3):
        dvae_amd._lib.check(fn(C.byref(d), sp))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters):
        fn(C.byref(d), sp)
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * M * HEp * Hp
    return dict(kernel="gemm_bf16_kernel<128,128> (v_mfma_f32_32x32x16_bf16)" if bf16 else
                "gemm_f32_kernel<128,128> (v_mfma_f32_32x32x2_f32)", shape=[M, HEp, Hp],
                avg_ms=ms, achieved=flops / (ms * 1e-3) / 1e12)


def cpu_baseline(cfg, budget_s=20.0):
    """Oracle 'refstyle' port (per-vertex / per-graph loops mirroring the reference
    op for op) + Adam, on the host cores.  Bounded sample: B = 32 graphs of the same
    workload, as many steps as fit the budget (>= 2)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dvae_oracle as O
    threads = max(1, min(16, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    Bs = 32
    ga = make_graphs(cfg, Bs, 123)
    types, adj, n_g = (torch.from_numpy(x) for x in ga.to_dense())
    ocfg = O.make_cfg(cfg["max_n"], cfg["nvt"], 0, 1, cfg["hs"], cfg["nz"], cfg["bidir"], True, cfg["cls"] == "DVAE_BN")
    model = make_model(cfg)
    params = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    opt = torch.optim.Adam(list(params.values()), lr=1e-4)
    eps = torch.randn(Bs, cfg["nz"])

    def step():
        opt.zero_grad()
        mu, lv = O.encode(ocfg, params, types, adj, n_g, mode="refstyle")
        loss, _, _ = O.loss(ocfg, params, mu, lv, types, adj, n_g, eps, mode="refstyle")
        loss.backward()
        opt.step()

    step()  # warm-up
    t0 = time.perf_counter()
    steps = 0
    while steps < 2 or (time.perf_counter() - t0) < budget_s:
        step()
        steps += 1
        if steps >= 200:
            break
    dt = time.perf_counter() - t0
    return dict(value=Bs * steps / dt, unit="DAGs/s", cores=threads, kind="port",
                sample="%d train steps of %d synthetic %s DAGs (oracle refstyle port + Adam), %.1f s"
                       % (steps, Bs, cfg["kind"], dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="matrix-unit input type: f32 = the parity mode (headline); bf16 = reduced precision, "
                         "reported separately")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ..." % (args.gpus, args.gpus))
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    # one rank per GPU.  (Rehearsal on a 1-GPU box: DVAE_BENCH_BACKEND=gloo lets several ranks
    # share device 0; the driver's runs use the default, nccl = RCCL over xGMI.)
    backend = os.environ.get("DVAE_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= ndev:
        raise SystemExit("rank %d has no GPU (%d visible)" % (local_rank, ndev))
    local_dev = local_rank % ndev
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)

    bf16 = args.dtype == "bf16"
    peak = PEAK_BF16_TFLOPS if bf16 else PEAK_F32_TFLOPS
    model = make_model(cfg, mfma_dtype="bf16" if bf16 else "fp32").to(dev)
    model.train()
    # flat gradient buffer (its slices are the parameters' .grad; backward adds into them in place) and
    # Adam as one launch over the flat parameter buffer: the update rule of train.py:174
    reducer = dvae_amd.dist.GradSumReducer(model)
    opt = dvae_amd.optim.FlatAdam(model, lr=1e-4, reducer=reducer)

    # pre-packed batches: weak scaling, every rank owns B graphs per step
    B = int(os.environ.get("DVAE_BENCH_B", cfg["B"]))  # diagnostic override (batch-size sweeps); not a bench line
    n_batches = cfg.get("n_batches", 4)
    graphs = [make_graphs(cfg, B, seed=1000 * rank + i) for i in range(n_batches)]  # synthetic data: not timed
    model._collate_fn(graphs[0])  # first call: library load, allocator warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    batches = [model._collate_fn(ga) for ga in graphs]  # host packer (C++) + the index blob's copy to the GPU
    torch.cuda.synchronize()
    pack_ms = (time.perf_counter() - t0) * 1e3 / n_batches
    gen = torch.Generator(device=dev).manual_seed(7 + rank)
    eps = [torch.randn(B, cfg["nz"], device=dev, generator=gen) for _ in range(n_batches)]

    def step(i):
        b = batches[i % n_batches]
        reducer.zero()
        mu, logvar = model.encode(b)
        loss, _, _ = model.loss(mu, logvar, b, eps=eps[i % n_batches])
        loss.backward()
        reducer.all_reduce()
        opt.step()
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = step(i)
    host_dt = time.perf_counter() - t0  # host time to ENQUEUE the steps (the GPU runs behind it)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss_val = float(last.item())

    if rank == 0:
        macs, E, od0 = f_alg_macs(cfg, batches[0].arrays)
        train_flop_per_dag = 6.0 * macs
        dags_per_s = world * B * args.steps / dt
        roof = time_dominant_gemm(cfg, batches[0], bf16=bf16)
        out = {
            "metric": "DAGs/sec train step (ENAS6, hs=501)" if args.config in ("cfg1", "cfg2") else
                      "DAGs/sec train step (%s)" % args.config,
            "value": dags_per_s, "unit": "DAGs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s: %s%s hs=%d nz=%d on synthetic %s DAGs (n=%d, nvt=%d), %d DAGs per GPU per step, "
                                   "full train step incl. Adam%s%s" % (
                                       args.config, cfg["cls"], " bidirectional" if cfg["bidir"] else "", cfg["hs"],
                                       cfg["nz"], "ENAS-%d" % cfg["layers"] if cfg["kind"] == "enas" else "asia-shape BN",
                                       cfg["max_n"], cfg["nvt"], B, " + RCCL grad all-reduce" if world > 1 else "",
                                       "; bf16 MFMA inputs, fp32 accumulate / weights / activations" if bf16 else ""),
                       "global_batch": world * B, "edges_per_dag": E, "outdeg_v0": od0,
                       "train_gflop_per_dag": train_flop_per_dag / 1e9, "grad_allreduce_mb": reducer.nbytes() / 1e6,
                       "pack_ms_per_batch": pack_ms, "host_enqueue_ms_per_step": host_dt / args.steps * 1e3,
                       "loss_last_step": loss_val},
            "roofline": {"bound": "mfma", "achieved": roof["achieved"], "peak": peak, "unit": "TFLOP/s",
                         "frac": roof["achieved"] / peak,
                         "traffic": None if bf16 else MEASURED_TRAFFIC.get(tuple(roof["shape"])), "kernel": roof["kernel"],
                         "shape_MNK": roof["shape"], "avg_launch_ms": roof["avg_ms"]},
            "step_roofline": {"achieved": train_flop_per_dag * dags_per_s / world / 1e12, "peak": peak,
                              "unit": "TFLOP/s per GPU",
                              "frac": train_flop_per_dag * dags_per_s / world / 1e12 / peak},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
