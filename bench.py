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
  roofline      the dominant kernel family of the step timed live with HIP events on the stream
                it runs on: the fused GRU level kernel of the state chains (chain_kernel, the
                largest share of the step's kernel time at cfg 2 / cfg 4) where the step runs
                it, otherwise the MFMA GEMM on the step's largest product
  roofline_gemm the fp32 / bf16 MFMA GEMM on the largest product of the step (always)
  step_roofline whole-step algorithmic FLOPs (6 * F_alg) / step time
  cpu_baseline  the oracle's reference-style CPU port timed on the host cores
                (rank 0, N = 1 only, bounded sample)
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------------------------
# Self-launch.  `python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset)
# starts the N ranks itself: one child process per GPU with RANK / LOCAL_RANK / WORLD_SIZE /
# MASTER_ADDR / MASTER_PORT set, exactly what torch.distributed.run would set.  The parent never
# touches the GPU (it runs before torch is imported), relays rank 0's ONE JSON line and exits with
# the first failing rank's code.  Under torch.distributed.run (WORLD_SIZE set) nothing of this runs.
# ----------------------------------------------------------------------------------------------
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="matrix-unit input type: f32 = the parity mode (headline); bf16 = reduced precision, "
                         "reported separately")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the config's per-GPU batch on every rank; strong: a fixed global batch "
                         "(--global-batch, default 8 x the per-GPU batch: cfg 4 = 8192) ceil-split over the ranks")
    ap.add_argument("--global-batch", type=int, default=0)
    ap.add_argument("--allreduce", default="overlap", choices=["overlap", "after"],
                    help="N > 1: overlap = the decoder-only part of the gradient buffer is all-reduced from inside "
                         "backward(), beside the encoder's backward (default); after = one all-reduce after backward()")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the train step from one captured HIP graph over shape-stable (padded) batches "
                         "(harness.GraphedTrainStep).  auto = off: measured on cfg 1 (profiles/r03_graph_replay_cfg1.txt) a "
                         "replay costs the host 0.08 ms instead of 0.73 ms per step but the GPU 0.86 instead of 0.82 ms, "
                         "and the GPU is what bounds the step on this box; 'on' is for hosts that cannot keep up")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args(argv)


def launch_plan(argv, environ, port=None):
    """[] when this process is itself a rank (single GPU, or a launcher set WORLD_SIZE); otherwise one
    (command, environment) per rank to start."""
    args = parse_args(argv)
    if args.gpus <= 1 or "WORLD_SIZE" in environ:
        return []
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    plan = []
    for r in range(args.gpus):
        env = dict(environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
        env.setdefault("OMP_NUM_THREADS", "4")
        plan.append(([sys.executable, os.path.abspath(__file__)] + list(argv), env))
    return plan


def self_launch(plan):
    procs = []
    for r, (cmd, env) in enumerate(plan):
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    rc, pending = 0, list(range(len(procs)))
    while pending:
        time.sleep(0.05)
        for r in list(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.remove(r)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:  # one rank failed: the others would wait in a collective for ever
                    procs[q].terminate()
    out0 = procs[0].stdout.read()  # rank 0 prints one line: far below the pipe's capacity, so it never blocked
    lines = [l for l in out0.splitlines() if l.startswith("{")]
    if rc == 0 and lines:
        print(lines[-1], flush=True)
    elif rc == 0:
        rc = 1
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
    return rc


if __name__ == "__main__":
    _plan = launch_plan(sys.argv[1:], os.environ)
    if _plan:
        sys.exit(self_launch(_plan))

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, ROOT)
from _dvae_pkg import dvae_amd  # noqa: E402

PEAK_F32_TFLOPS = 157.3    # MI355X fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak
PROFILES = os.path.join(ROOT, "profiles")


def newest_profile(suffix):
    """newest round's profiles/rNN_<suffix>, or None"""
    try:
        names = sorted(f for f in os.listdir(PROFILES) if f.endswith("_" + suffix) and f[0] == "r" and f[1:3].isdigit())
    except OSError:
        return None
    return os.path.join(PROFILES, names[-1]) if names else None


def measured_traffic(shape):
    """Memory-side bytes of ONE launch of the dominant GEMM, from the committed rocprofv3 PMC passes on that
    launch (tools/profile_round.sh -> profiles/rNN_gemm_pmc.json): FETCH_SIZE (KB) doubled per the gfx950
    correction of MI355X_MICROARCH.md (section HBM) + WRITE_SIZE (KB).  Infinity-Cache hits are counted by
    these counters: traffic behind L2, an upper bound on HBM bytes.  None if the passes were taken on
    another shape (or are absent)."""
    path = newest_profile("gemm_pmc.json")
    if path is None:
        return None, None
    pm = json.load(open(path))
    if list(pm.get("shape_MNK", [9251, 2048, 512])) != list(shape) or "FETCH_SIZE" not in pm or "WRITE_SIZE" not in pm:
        return None, os.path.relpath(path, ROOT)
    return (2.0 * pm["FETCH_SIZE"]["mean"] + pm["WRITE_SIZE"]["mean"]) * 1024.0, os.path.relpath(path, ROOT)

def measured_chain_traffic(shape):
    """Memory-side bytes of ONE launch of the GRU level kernel from the committed PMC passes
    (tools/pmc_chain.sh -> profiles/rNN_chain_pmc.json), as measured_traffic(); None when absent or another shape."""
    path = newest_profile("chain_pmc.json")
    if path is None:
        return None, None
    pm = json.load(open(path))
    if list(pm.get("shape_MNK", [])) != list(shape) or "FETCH_SIZE" not in pm or "WRITE_SIZE" not in pm:
        return None, os.path.relpath(path, ROOT)
    return (2.0 * pm["FETCH_SIZE"]["mean"] + pm["WRITE_SIZE"]["mean"]) * 1024.0, os.path.relpath(path, ROOT)


CONFIGS = {
    # name: (class, ctor kwargs, per-GPU batch, generator)
    "cfg1": dict(cls="DVAE", max_n=8, nvt=8, hs=501, nz=56, bidir=False, B=32, kind="enas", layers=6, ntypes=6),
    "cfg2": dict(cls="DVAE", max_n=8, nvt=8, hs=501, nz=56, bidir=True, B=1024, kind="enas", layers=6, ntypes=6),
    "cfg3": dict(cls="DVAE_BN", max_n=10, nvt=10, hs=501, nz=56, bidir=False, B=4096, kind="bn", layers=8, ntypes=8),
    # (strong scaling, SURVEY 8d/e: fixed global batch 8192 = 8 x 1024)
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


def executed_mfma_macs(cfg, c):
    """Forward MACs the step actually ISSUES on the matrix units for one batch, from the packed batch's
    counts `c`: feature dimensions padded (hs 501 -> 512, nz 56 -> 64, nvt -> 32), states without predecessors
    de-duplicated to one row per type (PT, NB), the first encoder level (h_prev = 0) skipped.  Every product
    has its two gradient products in the backward pass, so the train step issues 3 x this (x 2 FLOP/MAC).
    Row padding to whole 64 / 128-row tiles is not counted (< 1 %)."""
    n, B = cfg["max_n"], int(c.B)
    Hp = (cfg["hs"] + 31) // 32 * 32
    NZp = (cfg["nz"] + 31) // 32 * 32
    Tp, HVp = 32, 2 * Hp
    bn = cfg["cls"] == "DVAE_BN"
    HEp = Hp if bn else 4 * Hp
    dirs = 2 if cfg["bidir"] else 1
    nB, PT, NA, NB, R = int(c.nB), int(c.PT), int(c.NA), int(c.NB), int(c.R)
    if bn:
        enc = dirs * nB * 3 * Hp * Hp                          # no recurrence: one batched hidden product per direction
        dec = (nB + PT) * 3 * Hp * Hp + B * HEp * Hp           # + the H0 block of the edge predictor
    else:
        enc = dirs * (n - 1) * B * 5 * Hp * Hp                 # GRU (3Hp) + gate|mapper (2Hp) per level
        dec = n * B * 3 * Hp * Hp + (n - 1) * B * 2 * Hp * Hp + PT * 3 * Hp * Hp
    head = B * 2 * NZp * Hp + B * Hp * NZp + (B * 2 * Hp * Hp if cfg["bidir"] else 0)
    edge = (NA + NB) * HEp * Hp
    vert = R * HVp * Hp + R * Tp * HVp
    return float(enc + dec + head + edge + vert)


def roofline_by_kernel():
    """Per-kernel-family share of the step and fraction of the MFMA peak, from the newest committed rocprofv3
    kernel-trace summary of this workload (tools/roofline_by_kernel.py writes profiles/rNN_roofline_by_kernel.json
    from profiles/rNN_bench_cfg2_kernel_stats.csv).  A profile of an earlier run of the same command, not a
    measurement of this run: `source` names the file."""
    path = newest_profile("roofline_by_kernel.json")
    if path is None:
        return None
    d = json.load(open(path))
    d["source"] = os.path.relpath(path, ROOT)
    return d


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
    if bf16:  # as in the step: fp32 activations x the bf16 copy of the weight (csrc/orch.h: use_bf16_weight_copy)
        Wh = W.bfloat16()
        d.B, d.b_bf16 = Wh.data_ptr(), 1
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
    big = ((M + 255) // 256) * ((HEp + 255) // 256) >= 192  # csrc/gemm_bf16_nn.hip: 256 x 256 tiles from 3/4 of a round
    return dict(kernel=("gemm_bf16_nn_kernel<%s> fp32 x bf16-stored (v_mfma_f32_32x32x16_bf16)" % ("256,256" if big else "128,128")) if bf16 else
                "gemm_f32_kernel<128,128> (v_mfma_f32_32x32x2_f32)", shape=[M, HEp, Hp],
                avg_ms=ms, achieved=flops / (ms * 1e-3) / 1e12)


def chain_level_is_fused(cfg, B, bf16):
    """does the step run its chain levels on chain_kernel (csrc/train.hip: chain_level_small)?  fp32 mode, DVAE,
    levels of half a round to two rounds of the chip in 64 x 32 tiles"""
    if bf16 or cfg["cls"] == "DVAE_BN":
        return False
    Hp = (cfg["hs"] + 31) // 32 * 32
    tiles = ((B + 63) // 64) * (Hp // 32)
    return 128 <= tiles <= 512


def time_chain_level(cfg, B, iters=200):
    """HIP-event timing of the dominant kernel family of the fp32 step at cfg 2 / cfg 4: one GRU LEVEL of a state
    chain (csrc/chain.hip chain_kernel<3, false, 0, ...>: hidden product h_prev W_hh^T [B x Hp] x [Hp x 3Hp] and the
    GRUCell tail in one launch), on B rows, through the C-ABI entry point the step itself calls
    (dvae_gru_level_fwd).  Algorithmic FLOPs per launch = 2 * B * Hp * 3Hp (the product; the tail is not counted)."""
    lib = dvae_amd._lib.load()
    Hp = (cfg["hs"] + 31) // 32 * 32
    T = cfg["nvt"]
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    hprev = torch.randn(B, Hp, device=dev, generator=g)
    whh = torch.randn(3 * Hp, Hp, device=dev, generator=g) / Hp ** 0.5
    gi = torch.randn(32, 3 * Hp, device=dev, generator=g)
    bhn = torch.randn(Hp, device=dev, generator=g)
    types = torch.randint(0, T, (B,), device=dev, generator=g, dtype=torch.int32)
    gates = torch.empty(B, 3 * Hp, device=dev)
    ghn = torch.empty(B, Hp, device=dev)
    hout = torch.empty(B, Hp, device=dev)
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)

    def launch():
        return lib.dvae_gru_level_fwd(hprev.data_ptr(), Hp, whh.data_ptr(), gi.data_ptr(), types.data_ptr(),
                                      bhn.data_ptr(), gates.data_ptr(), ghn.data_ptr(), hout.data_ptr(), Hp, B, Hp, sp)
    for _ in range(10):
        dvae_amd._lib.check(launch())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters):
        launch()
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * B * Hp * 3 * Hp
    return dict(kernel="chain_kernel<3, false, 0, 4, ...> GRU level: h_prev W_hh^T + GRUCell tail (v_mfma_f32_32x32x2_f32)",
                shape=[B, 3 * Hp, Hp], avg_ms=ms, achieved=flops / (ms * 1e-3) / 1e12)


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
    args = parse_args()
    cfg = CONFIGS[args.config]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
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
    reducer.overlap = args.allreduce == "overlap"
    B = int(os.environ.get("DVAE_BENCH_B", cfg["B"]))  # diagnostic override (batch-size sweeps); not a bench line
    use_graph = world == 1 and os.environ.get("DVAE_FORCE_COLLECTIVE") != "1" and \
        args.graph == "on"
    opt = dvae_amd.optim.FlatAdam(model, lr=1e-4, reducer=reducer, capturable=use_graph)

    # pre-packed batches.  weak: every rank owns B graphs per step (its own synthetic draw);
    # strong: every rank owns its contiguous ceil-split chunk (util.py:566-572) of the same global batch.
    # Either way eps is the rank's row range of ONE global normal draw (N-rank parity rule).
    n_batches = cfg.get("n_batches", 4)
    strong = args.scaling == "strong"
    if strong:
        Bg = args.global_batch or 8 * B
        lo, hi = dvae_amd.batching.shard_bounds(Bg, world, rank)
        graphs = [make_graphs(cfg, Bg, seed=5000 + i)[lo:hi] for i in range(n_batches)]  # synthetic data: not timed
    else:
        Bg = world * B
        lo, hi = rank * B, (rank + 1) * B
        graphs = [make_graphs(cfg, B, seed=1000 * rank + i) for i in range(n_batches)]
    B_rank = hi - lo
    if B_rank <= 0:
        raise SystemExit("rank %d has no graphs (global batch %d over %d ranks)" % (rank, Bg, world))
    caps = None
    if use_graph:
        # shape-stable packing: caps from the statistics of the data the batches are drawn from (here: all of them)
        allg = graphs[0]
        for ga in graphs[1:]:
            allg = dvae_amd.GraphArrays(np.concatenate([allg.types, ga.types]), np.concatenate([allg.pred_mask, ga.pred_mask]),
                                        np.concatenate([allg.n_vertices, ga.n_vertices]), allg.max_n)
        caps = dvae_amd.batching.padding_caps(allg, B_rank, cfg["nvt"])
        use_graph = caps is not None
    graphed = dvae_amd.harness.GraphedTrainStep(model, opt) if use_graph else None
    model._collate_fn(graphs[0], caps)  # first call: library load, allocator warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    batches = [model._collate_fn(ga, caps) for ga in graphs]  # host packer (C++) + the index blob's copy to the GPU
    torch.cuda.synchronize()
    pack_ms = (time.perf_counter() - t0) * 1e3 / n_batches
    eps = []
    for i in range(n_batches):
        gen = torch.Generator().manual_seed(7 + i)
        eps.append(torch.randn(Bg, cfg["nz"], generator=gen)[lo:hi].contiguous().to(dev))

    stream = torch.cuda.current_stream()
    exchange = world > 1 or os.environ.get("DVAE_FORCE_COLLECTIVE") == "1"
    ar_events = []   # per timed step: (decoder gradients final, backward done, all-reduce done) on the compute stream

    def step(i, timed=False, events=None):
        b = batches[i % n_batches]
        if graphed is not None:
            return graphed(b, eps=eps[i % n_batches])[0]
        reducer.zero()
        mu, logvar = model.encode(b)
        loss, _, _ = model.loss(mu, logvar, b, eps=eps[i % n_batches])
        sink = events if events is not None else (ar_events if (exchange and timed) else None)
        if sink is not None:
            reducer.mark_decoder_done = marks = []
            loss.backward()
            reducer.mark_decoder_done = None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            reducer.all_reduce()
            e1.record(stream)
            sink.append((marks[0] if marks else None, e0, e1))
        else:
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
    launches0 = dvae_amd._lib.load().dvae_debug_launch_count()
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = step(i, True)
    host_dt = time.perf_counter() - t0  # host time to ENQUEUE the steps (the GPU runs behind it)
    launches = (dvae_amd._lib.load().dvae_debug_launch_count() - launches0) / float(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    # the exchange step as the compute stream sees it (the part not hidden behind the encoder's backward), and the
    # span from the point where the decoder's gradients are final -- where the early part is launched in overlap
    # mode -- to the end of the last all-reduce
    ar_ms = sum(a.elapsed_time(b) for _, a, b in ar_events) / len(ar_events) if ar_events else 0.0
    ar_span = [m.elapsed_time(b) for m, _, b in ar_events if m is not None]
    ar_span_ms = sum(ar_span) / len(ar_span) if ar_span else 0.0
    loss_global = float(reducer.reduce_scalars(last)[0].item())  # sum over the ranks' shards of the last global batch

    # host cost of ONE step with an empty queue in front of it: enqueue one step right after a device sync, read the
    # host clock before any further sync (median of 9).  Unlike host_enqueue_ms_per_step (K steps back to back, which
    # includes the time the host spends blocked on a full queue) this is the host's own time per step.
    unblocked = []
    for i in range(9):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step(i)
        unblocked.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize()
    unblocked.sort()
    host_unblocked_ms = unblocked[len(unblocked) // 2]

    # encoder backward (decoder gradients final -> backward() done) with and without the early all-reduce in flight
    enc_bwd = {}
    if exchange:
        for mode in ("overlap", "after"):
            reducer.overlap = mode == "overlap"
            evs = []
            for i in range(6):
                step(i, events=evs)
            torch.cuda.synchronize()
            v = [m.elapsed_time(a) for m, a, _ in evs[1:] if m is not None]
            enc_bwd[mode] = sum(v) / len(v) if v else None
        reducer.overlap = args.allreduce == "overlap"
    barrier()

    if rank == 0:
        macs, E, od0 = f_alg_macs(cfg, batches[0].arrays)
        train_flop_per_dag = 6.0 * macs
        dags_per_s = Bg * args.steps / dt
        exec_flop_per_step = 6.0 * executed_mfma_macs(cfg, batches[0].counts)
        roof_gemm = time_dominant_gemm(cfg, batches[0], bf16=bf16)
        traffic, traffic_src = (None, None) if bf16 else measured_traffic(roof_gemm["shape"])
        gemm_line = {"bound": "mfma", "achieved": roof_gemm["achieved"], "peak": peak, "unit": "TFLOP/s",
                     "frac": roof_gemm["achieved"] / peak, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": roof_gemm["kernel"], "shape_MNK": roof_gemm["shape"], "avg_launch_ms": roof_gemm["avg_ms"]}
        if chain_level_is_fused(cfg, B_rank, bf16):
            # the dominant family of this step (largest share of kernel time in profiles/rNN_roofline_by_kernel.json)
            roof = time_chain_level(cfg, B_rank)
            ctraffic, ctraffic_src = measured_chain_traffic(roof["shape"])
            roof_line = {"bound": "mfma", "achieved": roof["achieved"], "peak": peak, "unit": "TFLOP/s",
                         "frac": roof["achieved"] / peak, "traffic": ctraffic, "traffic_source": ctraffic_src,
                         "kernel": roof["kernel"], "shape_MNK": roof["shape"], "avg_launch_ms": roof["avg_ms"],
                         "flop_per_launch": 2.0 * roof["shape"][0] * roof["shape"][1] * roof["shape"][2]}
        else:
            roof_line = gemm_line
        out = {
            "metric": "DAGs/sec train step (ENAS6, hs=501)" if args.config in ("cfg1", "cfg2") else
                      "DAGs/sec train step (%s)" % args.config,
            "value": dags_per_s, "unit": "DAGs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s: %s%s hs=%d nz=%d on synthetic %s DAGs (n=%d, nvt=%d), %s, "
                                   "full train step incl. Adam%s%s" % (
                                       args.config, cfg["cls"], " bidirectional" if cfg["bidir"] else "", cfg["hs"],
                                       cfg["nz"], "ENAS-%d" % cfg["layers"] if cfg["kind"] == "enas" else "asia-shape BN",
                                       cfg["max_n"], cfg["nvt"],
                                       ("global batch %d split over the GPUs" % Bg) if strong else
                                       ("%d DAGs per GPU per step" % B),
                                       " + RCCL grad all-reduce" if world > 1 else "",
                                       "; bf16 MFMA inputs, fp32 accumulate / weights / activations" if bf16 else ""),
                       "global_batch": Bg, "dags_per_gpu": B_rank, "edges_per_dag": E, "outdeg_v0": od0,
                       "train_gflop_per_dag": train_flop_per_dag / 1e9, "grad_allreduce_mb": reducer.nbytes() / 1e6,
                       "rccl_ranks": dist.get_world_size() if (world > 1 and backend == "nccl") else 0,
                       "collective_backend": backend if world > 1 else "none",
                       "allreduce": args.allreduce if exchange else "none",
                       "allreduce_ms_per_step": ar_ms,          # on the compute stream, after backward(): the exposed part
                       "allreduce_total_ms": ar_span_ms,        # decoder gradients final -> last all-reduce done
                       "allreduce_early_mb": (reducer.n_early * 4 / 1e6) if (exchange and reducer.overlap) else 0.0,
                       "encode_bwd_ms": enc_bwd,                # {"overlap": with the early part in flight, "after": without}
                       "pack_ms_per_batch": pack_ms, "host_enqueue_ms_per_step": host_dt / args.steps * 1e3,
                       "host_enqueue_ms_per_step_unblocked": host_unblocked_ms,
                       "launches_per_step": launches,  # the library's kernel launches ENQUEUED BY THE HOST per step (its memsets / copies and torch's own are not counted; replays of a captured step enqueue none)
                       "graph": ({"replays": graphed.replays, "eager_steps": graphed.eager_steps,
                                  "padded_batches": sum(1 for b in batches if b.padded), "batches": len(batches)}
                                 if graphed is not None else None),
                       "loss_last_step": loss_global},
            "roofline": roof_line,
            "roofline_gemm": gemm_line,
            # algorithmic rate (the SURVEY 8d contract figure: every DISTINCT module evaluation at dense cost, 6 F_alg
            # per DAG) and, beside it, the FLOPs the matrix units really execute per step (padded, de-duplicated):
            # `executed` is the step's MFMA occupancy, `achieved` credits work the restatement avoids
            "step_roofline": {"achieved": train_flop_per_dag * dags_per_s / world / 1e12, "peak": peak,
                              "unit": "TFLOP/s per GPU",
                              "frac": train_flop_per_dag * dags_per_s / world / 1e12 / peak,
                              "executed": {"gflop_per_step_per_gpu": exec_flop_per_step / 1e9,
                                           "achieved": exec_flop_per_step / (dt / args.steps) / 1e12,
                                           "frac": exec_flop_per_step / (dt / args.steps) / 1e12 / peak}},
        }
        if graphed is not None:
            out["config"]["workload"] += "; step replayed from one captured HIP graph over shape-stable (padded) batches"
        by_kernel = roofline_by_kernel() if (args.config == "cfg2" and not bf16) else None
        if by_kernel is not None:
            out["roofline_by_kernel"] = by_kernel
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()  # rank 0 is still timing its roofline leg: the other ranks keep their communicators until it is done
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
