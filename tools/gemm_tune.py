"""Tile / split-K tuning of dvae_gemm_f32 on the products one train step actually issues.

  python tools/gemm_tune.py log   [cfg]        -> gpurun_out/gemm_shapes_<cfg>.txt  (shape, count per step)
  python tools/gemm_tune.py sweep [cfg]        -> gpurun_out/gemm_tune_<cfg>.txt    (time per shape x tile x split)

`log` runs one step with DVAE_GEMM_LOG=1 in a child process and counts the distinct products;
`sweep` times each of them under DVAE_GEMM_TILE / DVAE_GEMM_SPLIT overrides, one child process per
override pair (the knobs are read once per process).  DVAE_GEMM_BF16=1 tunes the bf16-input kernel."""
import collections
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def child_step(cfg):
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from _dvae_pkg import dvae_amd  # noqa: F401
    c = bench.CONFIGS[cfg]
    model = bench.make_model(c, 1, "bf16" if os.environ.get("DVAE_GEMM_BF16") == "1" else "fp32").cuda()
    ga = bench.make_graphs(c, c["B"], 0)
    batch = model._collate_fn(ga)
    model.train()
    mu, logvar = model.encode(batch)
    loss, _, _ = model.loss(mu, logvar, batch)
    loss.backward()
    torch.cuda.synchronize()


def child_time(shapes_file):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    import gemm_bench as g
    import ctypes as C
    L, lib = g.L, g.lib
    for line in open(shapes_file):
        f = line.split()
        if not f:
            continue
        cnt = int(f[0])
        M, N, K = int(f[1]), int(f[2]), int(f[3])
        kv = dict(x.split("=") for x in f[4:])
        akm, bkm, acc = int(kv["akm"]), int(kv["bkm"]), int(kv["acc"])
        gen = torch.Generator(device="cuda").manual_seed(0)
        ka = K if not int(kv["br"]) else K  # gathered K rows come from a matrix of the same height here
        A = torch.randn((K, M) if akm else (M, K), device="cuda", generator=gen)
        B = torch.randn((ka, N) if bkm else (N, K), device="cuda", generator=gen)
        Cm = torch.zeros(M, N, device="cuda")
        bias = torch.randn(N, device="cuda", generator=gen) if int(kv["bias"]) else None
        d = L.GemmDesc()
        d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cm.data_ptr()
        d.lda, d.ldb, d.ldc = A.stride(0), B.stride(0), N
        d.M, d.N, d.K = M, N, K
        d.a_kmajor, d.b_kmajor, d.accumulate, d.act = akm, bkm, acc, int(kv["act"])
        keep = []
        if bias is not None:
            d.bias = bias.data_ptr()
        if int(kv["ar"]) and not akm:
            r = torch.randperm(M, device="cuda", generator=gen).int(); keep.append(r); d.a_rows = r.data_ptr()
        if int(kv["br"]) and bkm:
            r = torch.randperm(K, device="cuda", generator=gen).int(); keep.append(r); d.b_rows = r.data_ptr(); d.b_rows_bound = K
        if int(kv["cr"]):
            r = torch.randperm(M, device="cuda", generator=gen).int(); keep.append(r); d.c_rows = r.data_ptr()
        st = torch.cuda.current_stream()
        sp = C.c_void_p(st.cuda_stream)
        fn = lib.dvae_gemm_bf16 if os.environ.get("DVAE_GEMM_BF16") == "1" else lib.dvae_gemm_f32
        for _ in range(3):
            L.check(fn(C.byref(d), sp))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(20):
            fn(C.byref(d), sp)
        e1.record(st)
        torch.cuda.synchronize()
        print("T %s | %.2f" % (" ".join(f[1:]), e0.elapsed_time(e1) / 20 * 1e3), flush=True)


def main():
    mode = sys.argv[1]
    cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
    os.makedirs(OUT, exist_ok=True)
    shapes_file = os.path.join(OUT, "gemm_shapes_%s.txt" % cfg)
    if mode == "_step":
        return child_step(cfg)
    if mode == "_time":
        return child_time(shapes_file)
    if mode == "log":
        env = dict(os.environ, DVAE_GEMM_LOG="1")
        r = subprocess.run([sys.executable, __file__, "_step", cfg], env=env, stderr=subprocess.PIPE, text=True, timeout=300)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-3000:])
            sys.exit(1)
        cnt = collections.Counter(l[5:].strip() for l in r.stderr.splitlines() if l.startswith("GEMM "))
        with open(shapes_file, "w") as f:
            for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
                f.write("%d %s\n" % (v, k))
        print("%d products, %d distinct -> %s" % (sum(cnt.values()), len(cnt), shapes_file))
        return
    if mode == "sweep":
        res = collections.defaultdict(dict)
        combos = [(0, 0)] + [(t, s) for t in (128, 64, 32) for s in (1, 2, 3, 4, 6, 8, 12, 16)]
        for tile, split in combos:
            env = dict(os.environ)
            if tile:
                env["DVAE_GEMM_TILE"], env["DVAE_GEMM_SPLIT"] = str(tile), str(split)
            r = subprocess.run([sys.executable, __file__, "_time", cfg], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               text=True, timeout=300)
            if r.returncode != 0:
                sys.stderr.write(r.stderr[-3000:])
                sys.exit(1)
            for l in r.stdout.splitlines():
                if l.startswith("T "):
                    k, v = l[2:].split(" | ")
                    res[k][(tile, split)] = float(v)
            print("done tile=%d split=%d" % (tile, split), flush=True)
        counts = {" ".join(l.split()[1:]): int(l.split()[0]) for l in open(shapes_file) if l.strip()}
        with open(os.path.join(OUT, "gemm_tune_%s.txt" % cfg), "w") as f:
            tot_now = tot_best = 0.0
            for k, r in res.items():
                best = min(r.items(), key=lambda kv: kv[1])
                now = r[(0, 0)]
                tot_now += now * counts[k]
                tot_best += best[1] * counts[k]
                f.write("%3d x %s\n      now %.1f us | best %.1f us at tile=%d split=%d\n      %s\n" % (
                    counts[k], k, now, best[1], best[0][0], best[0][1],
                    "  ".join("%d/%d:%.1f" % (t, s, v) for (t, s), v in sorted(r.items()) if t)))
            f.write("sum of kernel times per step: now %.1f us, best %.1f us\n" % (tot_now, tot_best))
        print(open(os.path.join(OUT, "gemm_tune_%s.txt" % cfg)).read())


if __name__ == "__main__":
    main()
