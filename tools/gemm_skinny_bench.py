"""The products of a batch-32 train step (cfg 1: DVAE hs=501, 32 ENAS-6 graphs) on gemm_f32_kernel's tiles and on the
small-M kernel (csrc/gemm_skinny.hip) with 1 / 2 / 4 blocks per tile; HIP events, 50 launches each.
usage: python tools/gemm_skinny_bench.py            """
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DVAE_GEMM_SKINNY_MMAC"] = "0"  # dvae_gemm_f32 = gemm_f32_kernel throughout this tool
from _dvae_pkg import dvae_amd  # noqa: E402

L = dvae_amd._lib
lib = L.load()

SHAPES = [
    # (label, M, N, K, b_kmajor, accumulate)
    ("fc1|fc2            ", 32, 128, 512, 0, 0),
    ("fc3                ", 32, 512, 64, 0, 0),
    ("partial-row GRU    ", 157, 1536, 512, 0, 0),
    ("EA / EB            ", 298, 2048, 512, 0, 0),
    ("add_vertex layer 1 ", 224, 1024, 512, 0, 0),
    ("add_vertex layer 2 ", 224, 32, 1024, 0, 0),
    ("d add_vertex 2     ", 224, 1024, 32, 1, 0),
    ("d EA / d EB        ", 298, 512, 2048, 1, 1),
    ("d add_vertex 1     ", 224, 512, 1024, 1, 1),
    ("d partial-row GRU  ", 157, 512, 1536, 1, 1),
    ("d fc3              ", 32, 64, 512, 1, 0),
    ("d fc1|fc2          ", 32, 512, 128, 1, 1),
    ("cfg 2 latent rows  ", 1024, 128, 512, 0, 0),
    ("batch 64 EA / EB   ", 596, 2048, 512, 0, 0),
    ("batch 64 d EA      ", 596, 512, 2048, 1, 1),
]


def time_it(fn, iters=50):
    st = torch.cuda.current_stream()
    for _ in range(5):
        L.check(fn())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def run(label, M, N, K, bkm, acc):
    g = torch.Generator(device="cuda").manual_seed(0)
    A = torch.randn(M, K, device="cuda", generator=g)
    B = torch.randn((K, N) if bkm else (N, K), device="cuda", generator=g)
    Cm = torch.zeros(M, N, device="cuda")
    d = L.GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cm.data_ptr()
    d.lda, d.ldb, d.ldc = K, B.stride(0), N
    d.M, d.N, d.K = M, N, K
    d.b_kmajor, d.accumulate = bkm, acc
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = ["%s M=%4d N=%5d K=%5d %s" % (label, M, N, K, "K-major B" if bkm else "         ")]
    out.append("tiled %6.1f us" % time_it(lambda: lib.dvae_gemm_f32(C.byref(d), sp)))
    for nt in (1, 2, 4):
        if bkm and nt == 2:
            continue
        out.append("16x%-2d %6.1f us" % (16 * nt, time_it(lambda: lib.dvae_gemm_skinny_f32(C.byref(d), nt, sp))))
    out.append("auto %6.1f us" % time_it(lambda: lib.dvae_gemm_skinny_f32(C.byref(d), 0, sp)))
    print("   ".join(out))


if __name__ == "__main__":
    print("floor for reference: an empty-ish launch is ~2-3 us back to back on one stream")
    for s in SHAPES:
        run(*s)
