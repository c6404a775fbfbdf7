"""Micro-benchmark of dvae_gemm_f32 on the shapes of the train step (HIP events)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _dvae_pkg import dvae_amd

L = dvae_amd._lib
if os.environ.get("DVAE_EXP_LIB"):  # experiments: an alternative build of the library
    L.LIB_PATH = os.path.abspath(os.environ["DVAE_EXP_LIB"])
lib = L.load()

SHAPES = [
    # (label, M, N, K, a_kmajor, b_kmajor, accumulate)
    ("chain fwd GRU   ", 1024, 1536, 512, 0, 0, 0),
    ("chain fwd gate  ", 1024, 1024, 512, 0, 0, 0),
    ("chain bwd dGRU  ", 1024, 512, 1536, 0, 1, 1),
    ("chain bwd dgate ", 1024, 512, 1024, 0, 1, 1),
    ("EB fwd          ", 9251, 2048, 512, 0, 0, 0),
    ("EA fwd          ", 7168, 2048, 512, 0, 0, 0),
    ("V1 fwd          ", 7168, 1024, 512, 0, 0, 0),
    ("dEB dgrad       ", 9251, 512, 2048, 0, 1, 1),
    ("wgrad W1b       ", 2048, 512, 9251, 1, 1, 1),
    ("wgrad Whh       ", 1536, 512, 8192, 1, 1, 1),
    ("wgrad Wgm       ", 1024, 512, 8192, 1, 1, 1),
    ("square 4096     ", 4096, 4096, 4096, 0, 0, 0),
    ("square 8192     ", 8192, 8192, 8192, 0, 0, 0),
    ("cfg5 GRU level  ", 2048, 3072, 1024, 0, 0, 0),
    ("cfg5 gate level ", 2048, 2048, 1024, 0, 0, 0),
    ("cfg5 dGRU as NN ", 2048, 1024, 3072, 0, 0, 1),
    ("cfg5 dgate as NN", 2048, 1024, 2048, 0, 0, 1),
    ("cfg5 EB fwd     ", 457494, 4096, 1024, 0, 0, 0),
    ("cfg5 dEB as NN  ", 457494, 1024, 4096, 0, 0, 1),
]


BF16 = os.environ.get("DVAE_GEMM_BF16") == "1"
STORE = os.environ.get("DVAE_GEMM_STORE", "f32xf32")  # with DVAE_GEMM_BF16=1: how A x B are stored (K-contiguous shapes only)
PEAK = 2500.0 if BF16 else 157.3


def run(label, M, N, K, akm, bkm, acc, iters=20):
    g = torch.Generator(device="cuda").manual_seed(0)
    A = torch.randn((K, M) if akm else (M, K), device="cuda", generator=g)
    B = torch.randn((K, N) if bkm else (N, K), device="cuda", generator=g)
    Cm = torch.zeros(M, N, device="cuda")
    d = L.GemmDesc()
    if BF16 and not akm and not bkm:
        sa, sb = STORE.split("x")
        A = A.bfloat16() if sa == "bf16" else A
        B = B.bfloat16() if sb == "bf16" else B
        d.a_bf16, d.b_bf16 = int(sa == "bf16"), int(sb == "bf16")
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cm.data_ptr()
    d.lda, d.ldb, d.ldc = A.stride(0), B.stride(0), N
    d.M, d.N, d.K = M, N, K
    d.a_kmajor, d.b_kmajor, d.accumulate = akm, bkm, acc
    fn = lib.dvae_gemm_bf16 if BF16 else lib.dvae_gemm_f32
    st = torch.cuda.current_stream()
    sp = C.c_void_p(st.cuda_stream)
    for _ in range(3):
        L.check(fn(C.byref(d), sp))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn(C.byref(d), sp)
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * M * N * K / ms / 1e9
    yard = ""
    if os.environ.get("DVAE_GEMM_YARDSTICK") == "1" and not BF16:
        # the vendor library on the same product (torch.mm -> hipBLASLt / rocBLAS), as a yardstick only
        a2 = A.t() if akm else A
        b2 = B if bkm else B.t()
        out = torch.empty(M, N, device="cuda")
        for _ in range(3):
            torch.mm(a2, b2, out=out)
        e0.record(st)
        for _ in range(iters):
            torch.mm(a2, b2, out=out)
        e1.record(st)
        torch.cuda.synchronize()
        yard = "   | torch.mm %8.1f us" % (e0.elapsed_time(e1) / iters * 1e3)
    print("%s M=%5d N=%5d K=%5d  %8.1f us  %6.1f TF/s  (%4.1f%% of %.1f)%s" % (label, M, N, K, ms * 1e3, tf, tf / PEAK * 100, PEAK, yard))


if __name__ == "__main__":
    for s in SHAPES:
        run(*s)
