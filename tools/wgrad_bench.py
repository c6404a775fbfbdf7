"""Weight-gradient products of cfg 5 on dvae_gemm_bf16 (two K-major operands, optional row gather along K):
checks the result against torch on bf16-rounded operands and times it.  DVAE_GEMM_TILE / DVAE_GEMM_SPLIT pick the
kernel (tools/ab.sh style: run the script once per setting)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _dvae_pkg import dvae_amd

L = dvae_amd._lib
lib = L.load()

SHAPES = [
    # M, N, K (rows contracted), rows of the gathered source (0: no gather)
    (4096, 1024, 131072, 0),
    (4096, 1024, 131072, 40000),
    (3072, 1024, 129143, 0),
    (2048, 1024, 129024, 0),
    (4000, 1000, 65536 + 37, 0),   # ragged everywhere
]


def run(M, N, K, src_rows, iters=5):
    g = torch.Generator(device="cuda").manual_seed(0)
    A = torch.randn(K, M, device="cuda", generator=g)
    Bsrc = torch.randn(src_rows if src_rows else K, N, device="cuda", generator=g)
    rows = torch.randint(0, src_rows, (K,), device="cuda", dtype=torch.int32, generator=g) if src_rows else None
    C0 = torch.randn(M, N, device="cuda", generator=g)
    Cm = C0.clone()
    d = L.GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), Bsrc.data_ptr(), Cm.data_ptr()
    d.lda, d.ldb, d.ldc = M, N, N
    d.M, d.N, d.K = M, N, K
    d.a_kmajor, d.b_kmajor, d.accumulate = 1, 1, 1
    if rows is not None:
        d.b_rows = rows.data_ptr()
        d.b_rows_bound = src_rows
    st = torch.cuda.current_stream()
    sp = C.c_void_p(st.cuda_stream)
    L.check(lib.dvae_gemm_bf16(C.byref(d), sp))
    torch.cuda.synchronize()
    Bm = Bsrc[rows.long()] if rows is not None else Bsrc
    ref = C0.double()
    step = 16384
    for k in range(0, K, step):
        ref += A[k:k + step].bfloat16().double().t() @ Bm[k:k + step].bfloat16().double()
    err = ((Cm.double() - ref).norm() / ref.norm()).item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        lib.dvae_gemm_bf16(C.byref(d), sp)
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("wgrad M=%5d N=%5d K=%7d gather=%d  %9.1f us  %6.1f TF/s  rel err %.2e %s" % (
        M, N, K, int(rows is not None), ms * 1e3, 2.0 * M * N * K / ms / 1e9, err, "OK" if err < 1e-5 else "MISMATCH"),
        flush=True)
    return err < 1e-5


if __name__ == "__main__":
    ok = True
    for s in SHAPES:
        ok = run(*s) and ok
    sys.exit(0 if ok else 1)
