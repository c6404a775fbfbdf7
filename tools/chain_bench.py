"""Micro-benchmark of one forward chain level: fused level kernels (csrc/chain.hip) against the
GEMM + pointwise launches they replace (HIP events on the current stream).
usage: python tools/chain_bench.py [B] [Hp]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _dvae_pkg import dvae_amd  # noqa: E402

L = dvae_amd._lib
if os.environ.get("DVAE_EXP_LIB"):  # A/B runs against an alternative build of the library
    L.LIB_PATH = os.path.abspath(os.environ["DVAE_EXP_LIB"])


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    Hp = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    lib = L.load()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(0)
    T = 8
    hprev = torch.randn(B, Hp, generator=g).cuda()
    whh = (torch.randn(3 * Hp, Hp, generator=g) / Hp ** 0.5).cuda()
    wgm = (torch.randn(2 * Hp, Hp, generator=g) / Hp ** 0.5).cuda()
    gi = torch.randn(T, 3 * Hp, generator=g).cuda()
    bhn = torch.randn(Hp, generator=g).cuda()
    gv = torch.randn(2 * Hp, generator=g).cuda()
    types = torch.randint(0, T, (B,), generator=g).int().cuda()
    gates = torch.zeros(B, 3 * Hp).cuda()
    ghn = torch.zeros(B, Hp).cuda()
    hout = torch.zeros(B, Hp).cuda()
    gm = torch.zeros(B, 2 * Hp).cuda()
    msg = torch.randn(3 * B, Hp, generator=g).cuda()
    agg = torch.zeros(B, Hp).cuda()
    off = torch.arange(0, 2 * B + 1, 2, dtype=torch.int32).cuda()       # two predecessors per row:
    idx = torch.stack([torch.arange(B), 2 * B + torch.arange(B)], 1).reshape(-1).int().cuda()  # level 0 and own

    def gemm(A, W, Cm, M, N, K):
        d = L.GemmDesc()
        d.A, d.B, d.C = A.data_ptr(), W.data_ptr(), Cm.data_ptr()
        d.lda, d.ldb, d.ldc = A.stride(0), W.stride(0), Cm.stride(0)
        d.M, d.N, d.K = M, N, K
        L.check(lib.dvae_gemm_f32(C.byref(d), s))

    def unfused():
        gemm(hprev, whh, gates, B, 3 * Hp, Hp)
        L.check(lib.dvae_gru_pointwise_fwd(gates.data_ptr(), gi.data_ptr(), types.data_ptr(), bhn.data_ptr(),
                                           hprev.data_ptr(), Hp, hout.data_ptr(), Hp, ghn.data_ptr(), B, Hp, s))
        gemm(hout, wgm, gm, B, 2 * Hp, Hp)
        L.check(lib.dvae_gated_fwd_gather(gm.data_ptr(), gv.data_ptr(), msg.data_ptr(), Hp, 2 * B, agg.data_ptr(), Hp,
                                          off.data_ptr(), idx.data_ptr(), B, Hp, s))

    def gru_fused():
        L.check(lib.dvae_gru_level_fwd(hprev.data_ptr(), Hp, whh.data_ptr(), gi.data_ptr(), types.data_ptr(),
                                       bhn.data_ptr(), gates.data_ptr(), ghn.data_ptr(), hout.data_ptr(), Hp, B, Hp, s))

    def gate_fused():
        L.check(lib.dvae_gated_level_fwd(hout.data_ptr(), Hp, wgm.data_ptr(), gv.data_ptr(), gm.data_ptr(),
                                         msg.data_ptr(), Hp, 2 * B, agg.data_ptr(), Hp, off.data_ptr(), idx.data_ptr(),
                                         B, Hp, s))

    def fused():
        gru_fused()
        gate_fused()

    # backward level: d h += d gm [Wg|Wm]; GRU pointwise backward; d agg += d gh W_hh
    dgm = torch.randn(B, 2 * Hp, generator=g).cuda()
    dh = torch.zeros(B, Hp).cuda()
    dgh = torch.zeros(B, 3 * Hp).cuda()
    dgin = torch.zeros(B, Hp).cuda()
    dagg = torch.zeros(B, Hp).cuda()

    def dgrad(dY, W, dX, M, N, K):
        d = L.GemmDesc()
        d.A, d.B, d.C = dY.data_ptr(), W.data_ptr(), dX.data_ptr()
        d.lda, d.ldb, d.ldc = dY.stride(0), W.stride(0), dX.stride(0)
        d.M, d.N, d.K = M, K, N
        d.b_kmajor, d.accumulate = 1, 1
        L.check(lib.dvae_gemm_f32(C.byref(d), s))

    def bwd_unfused():
        dgrad(dgm, wgm, dh, B, 2 * Hp, Hp)
        L.check(lib.dvae_gru_pointwise_bwd(dh.data_ptr(), Hp, gates.data_ptr(), ghn.data_ptr(), hprev.data_ptr(), Hp,
                                           dgh.data_ptr(), dgin.data_ptr(), dagg.data_ptr(), Hp, B, Hp, s))
        dgrad(dgh, whh, dagg, B, 3 * Hp, Hp)

    def bwd_gru():
        L.check(lib.dvae_gru_level_bwd(dgm.data_ptr(), 2 * Hp, 2 * Hp, wgm.data_ptr(), dh.data_ptr(), Hp,
                                       gates.data_ptr(), ghn.data_ptr(), hprev.data_ptr(), Hp, dgh.data_ptr(),
                                       dgin.data_ptr(), dagg.data_ptr(), Hp, B, Hp, s))

    def bwd_acc():
        L.check(lib.dvae_level_dgrad_acc(dgh.data_ptr(), 3 * Hp, 3 * Hp, whh.data_ptr(), dagg.data_ptr(), Hp, B, Hp, s))

    def bwd_fused():
        bwd_gru()
        bwd_acc()

    def timeit(fn, reps=200):
        for _ in range(20):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    flops = 2.0 * B * Hp * 5 * Hp
    for name, fn in (("unfused level (4 launches)", unfused), ("fused level (2 launches)", fused),
                     ("  GRU level kernel", gru_fused), ("  gate level kernel", gate_fused)):
        us = timeit(fn)
        print(f"{name:32s} {us:8.2f} us" + (f"  {flops / us * 1e-6:6.1f} TFLOP/s" if "level (" in name else ""), flush=True)
    for name, fn in (("unfused bwd level (3 launches)", bwd_unfused), ("fused bwd level (2 launches)", bwd_fused),
                     ("  dgrad + GRU bwd kernel", bwd_gru), ("  dgrad accumulate kernel", bwd_acc)):
        dh.zero_(); dagg.zero_()
        us = timeit(fn)
        print(f"{name:32s} {us:8.2f} us" + (f"  {flops / us * 1e-6:6.1f} TFLOP/s" if "level (" in name else ""), flush=True)


if __name__ == "__main__":
    main()
