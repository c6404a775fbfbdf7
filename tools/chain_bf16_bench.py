"""Micro-benchmark of the bf16 level kernels (csrc/chain.hip: chain_bf16_kernel) against the product + pointwise
launches they replace.  usage: python tools/chain_bf16_bench.py [B] [Hp]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _dvae_pkg import dvae_amd  # noqa: E402

L = dvae_amd._lib
if os.environ.get("DVAE_EXP_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["DVAE_EXP_LIB"])


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    Hp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    lib = L.load()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(0)
    T = 8
    hprev = torch.randn(B, Hp, generator=g).cuda()
    whh = (torch.randn(3 * Hp, Hp, generator=g) / Hp ** 0.5).cuda()
    wgm = (torch.randn(2 * Hp, Hp, generator=g) / Hp ** 0.5).cuda()
    whh_h, wgm_h = whh.bfloat16(), wgm.bfloat16()
    whh_t, wgm_t = whh.t().contiguous().bfloat16(), wgm.t().contiguous().bfloat16()
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
    off = torch.arange(0, 2 * B + 1, 2, dtype=torch.int32).cuda()
    idx = torch.stack([torch.arange(B), 2 * B + torch.arange(B)], 1).reshape(-1).int().cuda()
    dgm = torch.randn(B, 2 * Hp, generator=g).cuda()
    dh = torch.zeros(B, Hp).cuda()
    dgh = torch.randn(B, 3 * Hp, generator=g).cuda()
    dgin = torch.zeros(B, Hp).cuda()
    dagg = torch.zeros(B, Hp).cuda()

    def gemm(A, W, Cm, M, N, K, acc=0):
        d = L.GemmDesc()
        d.A, d.B, d.C = A.data_ptr(), W.data_ptr(), Cm.data_ptr()
        d.lda, d.ldb, d.ldc = A.stride(0), W.stride(0), Cm.stride(0)
        d.M, d.N, d.K = M, N, K
        d.b_bf16, d.accumulate = 1, acc
        L.check(lib.dvae_gemm_bf16(C.byref(d), s))

    def level(op, fill):
        d = L.LevelDesc()
        fill(d)
        return lambda: L.check(lib.dvae_chain_level_batch(op, C.byref(d), 1, B, Hp, s))

    def f_gru(d):
        d.a, d.lda, d.w, d.w_bf16, d.ldw_bf16 = hprev.data_ptr(), Hp, whh.data_ptr(), whh_h.data_ptr(), Hp
        d.gi_tab, d.types, d.bhn = gi.data_ptr(), types.data_ptr(), bhn.data_ptr()
        d.gates, d.ghn, d.hout, d.ld_hout = gates.data_ptr(), ghn.data_ptr(), hout.data_ptr(), Hp

    def f_gate(d, nxt=True):
        d.a, d.lda, d.w, d.w_bf16, d.ldw_bf16 = hout.data_ptr(), Hp, wgm.data_ptr(), wgm_h.data_ptr(), Hp
        d.gv_row, d.gm, d.msg, d.ld_msg, d.cur_row0 = gv.data_ptr(), gm.data_ptr(), msg.data_ptr(), Hp, 2 * B
        if nxt:
            d.next_agg, d.ld_agg, d.next_off, d.idx = agg.data_ptr(), Hp, off.data_ptr(), idx.data_ptr()

    def f_gbwd(d):
        d.a, d.lda, d.K, d.w, d.w_bf16, d.ldw_bf16 = dgm.data_ptr(), 2 * Hp, 2 * Hp, wgm.data_ptr(), wgm_t.data_ptr(), 2 * Hp
        d.dx, d.ld_dx = dh.data_ptr(), Hp
        d.gates, d.ghn, d.hprev, d.ld_hprev = gates.data_ptr(), ghn.data_ptr(), hprev.data_ptr(), Hp
        d.dgh, d.dgin, d.dhprev, d.ld_dhprev = dgh.data_ptr(), dgin.data_ptr(), dagg.data_ptr(), Hp

    def f_acc(d, gather=False):
        d.a, d.lda, d.K, d.w, d.w_bf16, d.ldw_bf16 = dgh.data_ptr(), 3 * Hp, 3 * Hp, whh.data_ptr(), whh_t.data_ptr(), 3 * Hp
        d.dx, d.ld_dx = dagg.data_ptr(), Hp
        if gather:
            d.gather_src, d.next_off, d.idx, d.cur_row0 = msg.data_ptr(), off.data_ptr(), idx.data_ptr(), 2 * B
            d.gm, d.dgm_next = gm.data_ptr(), dgm.data_ptr()

    def u_gru():
        gemm(hprev, whh_h, gates, B, 3 * Hp, Hp)
        L.check(lib.dvae_gru_pointwise_fwd(gates.data_ptr(), gi.data_ptr(), types.data_ptr(), bhn.data_ptr(),
                                           hprev.data_ptr(), Hp, hout.data_ptr(), Hp, ghn.data_ptr(), B, Hp, s))

    def u_gate():
        gemm(hout, wgm_h, gm, B, 2 * Hp, Hp)
        L.check(lib.dvae_gated_fwd_gather(gm.data_ptr(), gv.data_ptr(), msg.data_ptr(), Hp, 2 * B, agg.data_ptr(), Hp,
                                          off.data_ptr(), idx.data_ptr(), B, Hp, s))

    def u_gbwd():
        gemm(dgm, wgm_t, dh, B, Hp, 2 * Hp, acc=1)
        L.check(lib.dvae_gru_pointwise_bwd(dh.data_ptr(), Hp, gates.data_ptr(), ghn.data_ptr(), hprev.data_ptr(), Hp,
                                           dgh.data_ptr(), dgin.data_ptr(), dagg.data_ptr(), Hp, B, Hp, s))

    def u_acc():
        gemm(dgh, whh_t, dagg, B, Hp, 3 * Hp, acc=1)

    def timeit(fn, reps=100):
        for _ in range(10):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    rows = [("GRU fwd: product + pointwise", u_gru), ("GRU fwd: bf16 level kernel", level(0, f_gru)),
            ("gate fwd: product + gather", u_gate), ("gate fwd: bf16 level kernel", level(1, f_gate)),
            ("gate fwd: bf16 level kernel, no gather", level(1, lambda d: f_gate(d, False))),
            ("GRU bwd: product + pointwise", u_gbwd), ("GRU bwd: bf16 level kernel", level(3, f_gbwd)),
            ("dgrad acc: product", u_acc), ("dgrad acc: bf16 level kernel", level(2, f_acc)),
            ("dgrad acc: bf16 level kernel + gather tail", level(2, lambda d: f_acc(d, True)))]
    for name, fn in rows:
        dh.zero_(); dagg.zero_()
        print("%-46s %8.2f us" % (name, timeit(fn)), flush=True)


if __name__ == "__main__":
    main()
