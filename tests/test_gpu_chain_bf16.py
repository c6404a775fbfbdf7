"""chain.hip, chain_bf16_kernel: the four level operations with bf16 matrix-unit inputs (a K-contiguous bf16 copy of
the weight in dvae_level_desc.w_bf16, the fp32 operand rounded on its way to the matrix unit) against the exact-fp32
level kernels run on operands that were rounded to bf16 beforehand: both then contract the same values with fp32
accumulation and differ by the summation order over K only (2e-5 absolute on O(1) values, as the fp32 tests)."""
import ctypes as C

import pytest
import torch

from _dvae_pkg import dvae_amd

L = dvae_amd._lib
pytestmark = pytest.mark.gpu

GRU_FWD, GATED_FWD, DGRAD_ACC, GRU_BWD = 0, 1, 2, 3


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _r(x):
    """round to bf16, keep fp32 storage"""
    return x.bfloat16().float()


def _run(op, d, rows, Hp):
    L.check(L.load().dvae_chain_level_batch(op, C.byref(d), 1, rows, Hp, _s()), "chain level")
    torch.cuda.synchronize()


def _nan(*shape):
    return torch.full(shape, float("nan"), device="cuda")


@pytest.mark.parametrize("rows,Hp", [(300, 128), (2048, 1024), (5, 64)])
def test_gru_level_fwd_bf16(rows, Hp):
    g = torch.Generator().manual_seed(rows + Hp)
    T = 8
    hprev = torch.randn(rows, Hp, generator=g).cuda()
    whh = (torch.randn(3 * Hp, Hp, generator=g) / Hp ** 0.5).cuda()
    gi = torch.randn(T, 3 * Hp, generator=g).cuda()
    bhn = torch.randn(Hp, generator=g).cuda()
    types = torch.randint(0, T, (rows,), generator=g).int().cuda()

    def desc(a, w, outs):
        d = L.LevelDesc()
        d.a, d.lda, d.w = a.data_ptr(), Hp, w.data_ptr()
        d.gi_tab, d.types, d.bhn = gi.data_ptr(), types.data_ptr(), bhn.data_ptr()
        d.gates, d.ghn, d.hout, d.ld_hout = outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), Hp
        return d
    ref = [_nan(rows, 3 * Hp), _nan(rows, Hp), _nan(rows, Hp)]
    hr, wr = _r(hprev), _r(whh)
    _run(GRU_FWD, desc(hr, wr, ref), rows, Hp)
    out = [_nan(rows, 3 * Hp), _nan(rows, Hp), _nan(rows, Hp)]
    wh = whh.bfloat16()
    d = desc(hprev, whh, out)
    d.w_bf16, d.ldw_bf16 = wh.data_ptr(), Hp
    _run(GRU_FWD, d, rows, Hp)
    for a, b in zip(out[:2], ref[:2]):  # gates (r, z, n) and W_hn h + b_hn
        assert torch.isfinite(a).all() and (a - b).abs().max() < 2e-5
    # the new state mixes the UNROUNDED previous state (the tail reads fp32): h' = (1 - z) n + z h
    z, n_ = out[0][:, Hp:2 * Hp], out[0][:, 2 * Hp:]
    assert (out[2] - ((1 - z) * n_ + z * hprev)).abs().max() < 1e-6


@pytest.mark.parametrize("B,Hp,with_next", [(300, 128, True), (2048, 1024, True), (70, 64, False)])
def test_gated_level_fwd_bf16(B, Hp, with_next):
    g = torch.Generator().manual_seed(B + Hp)
    levels, cur = 3, 1
    msg = torch.randn(levels * B, Hp, generator=g).cuda()
    h = torch.randn(B, Hp, generator=g).cuda()
    wgm = (torch.randn(2 * Hp, Hp, generator=g) / Hp ** 0.5).cuda()
    gv = torch.randn(levels, 2 * Hp, generator=g).cuda()
    keep = torch.rand(B, cur + 1, generator=g) < 0.6
    off, idx = [0], []
    for b in range(B):
        idx += [u * B + b for u in range(cur + 1) if keep[b, u]]
        off.append(len(idx))
    off_d = torch.tensor(off, dtype=torch.int32).cuda()
    idx_d = torch.tensor(idx + [0], dtype=torch.int32).cuda()

    def desc(a, w, gm, m, agg):
        d = L.LevelDesc()
        d.a, d.lda, d.w = a.data_ptr(), Hp, w.data_ptr()
        d.gv_row, d.gm, d.msg, d.ld_msg, d.cur_row0 = gv[cur].data_ptr(), gm.data_ptr(), m.data_ptr(), Hp, cur * B
        if with_next:
            d.next_agg, d.ld_agg, d.next_off, d.idx = agg.data_ptr(), Hp, off_d.data_ptr(), idx_d.data_ptr()
        return d
    gm_ref, msg_ref, agg_ref = _nan(B, 2 * Hp), msg.clone(), _nan(B, Hp)
    hr, wr = _r(h), _r(wgm)
    _run(GATED_FWD, desc(hr, wr, gm_ref, msg_ref, agg_ref), B, Hp)
    gm, msg_f, agg = _nan(B, 2 * Hp), msg.clone(), _nan(B, Hp)
    wh = wgm.bfloat16()
    d = desc(h, wgm, gm, msg_f, agg)
    d.w_bf16, d.ldw_bf16 = wh.data_ptr(), Hp
    _run(GATED_FWD, d, B, Hp)
    assert torch.isfinite(gm).all() and (gm - gm_ref).abs().max() < 2e-5
    assert (msg_f - msg_ref).abs().max() < 2e-5
    assert torch.equal(msg_f[:cur * B], msg[:cur * B]) and torch.equal(msg_f[(cur + 1) * B:], msg[(cur + 1) * B:])
    if with_next:
        assert (agg - agg_ref).abs().max() < 5e-5


@pytest.mark.parametrize("rows,Hp,K", [(300, 128, 384), (2048, 1024, 3072), (9, 64, 128)])
def test_level_dgrad_acc_bf16(rows, Hp, K):
    g = torch.Generator().manual_seed(rows + Hp + K)
    dy = torch.randn(rows, K, generator=g).cuda()
    w = (torch.randn(K, Hp, generator=g) / K ** 0.5).cuda()
    dx0 = torch.randn(rows, Hp, generator=g).cuda()
    dx = dx0.clone()
    wt = w.t().contiguous().bfloat16()  # [Hp][K]: the transposed copy pack.hip keeps
    d = L.LevelDesc()
    d.a, d.lda, d.K, d.w, d.dx, d.ld_dx = dy.data_ptr(), K, K, w.data_ptr(), dx.data_ptr(), Hp
    d.w_bf16, d.ldw_bf16 = wt.data_ptr(), K
    _run(DGRAD_ACC, d, rows, Hp)
    ref = dx0.double() + _r(dy).double() @ _r(w).double()
    assert torch.isfinite(dx).all() and (dx.double() - ref).abs().max() < 2e-5


@pytest.mark.parametrize("rows,Hp", [(300, 128), (2048, 1024)])
def test_gru_level_bwd_bf16(rows, Hp):
    g = torch.Generator().manual_seed(rows * 3 + Hp)
    K = 2 * Hp
    dgm = torch.randn(rows, K, generator=g).cuda()
    w = (torch.randn(K, Hp, generator=g) / K ** 0.5).cuda()
    dh0 = torch.randn(rows, Hp, generator=g).cuda()
    gates = torch.rand(rows, 3 * Hp, generator=g).cuda()
    gates[:, 2 * Hp:] = gates[:, 2 * Hp:] * 2 - 1
    ghn = torch.randn(rows, Hp, generator=g).cuda()
    hprev = torch.randn(rows, Hp, generator=g).cuda()

    def desc(a, wf, dh, outs):
        d = L.LevelDesc()
        d.a, d.lda, d.K, d.w, d.dx, d.ld_dx = a.data_ptr(), K, K, wf.data_ptr(), dh.data_ptr(), Hp
        d.gates, d.ghn, d.hprev, d.ld_hprev = gates.data_ptr(), ghn.data_ptr(), hprev.data_ptr(), Hp
        d.dgh, d.dgin, d.dhprev, d.ld_dhprev = outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), Hp
        return d
    dh_ref, outs_ref = dh0.clone(), [_nan(rows, 3 * Hp), _nan(rows, Hp), _nan(rows, Hp)]
    ar, wr = _r(dgm), _r(w)
    _run(GRU_BWD, desc(ar, wr, dh_ref, outs_ref), rows, Hp)
    dh, outs = dh0.clone(), [_nan(rows, 3 * Hp), _nan(rows, Hp), _nan(rows, Hp)]
    wt = w.t().contiguous().bfloat16()
    d = desc(dgm, w, dh, outs)
    d.w_bf16, d.ldw_bf16 = wt.data_ptr(), K
    _run(GRU_BWD, d, rows, Hp)
    assert (dh - dh_ref).abs().max() < 2e-5
    for a, b in zip(outs, outs_ref):
        assert torch.isfinite(a).all() and (a - b).abs().max() < 5e-5


def test_bf16_level_rules():
    Hp, rows = 96, 10  # Hp % 64 != 0
    a = torch.zeros(rows, Hp, device="cuda")
    w = torch.zeros(3 * Hp, Hp, device="cuda")
    wh = w.bfloat16()
    outs = [torch.zeros(rows, 3 * Hp, device="cuda"), torch.zeros(rows, Hp, device="cuda"), torch.zeros(rows, Hp, device="cuda")]
    d = L.LevelDesc()
    d.a, d.lda, d.w = a.data_ptr(), Hp, w.data_ptr()
    gi = torch.zeros(4, 3 * Hp, device="cuda")
    types = torch.zeros(rows, dtype=torch.int32, device="cuda")
    bhn = torch.zeros(Hp, device="cuda")
    d.gi_tab, d.types, d.bhn = gi.data_ptr(), types.data_ptr(), bhn.data_ptr()
    d.gates, d.ghn, d.hout, d.ld_hout = outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), Hp
    d.w_bf16, d.ldw_bf16 = wh.data_ptr(), Hp
    with pytest.raises(L.DvaeHipError, match="multiples of 64"):
        _run(GRU_FWD, d, rows, Hp)
