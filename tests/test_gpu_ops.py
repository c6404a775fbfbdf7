"""Each HIP kernel on its own, through the C-ABI, against plain PyTorch fp32/fp64
on the same inputs (tolerances stated per test)."""
import ctypes as C

import numpy as np
import pytest
import torch

from _dvae_pkg import dvae_amd

pytestmark = pytest.mark.gpu
L = dvae_amd._lib


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _gemm(A, B, Cm, M, N, K, akm=0, bkm=0, a_rows=None, b_rows=None, c_rows=None, bias=None, act=0, acc=0,
          bf16=False):
    d = L.GemmDesc()
    d.a_bf16, d.b_bf16 = int(A.dtype == torch.bfloat16), int(B.dtype == torch.bfloat16)
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cm.data_ptr()
    d.a_rows = a_rows.data_ptr() if a_rows is not None else None
    d.b_rows = b_rows.data_ptr() if b_rows is not None else None
    d.c_rows = c_rows.data_ptr() if c_rows is not None else None
    d.bias = bias.data_ptr() if bias is not None else None
    d.lda, d.ldb, d.ldc = A.stride(0), B.stride(0), Cm.stride(0)
    d.M, d.N, d.K = M, N, K
    d.a_kmajor, d.b_kmajor, d.act, d.accumulate = akm, bkm, act, acc
    d.b_rows_bound = B.shape[0] if (b_rows is not None and bkm) else 0
    fn = L.load().dvae_gemm_bf16 if bf16 else L.load().dvae_gemm_f32
    L.check(fn(C.byref(d), _s()), "gemm")
    torch.cuda.synchronize()


def _ref(a, b):
    return (a.double() @ b.double()).float()


@pytest.mark.parametrize("M,N,K", [(1024, 1536, 512), (100, 96, 64), (7, 32, 32), (4000, 2048, 512), (300, 40, 32)])
def test_gemm_nt_asymmetric(M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g)
    out = torch.full((M, N), float("nan"), device="cuda")
    _gemm(A, W, out, M, N, K)
    ref = _ref(A, W.t())
    assert (out - ref).abs().max() <= 2e-6 * K ** 0.5 * 16 + 1e-5 * ref.abs().max() * 0


def test_gemm_layout_identity_check():
    # A = I with an asymmetric B catches a transposed C write (guide section 3)
    K = 64
    A = torch.eye(K, device="cuda")
    W = torch.arange(96 * K, device="cuda", dtype=torch.float32).reshape(96, K)
    out = torch.zeros(K, 96, device="cuda")
    _gemm(A, W, out, K, 96, K)
    assert torch.equal(out, W.t().contiguous())


def test_gemm_bias_act_accumulate_gather_scatter():
    g = torch.Generator(device="cuda").manual_seed(5)
    rows_total, M, N, K = 500, 260, 160, 96
    X = torch.randn(rows_total, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g)
    bias = torch.randn(N, device="cuda", generator=g)
    a_rows = torch.randperm(rows_total, device="cuda", generator=g)[:M].int()
    c_rows = torch.randperm(400, device="cuda", generator=g)[:M].int()
    for act in (0, 1, 2):
        out = torch.ones(400, N, device="cuda")
        _gemm(X, W, out, M, N, K, a_rows=a_rows, c_rows=c_rows, bias=bias, act=act, acc=1)
        y = _ref(X[a_rows.long()], W.t()) + bias
        y = [y, torch.relu(y), torch.tanh(y)][act]
        ref = torch.ones(400, N, device="cuda")
        ref[c_rows.long()] += y
        assert (out - ref).abs().max() < 2e-4


def test_gemm_dgrad_kmajor_b():
    g = torch.Generator(device="cuda").manual_seed(6)
    M, N, K = 1000, 1536, 512  # dX[M,K] = dY[M,N] W[N,K]
    dY = torch.randn(M, N, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g)
    out = torch.zeros(M, K, device="cuda")
    _gemm(dY, W, out, M, K, N, akm=0, bkm=1)
    ref = _ref(dY, W)
    assert (out - ref).abs().max() < 1e-3


def _gemm_skinny(A, B, Cm, M, N, K, nt, bkm=0, a_rows=None, c_rows=None, bias=None, act=0, acc=0, mask=None):
    d = L.GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cm.data_ptr()
    d.a_rows = a_rows.data_ptr() if a_rows is not None else None
    d.c_rows = c_rows.data_ptr() if c_rows is not None else None
    d.bias = bias.data_ptr() if bias is not None else None
    d.relu_mask = mask.data_ptr() if mask is not None else None
    d.lda, d.ldb, d.ldc = A.stride(0), B.stride(0), Cm.stride(0)
    d.M, d.N, d.K = M, N, K
    d.b_kmajor, d.act, d.accumulate = bkm, act, acc
    L.check(L.load().dvae_gemm_skinny_f32(C.byref(d), nt, _s()), "gemm_skinny")
    torch.cuda.synchronize()


# the shapes of a batch-32 step (tools/gemm_skinny_bench.py) plus ragged ones: M not a multiple of 16, N a multiple of
# 4 only (partial blocks and tiles), K = 16 (one step: seven of the eight K slices idle), one row
_SKINNY = [(32, 128, 512), (224, 32, 1024), (298, 2048, 512), (157, 1536, 512), (32, 512, 64), (1, 16, 16), (37, 52, 48),
           (333, 100, 96), (16, 4, 16)]


@pytest.mark.parametrize("M,N,K", _SKINNY)
@pytest.mark.parametrize("bkm", [0, 1])
@pytest.mark.parametrize("nt", [0, 1, 2, 4])
def test_gemm_skinny_matches_fp64(M, N, K, bkm, nt):
    """csrc/gemm_skinny.hip against a float64 product: tolerance = fp32 rounding of a K-term sum, 16 eps sqrt(K) |a||b|"""
    if bkm and nt == 2:
        pytest.skip("a K-major B takes one block or four interleaved blocks per tile")
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K + bkm)
    A = torch.randn(M, K, device="cuda", generator=g)
    B = torch.randn((K, N) if bkm else (N, K), device="cuda", generator=g)
    out = torch.full((M, N), float("nan"), device="cuda")
    _gemm_skinny(A, B, out, M, N, K, nt, bkm=bkm)
    ref = _ref(A, B if bkm else B.t())
    assert (out - ref).abs().max() <= 2e-6 * K ** 0.5 * 16
    # the same order of additions whatever the tile width: bit-identical across nt
    if nt in (1, 4):
        out2 = torch.full((M, N), float("nan"), device="cuda")
        _gemm_skinny(A, B, out2, M, N, K, 4 if nt == 1 else 1, bkm=bkm)
        assert torch.equal(out, out2)


@pytest.mark.parametrize("bkm", [0, 1])
@pytest.mark.parametrize("nt", [1, 4])
def test_gemm_skinny_epilogue_gather_scatter_mask(bkm, nt):
    g = torch.Generator(device="cuda").manual_seed(11 + bkm)
    rows_total, M, N, K = 500, 260, 160, 96
    X = torch.randn(rows_total, K, device="cuda", generator=g)
    W = torch.randn((K, N) if bkm else (N, K), device="cuda", generator=g)
    Wt = W if bkm else W.t()
    bias = torch.randn(N, device="cuda", generator=g)
    a_rows = torch.randperm(rows_total, device="cuda", generator=g)[:M].int()
    c_rows = torch.randperm(400, device="cuda", generator=g)[:M].int()
    for act in (0, 1, 2):
        out = torch.ones(400, N, device="cuda")
        _gemm_skinny(X, W, out, M, N, K, nt, bkm=bkm, a_rows=a_rows, c_rows=c_rows, bias=bias, act=act, acc=1)
        y = _ref(X[a_rows.long()], Wt) + bias
        y = [y, torch.relu(y), torch.tanh(y)][act]
        ref = torch.ones(400, N, device="cuda")
        ref[c_rows.long()] += y
        assert (out - ref).abs().max() < 2e-4
    # ReLU-backward mask laid out like C (rows named by c_rows), no accumulate
    mask = torch.randn(400, N, device="cuda", generator=g)
    out = torch.full((400, N), 7.0, device="cuda")
    _gemm_skinny(X, W, out, M, N, K, nt, bkm=bkm, a_rows=a_rows, c_rows=c_rows, mask=mask)
    ref = torch.full((400, N), 7.0, device="cuda")
    ref[c_rows.long()] = torch.where(mask[c_rows.long()] > 0, _ref(X[a_rows.long()], Wt), torch.zeros((), device="cuda"))
    assert (out - ref).abs().max() < 2e-4


def test_gemm_skinny_layout_identity_and_refusals():
    K = 64
    A = torch.eye(K, device="cuda")
    W = torch.arange(96 * K, device="cuda", dtype=torch.float32).reshape(96, K)
    for nt in (1, 2, 4):
        out = torch.zeros(K, 96, device="cuda")
        _gemm_skinny(A, W, out, K, 96, K, nt)
        assert torch.equal(out, W.t().contiguous())
    Wk = W.t().contiguous()  # [K][96]: K-major
    for nt in (1, 4):
        out = torch.zeros(K, 96, device="cuda")
        _gemm_skinny(A, Wk, out, K, 96, K, nt, bkm=1)
        assert torch.equal(out, Wk)
    # dvae_gemm_f32 picks this kernel for few rows and gives the same bits as the forced call
    out1, out2 = torch.zeros(K, 96, device="cuda"), torch.zeros(K, 96, device="cuda")
    X = torch.randn(K, K, device="cuda")
    _gemm(X, W, out1, K, 96, K)
    _gemm_skinny(X, W, out2, K, 96, K, 0)
    assert torch.equal(out1, out2)
    d = L.GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), W.data_ptr(), out.data_ptr()
    d.lda, d.ldb, d.ldc, d.M, d.N, d.K = K, K, 96, K, 96, K
    assert L.load().dvae_gemm_skinny_f32(C.byref(d), 3, _s()) != 0  # nt must be 0, 1, 2 or 4
    d.a_kmajor = 1
    assert L.load().dvae_gemm_skinny_f32(C.byref(d), 1, _s()) != 0  # a K-major A is not served


@pytest.mark.parametrize("rows,N,K", [(8192, 1536, 512), (333, 64, 32), (5000, 32, 1536), (70, 128, 512)])
def test_gemm_wgrad_splitk(rows, N, K):
    g = torch.Generator(device="cuda").manual_seed(rows)
    dY = torch.randn(rows, N, device="cuda", generator=g)
    X = torch.randn(rows + 50, K, device="cuda", generator=g)
    b_rows = torch.randperm(rows + 50, device="cuda", generator=g)[:rows].int()
    out = torch.zeros(N, K, device="cuda")
    _gemm(dY, X, out, N, K, rows, akm=1, bkm=1, b_rows=b_rows, acc=1)
    ref = _ref(dY.t(), X[b_rows.long()])
    assert (out - ref).abs().max() <= 1e-5 * rows ** 0.5 * 4 + 1e-4


@pytest.fixture
def span_limit():
    """lowers the span a K-major operand may cover (normally 4 GiB) for one test"""
    lib = L.load()

    def set_limit(nbytes):
        L.check(lib.dvae_debug_set_kmajor_span_limit(int(nbytes)), "span limit")
    yield set_limit
    lib.dvae_debug_set_kmajor_span_limit(0)


@pytest.mark.parametrize("bf16", [False, True], ids=["f32", "bf16"])
@pytest.mark.parametrize("form", ["wgrad", "dgrad"])
def test_gemm_kmajor_operand_beyond_the_span_limit_runs_in_k_chunks(form, bf16, span_limit):
    """A contraction whose K-major operand spans more than 4 GiB (cfg 5: the 3072 x 1024 x 479 k weight
    gradient) runs as accumulating K chunks (gemm_k_chunk / gemm_next_chunk).  With the span limit lowered
    to 1 MB the same path runs at a test size: ragged last chunk, chunk offsets on both operands, a
    pre-loaded accumulator, and the same numbers as the unchunked product."""
    g = torch.Generator(device="cuda").manual_seed(77)
    if form == "wgrad":  # dW[N, K] += dY[rows, N]^T X[rows, K]: both operands K-major, K = rows
        rows, N, K = 2500, 160, 192
        A = torch.randn(rows, N, device="cuda", generator=g)
        Bm = torch.randn(rows, K, device="cuda", generator=g)
        args = dict(M=N, N=K, K=rows, akm=1, bkm=1, acc=1)
        ref = 0.25 + (A.double().t() @ Bm.double()).float()
        shape = (N, K)
    else:                # dX[M, K] = dY[M, N] W[N, K]: B K-major, contraction over N
        M, N, K = 300, 2208, 160
        A = torch.randn(M, N, device="cuda", generator=g)
        Bm = torch.randn(N, K, device="cuda", generator=g)
        args = dict(M=M, N=K, K=N, akm=0, bkm=1, acc=1)
        ref = 0.25 + (A.double() @ Bm.double()).float()
        shape = (M, K)
    if bf16:
        ra, rb = A.bfloat16().double(), Bm.bfloat16().double()
        ref = 0.25 + ((ra.t() @ rb) if form == "wgrad" else (ra @ rb)).float()
    whole = torch.full(shape, 0.25, device="cuda")
    _gemm(A, Bm, whole, bf16=bf16, **args)
    span_limit(1 << 20)  # 1 MB: rows of 640 / 768 bytes -> chunks of about 1300 - 1600 k, so 2 chunks
    chunked = torch.full(shape, 0.25, device="cuda")
    _gemm(A, Bm, chunked, bf16=bf16, **args)
    tol = 2e-3 if not bf16 else 2e-2
    assert (whole - ref).abs().max() < tol and (chunked - ref).abs().max() < tol
    # (split-K sums arrive in any order: the two runs agree to rounding, not bit for bit)
    assert (chunked - whole).abs().max() < 1e-3
    span_limit(1 << 12)  # a limit below one K step of rows is refused, not mis-addressed
    with pytest.raises(L.DvaeHipError, match="leading dimension too large"):
        _gemm(A, Bm, chunked, bf16=bf16, **args)


@pytest.mark.parametrize("bf16", [False, True], ids=["f32", "bf16"])
def test_gemm_row_gathered_kmajor_operand_is_bounded(bf16, span_limit):
    """A row-gathered K-major operand (the edge predictor's weight gradients gather state rows) is addressed
    with 32-bit offsets and cannot be chunked: the caller states how many rows the stored matrix has and a
    matrix beyond the span limit is refused instead of wrapping."""
    g = torch.Generator(device="cuda").manual_seed(78)
    rows, N, K = 600, 64, 96
    dY = torch.randn(rows, N, device="cuda", generator=g)
    X = torch.randn(rows + 40, K, device="cuda", generator=g)
    b_rows = torch.randperm(rows + 40, device="cuda", generator=g)[:rows].int()
    out = torch.zeros(N, K, device="cuda")
    _gemm(dY, X, out, N, K, rows, akm=1, bkm=1, b_rows=b_rows, acc=1, bf16=bf16)
    span_limit(X.numel() * 4 - 4)  # one float short of the stored matrix
    with pytest.raises(L.DvaeHipError, match="row-gathered K-major B spans"):
        _gemm(dY, X, out, N, K, rows, akm=1, bkm=1, b_rows=b_rows, acc=1, bf16=bf16)
    span_limit(0)
    d = L.GemmDesc()  # the bound is required
    d.A, d.B, d.C, d.b_rows = dY.data_ptr(), X.data_ptr(), out.data_ptr(), b_rows.data_ptr()
    d.lda, d.ldb, d.ldc, d.M, d.N, d.K = N, K, K, N, K, rows
    d.a_kmajor = d.b_kmajor = d.accumulate = 1
    assert L.load().dvae_gemm_f32(C.byref(d), _s()) != 0
    assert b"b_rows_bound" in L.load().dvae_last_error()


@pytest.mark.parametrize("case", ["n_not_mult4", "ldc_not_mult4", "bias_unaligned", "c_unaligned"])
def test_gemm_epilogue_paths_agree(case):
    """the 16-byte (transposed-accumulation) epilogue needs N % 4 == 0, ldc % 4 == 0 and 16-byte aligned C / bias;
    each violation must fall back to the dword epilogue and give the same numbers"""
    g = torch.Generator(device="cuda").manual_seed(11)
    M, K = 333, 96
    N = 30 if case == "n_not_mult4" else 72
    X = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g)
    bias_store = torch.randn(N + 4, device="cuda", generator=g)
    bias = bias_store[1:N + 1] if case == "bias_unaligned" else bias_store[:N]
    ldc = N + 2 if case == "ldc_not_mult4" else N + 4
    store = torch.full((M * ldc + 8,), 0.5, device="cuda")
    off = 1 if case == "c_unaligned" else 0
    out = store[off:off + M * ldc].view(M, ldc)
    for act, acc in ((0, 0), (1, 1), (2, 0)):
        out.fill_(0.5)
        _gemm(X, W, out, M, N, K, bias=bias, act=act, acc=acc)
        y = _ref(X, W.t()) + bias
        y = [y, torch.relu(y), torch.tanh(y)][act]
        ref = torch.full((M, ldc), 0.5, device="cuda")
        ref[:, :N] = y + (0.5 if acc else 0.0)
        assert (out - ref).abs().max() < 2e-4          # padding columns [N, ldc) untouched


@pytest.mark.parametrize("M,N,K,acc", [(9251, 512, 2048, 1), (7168, 32, 1024, 0), (129, 132, 4096, 1), (2048, 512, 9248, 1)])
def test_gemm_tile_and_split_choices(M, N, K, acc):
    """shapes on either side of the tile / split-K cost model's decisions, K-major B (dgrad form)"""
    g = torch.Generator(device="cuda").manual_seed(M + K)
    A = torch.randn(M, K, device="cuda", generator=g)
    Bm = torch.randn(K, N, device="cuda", generator=g)
    out = torch.ones(M, N, device="cuda")
    _gemm(A, Bm, out, M, N, K, bkm=1, acc=acc)
    ref = _ref(A, Bm) + (1.0 if acc else 0.0)
    assert (out - ref).abs().max() <= 1e-5 * K ** 0.5 * 4 + 1e-4


def test_gemm_rejects_bad_shapes():
    A = torch.zeros(8, 48, device="cuda")
    out = torch.zeros(8, 8, device="cuda")
    with pytest.raises(L.DvaeHipError, match="K"):
        _gemm(A, A, out, 8, 8, 48)


def test_gather_sum_and_scatter():
    g = torch.Generator().manual_seed(1)
    src = torch.randn(300, 64, generator=g).cuda()
    lens = torch.randint(0, 6, (200,), generator=g)
    off = torch.zeros(201, dtype=torch.int32)
    off[1:] = lens.cumsum(0)
    idx = torch.randint(0, 300, (int(off[-1]),), generator=g).int()
    out = torch.full((200, 64), float("nan")).cuda()
    lib = L.load()
    off_d, idx_d = off.cuda(), idx.cuda()  # keep the device copies alive across the call
    L.check(lib.dvae_gather_sum_f32(out.data_ptr(), 64, src.data_ptr(), 64, off_d.data_ptr(), idx_d.data_ptr(),
                                    200, 64, 0, _s()))
    ref = torch.zeros(200, 64)
    for r in range(200):
        for k in range(off[r], off[r + 1]):
            ref[r] += src[idx[k]].cpu()
    assert (out.cpu() - ref).abs().max() < 1e-5
    # scatter with unique indices
    perm = torch.randperm(300, generator=g)[: int(off[-1])].int() if off[-1] <= 300 else None
    if perm is not None:
        dst = torch.ones(300, 64).cuda()
        perm_d = perm.cuda()
        L.check(lib.dvae_scatter_rows_add_f32(dst.data_ptr(), 64, out.data_ptr(), 64, off_d.data_ptr(),
                                              perm_d.data_ptr(), 200, 64, _s()))
        ref2 = torch.ones(300, 64)
        for r in range(200):
            for k in range(off[r], off[r + 1]):
                ref2[perm[k]] += out[r].cpu()
        assert (dst.cpu() - ref2).abs().max() < 1e-5


def test_gru_pointwise_matches_torch_grucell():
    torch.manual_seed(3)
    H, Hp, T, rows = 20, 32, 5, 37
    cell = torch.nn.GRUCell(T, H).cuda()
    types = torch.randint(0, T, (rows,), device="cuda")
    hprev = torch.randn(rows, H, device="cuda", requires_grad=True)
    x = torch.nn.functional.one_hot(types, T).float()
    ref = cell(x, hprev)
    dh = torch.randn_like(ref)
    ref.backward(dh)
    # padded operands
    gh = torch.zeros(rows, 3 * Hp, device="cuda")
    gi = torch.zeros(T, 3 * Hp, device="cuda")
    bhn = torch.zeros(Hp, device="cuda")
    hp = torch.zeros(rows, Hp, device="cuda")
    hp[:, :H] = hprev.detach()
    ghraw = hprev.detach() @ cell.weight_hh.t()
    for q in range(3):
        gh[:, q * Hp:q * Hp + H] = ghraw[:, q * H:(q + 1) * H]
        gi[:, q * Hp:q * Hp + H] = (cell.weight_ih[q * H:(q + 1) * H, :].t() + cell.bias_ih[q * H:(q + 1) * H]
                                    + (cell.bias_hh[q * H:(q + 1) * H] if q < 2 else 0)).detach()
    bhn[:H] = cell.bias_hh[2 * H:].detach()
    hout = torch.zeros(rows, Hp, device="cuda")
    ghn = torch.zeros(rows, Hp, device="cuda")
    lib = L.load()
    types_i = types.int()
    L.check(lib.dvae_gru_pointwise_fwd(gh.data_ptr(), gi.data_ptr(), types_i.data_ptr(), bhn.data_ptr(), hp.data_ptr(),
                                       Hp, hout.data_ptr(), Hp, ghn.data_ptr(), rows, Hp, _s()))
    assert (hout[:, :H] - ref.detach()).abs().max() < 1e-6
    assert hout[:, H:].abs().max() == 0
    dhp = torch.zeros(rows, Hp, device="cuda")
    dhp[:, :H] = dh
    dgh = torch.zeros(rows, 3 * Hp, device="cuda")
    dgin = torch.zeros(rows, Hp, device="cuda")
    dprev = torch.zeros(rows, Hp, device="cuda")
    L.check(lib.dvae_gru_pointwise_bwd(dhp.data_ptr(), Hp, gh.data_ptr(), ghn.data_ptr(), hp.data_ptr(), Hp,
                                       dgh.data_ptr(), dgin.data_ptr(), dprev.data_ptr(), Hp, rows, Hp, _s()))
    # d hprev = direct + dgh @ Whh
    dgh_n = torch.cat([dgh[:, q * Hp:q * Hp + H] for q in range(3)], 1)
    dh_total = dprev[:, :H] + dgh_n @ cell.weight_hh.detach()
    assert (dh_total - hprev.grad).abs().max() < 2e-6
    dWhh = dgh_n.t() @ hprev.detach()
    assert (dWhh - cell.weight_hh.grad).abs().max() < 2e-5
    dbhh = dgh_n.sum(0)
    assert (dbhh - cell.bias_hh.grad).abs().max() < 2e-5
    dgi_n = torch.cat([dgh[:, :Hp][:, :H], dgh[:, Hp:2 * Hp][:, :H], dgin[:, :H]], 1)
    assert (dgi_n.sum(0) - cell.bias_ih.grad).abs().max() < 2e-5


def test_pair_kernels_match_torch():
    torch.manual_seed(4)
    HEp, NA, NB, NP = 96, 11, 13, 150
    ea = torch.randn(NA + 1, HEp, device="cuda", requires_grad=True)
    eb = torch.randn(NB + 1, HEp, device="cuda", requires_grad=True)
    b1 = torch.randn(HEp, device="cuda", requires_grad=True)
    w2 = torch.randn(HEp, device="cuda", requires_grad=True)
    b2 = torch.randn(1, device="cuda", requires_grad=True)
    pa = torch.randint(0, NA + 1, (NP,), device="cuda")
    ph = torch.randint(0, NB + 1, (NP,), device="cuda")
    tgt = torch.randint(0, 2, (NP,), device="cuda")
    s = (torch.relu(ea[pa] + eb[ph] + b1) * w2).sum(1) + b2
    ref = torch.nn.functional.binary_cross_entropy(torch.sigmoid(s), tgt.float(), reduction="sum")
    ref.backward()
    lib = L.load()
    dl = torch.zeros(NP, device="cuda")
    lsum = torch.zeros(1, device="cuda")
    pa_i, ph_i, tgt_u = pa.int(), ph.int(), tgt.to(torch.uint8)
    nblk = 8
    mask = torch.zeros(int(lib.dvae_pair_mask_words(NP, HEp)), dtype=torch.int32, device="cuda")
    dw2p = torch.full((nblk, HEp), float("nan"), device="cuda")
    L.check(lib.dvae_pair_bce_fwd(ea.data_ptr(), eb.data_ptr(), None, pa_i.data_ptr(), ph_i.data_ptr(), None,
                                  tgt_u.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                  dl.data_ptr(), lsum.data_ptr(), mask.data_ptr(), dw2p.data_ptr(), nblk, NP, HEp, _s()))
    assert abs(lsum.item() - ref.item()) < 1e-4 * ref.item()
    assert abs(dl.sum().item() - b2.grad.item()) < 1e-4
    assert (dw2p.sum(0) - w2.grad).abs().max() < 1e-3
    # grouped backward by the EA row, chunks of 4 pairs
    def chunks(key, nrows, ch):
        order = torch.argsort(key.cpu(), stable=True)
        cnt = torch.bincount(key.cpu(), minlength=nrows)
        off, row, start = [], [], 0
        for r in range(nrows):
            multi = cnt[r] > ch
            for s0 in range(0, int(cnt[r]), ch):
                off.append(start + s0)
                row.append(r | (0x80000000 if multi else 0))
            start += int(cnt[r])
        off.append(NP)
        row = np.array(row, np.uint32).view(np.int32)
        return (torch.tensor(off, dtype=torch.int32).cuda(), torch.from_numpy(row).cuda(), order.int().cuda())
    for mine, key, nrows, gref in ((ea, pa, NA + 1, ea.grad), (eb, ph, NB + 1, eb.grad)):
        off, row, pairs = chunks(key, nrows, 4)
        dm = torch.zeros_like(mine)
        L.check(lib.dvae_pair_bce_bwd_grouped(off.data_ptr(), row.data_ptr(), pairs.data_ptr(), dl.data_ptr(),
                                              mask.data_ptr(), w2.data_ptr(), None, dm.data_ptr(), row.numel(), HEp,
                                              _s()))
        assert (dm - gref).abs().max() < 1e-4
    assert (ea.grad.sum(0) - b1.grad).abs().max() < 1e-3  # db1 = column sum of dEA


@pytest.mark.parametrize("HEp", [128, 384, 2048, 3072, 4096])
def test_pair_kernels_widths(HEp):
    """every compiled width class of the pair kernels (1, 2, 8, 16 column iterations; 1 or 2 mask words; above 2048
    hidden units the four-waves-per-pair forward and the quarter-row backward)"""
    torch.manual_seed(HEp)
    NA, NB, NP = 9, 10, 60
    ea = torch.randn(NA, HEp, device="cuda", requires_grad=True)
    eb = torch.randn(NB, HEp, device="cuda", requires_grad=True)
    ec = torch.randn(5, HEp, device="cuda", requires_grad=True)
    b1 = torch.randn(HEp, device="cuda")
    w2 = (torch.randn(HEp, device="cuda") / HEp ** 0.5).requires_grad_(True)
    b2 = torch.randn(1, device="cuda")
    pa = torch.randint(0, NA, (NP,), device="cuda")
    ph = torch.randint(0, NB, (NP,), device="cuda")
    pc = torch.sort(torch.randint(0, 5, (NP,), device="cuda"))[0]
    tgt = torch.randint(0, 2, (NP,), device="cuda")
    s = (torch.relu(ea[pa] + eb[ph] + ec[pc] + b1) * w2).sum(1) + b2
    ref = torch.nn.functional.binary_cross_entropy(torch.sigmoid(s), tgt.float(), reduction="sum")
    ref.backward()
    lib = L.load()
    dl = torch.zeros(NP, device="cuda")
    lsum = torch.zeros(1, device="cuda")
    mask = torch.zeros(int(lib.dvae_pair_mask_words(NP, HEp)), dtype=torch.int32, device="cuda")
    dw2p = torch.zeros(4, HEp, device="cuda")
    pa_i, ph_i, pc_i, tgt_u = pa.int(), ph.int(), pc.int(), tgt.to(torch.uint8)
    L.check(lib.dvae_pair_bce_fwd(ea.data_ptr(), eb.data_ptr(), ec.data_ptr(), pa_i.data_ptr(), ph_i.data_ptr(),
                                  pc_i.data_ptr(), tgt_u.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                  dl.data_ptr(), lsum.data_ptr(), mask.data_ptr(), dw2p.data_ptr(), 4, NP, HEp, _s()))
    assert abs(lsum.item() - ref.item()) < 1e-4 * ref.item()
    assert (dw2p.sum(0) - w2.grad).abs().max() < 1e-4
    # by graph: pairs are contiguous per graph (grp_pairs NULL), one chunk per graph
    cnt = torch.bincount(pc.cpu(), minlength=5)
    off = torch.zeros(6, dtype=torch.int32)
    off[1:] = cnt.cumsum(0)
    rows = torch.arange(5, dtype=torch.int32).cuda()
    off_d = off.cuda()
    dc = torch.zeros(5, HEp, device="cuda")
    L.check(lib.dvae_pair_bce_bwd_grouped(off_d.data_ptr(), rows.data_ptr(), None, dl.data_ptr(), mask.data_ptr(),
                                          w2.data_ptr(), None, dc.data_ptr(), 5, HEp, _s()))
    assert (dc - ec.grad).abs().max() < 1e-4


def test_vertex_ce_matches_torch():
    torch.manual_seed(5)
    R, T, Tp = 70, 8, 32
    logits = torch.randn(R, Tp, device="cuda")
    lg = logits[:, :T].clone().requires_grad_(True)
    cnt = torch.randint(1, 4, (R,))
    off = torch.zeros(R + 1, dtype=torch.int32)
    off[1:] = cnt.cumsum(0)
    tg = torch.randint(0, T, (int(off[-1]),))
    lsm = torch.log_softmax(lg, 1)
    ref = 0
    for r in range(R):
        for k in range(off[r], off[r + 1]):
            ref = ref - lsm[r, tg[k]]
    ref.backward()
    dlog = torch.zeros(R, Tp, device="cuda")
    lsum = torch.zeros(1, device="cuda")
    off_d, tg_d = off.cuda(), tg.int().cuda()
    L.check(L.load().dvae_vertex_ce(logits.data_ptr(), Tp, off_d.data_ptr(), tg_d.data_ptr(),
                                    dlog.data_ptr(), lsum.data_ptr(), R, T, Tp, _s()))
    assert abs(lsum.item() - ref.item()) < 1e-4 * abs(ref.item())
    assert (dlog[:, :T] - lg.grad).abs().max() < 1e-5 and dlog[:, T:].abs().max() == 0


def test_gated_fused_gather_kernels_match_unfused():
    """gate/mapper pointwise part fused with the neighbouring CSR gather-sum (forward: next
    level's predecessor sum; backward: d msg from the consumers' d agg) == the two-kernel
    sequence, bit for bit (same fp32 operations in the same order)."""
    g = torch.Generator().manual_seed(11)
    B, Hp, levels = 50, 64, 4
    lib = L.load()
    msg = torch.randn(levels * B, Hp, generator=g).cuda()
    cur = 2  # current level; its rows of msg are overwritten by the kernel
    gm0 = torch.randn(B, 2 * Hp, generator=g).cuda()
    gv = torch.randn(levels, 2 * Hp, generator=g).cuda()
    # next level's predecessor lists: slots of the same graph in levels <= cur
    off = [0]
    idx = []
    for b in range(B):
        ks = [u for u in range(cur + 1) if torch.rand((), generator=g) < 0.6]
        idx += [u * B + b for u in ks]
        off.append(len(idx))
    off_d = torch.tensor(off, dtype=torch.int32).cuda()
    idx_d = torch.tensor(idx + [0], dtype=torch.int32).cuda()
    # reference: unfused pair
    gm_ref, msg_ref = gm0.clone(), msg.clone()
    L.check(lib.dvae_gated_pointwise_fwd(gm_ref.data_ptr(), gv.data_ptr(), msg_ref[cur * B:].data_ptr(), Hp, B, Hp,
                                         cur * B, B, _s()))
    agg_ref = torch.full((B, Hp), float("nan")).cuda()
    L.check(lib.dvae_gather_sum_f32(agg_ref.data_ptr(), Hp, msg_ref.data_ptr(), Hp, off_d.data_ptr(), idx_d.data_ptr(),
                                    B, Hp, 0, _s()))
    gm_f, msg_f = gm0.clone(), msg.clone()
    agg_f = torch.full((B, Hp), float("nan")).cuda()
    L.check(lib.dvae_gated_fwd_gather(gm_f.data_ptr(), gv[cur].data_ptr(), msg_f.data_ptr(), Hp, cur * B,
                                      agg_f.data_ptr(), Hp, off_d.data_ptr(), idx_d.data_ptr(), B, Hp, _s()))
    torch.cuda.synchronize()
    assert torch.equal(gm_f, gm_ref) and torch.equal(msg_f, msg_ref) and torch.equal(agg_f, agg_ref)
    # against plain torch
    a = gm0[:, :Hp] + gv[cur, :Hp]
    m = gm0[:, Hp:] + gv[cur, Hp:]
    assert (msg_f[cur * B:(cur + 1) * B] - torch.sigmoid(a) * m).abs().max() < 1e-6
    # backward twin: two lists
    dagg = torch.randn(levels * B, Hp, generator=g).cuda()
    dagg2 = torch.randn(30, Hp, generator=g).cuda()
    lens2 = torch.randint(0, 3, (B,), generator=g)
    off2 = torch.zeros(B + 1, dtype=torch.int32)
    off2[1:] = lens2.cumsum(0)
    idx2 = torch.randint(0, 30, (int(off2[-1]) + 1,), generator=g).int()
    off2_d, idx2_d = off2.cuda(), idx2.cuda()
    dmsg = torch.zeros(B, Hp).cuda()
    L.check(lib.dvae_gather_sum_f32(dmsg.data_ptr(), Hp, dagg.data_ptr(), Hp, off_d.data_ptr(), idx_d.data_ptr(),
                                    B, Hp, 0, _s()))
    L.check(lib.dvae_gather_sum_f32(dmsg.data_ptr(), Hp, dagg2.data_ptr(), Hp, off2_d.data_ptr(), idx2_d.data_ptr(),
                                    B, Hp, 1, _s()))
    dgm_ref = torch.zeros(B, 2 * Hp).cuda()
    L.check(lib.dvae_gated_pointwise_bwd(dmsg.data_ptr(), Hp, gm_ref.data_ptr(), dgm_ref.data_ptr(), B, Hp, _s()))
    dgm_f = torch.zeros(B, 2 * Hp).cuda()
    L.check(lib.dvae_gated_bwd_gather(dagg.data_ptr(), Hp, off_d.data_ptr(), idx_d.data_ptr(), dagg2.data_ptr(),
                                      off2_d.data_ptr(), idx2_d.data_ptr(), gm_ref.data_ptr(), dgm_f.data_ptr(), B, Hp,
                                      _s()))
    torch.cuda.synchronize()
    assert (dgm_f - dgm_ref).abs().max() < 1e-6


# ---------------------------------------------------------------- bf16-input MFMA GEMM
def _ref_bf16(a, b):
    """exact product of the bf16-rounded operands (what the kernel computes up to fp32 accumulation)"""
    return (a.bfloat16().double() @ b.bfloat16().double()).float()


@pytest.mark.parametrize("M,N,K", [(1024, 1536, 512), (100, 96, 64), (7, 32, 36), (4000, 2048, 512), (300, 40, 96),
                                   (129, 130 // 2 * 2, 200)])
def test_gemm_bf16_nt(M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g)
    out = torch.full((M, N), float("nan"), device="cuda")
    _gemm(A, W, out, M, N, K, bf16=True)
    ref = _ref_bf16(A, W.t())
    assert (out - ref).abs().max() <= 2e-6 * K ** 0.5 * 16


@pytest.mark.parametrize("M,N,K", [(300, 200, 72), (1100, 1300, 200), (2048, 3072, 1024), (5000, 4096, 136)])
@pytest.mark.parametrize("store", ["bf16xbf16", "f32xbf16", "bf16xf32", "f32xf32"])
def test_gemm_bf16_stored_operands(M, N, K, store, monkeypatch):
    """gemm_bf16_nn.hip: K-contiguous operands stored as bf16 or fp32, each side independently; 128 x 128 tiles for
    small products, 256 x 256 from three quarters of a round of the chip (the two large shapes), ragged M / N / K
    edges, row gather + scatter + bias + relu + accumulate through the shared epilogue.  The fp32-stored form must
    agree with the bf16-stored one exactly: both feed the matrix unit the same rounded values."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    X = torch.randn(M + 77, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g)
    bias = torch.randn(N, device="cuda", generator=g)
    a_rows = torch.randperm(M + 77, device="cuda", generator=g)[:M].int()
    c_rows = torch.randperm(M + 9, device="cuda", generator=g)[:M].int()
    sa, sb = store.split("x")
    Xs = X.bfloat16() if sa == "bf16" else X
    Ws = W.bfloat16() if sb == "bf16" else W
    if store == "f32xf32" and M * N < 256 * 256 * 192:
        pytest.skip("small fp32-stored products stay on gemm_bf16.hip (covered by the tests around this one)")
    out = torch.ones(M + 9, N, device="cuda")
    _gemm(Xs, Ws, out, M, N, K, a_rows=a_rows, c_rows=c_rows, bias=bias, act=1, acc=1, bf16=True)
    y = torch.relu(_ref_bf16(X[a_rows.long()], W.t()) + bias)
    ref = torch.ones(M + 9, N, device="cuda")
    ref[c_rows.long()] += y
    assert (out - ref).abs().max() <= 2e-6 * K ** 0.5 * 16 + 1e-5
    plain = torch.zeros(M, N, device="cuda")
    _gemm(Xs, Ws, plain, M, N, K, bf16=True)
    both = torch.zeros(M, N, device="cuda")
    _gemm(X.bfloat16(), W.bfloat16(), both, M, N, K, bf16=True)
    assert torch.equal(plain, both)


def test_gemm_bf16_stored_operand_rules():
    A = torch.randn(64, 64, device="cuda").bfloat16()
    W = torch.randn(64, 64, device="cuda").bfloat16()
    out = torch.zeros(64, 64, device="cuda")
    with pytest.raises(L.DvaeHipError, match="K-contiguous"):
        _gemm(A, W, out, 64, 64, 64, bkm=1, bf16=True)
    with pytest.raises(L.DvaeHipError, match="belong to dvae_gemm_bf16"):
        _gemm(A, W, out, 64, 64, 64)
    with pytest.raises(L.DvaeHipError, match="K % 8"):
        _gemm(A, W, out, 64, 64, 60, bf16=True)
    src = torch.randn(1000, device="cuda")
    dst = torch.empty(1000, device="cuda", dtype=torch.bfloat16)
    L.check(L.load().dvae_f32_to_bf16(src.data_ptr(), dst.data_ptr(), 1000, _s()))
    torch.cuda.synchronize()
    assert torch.equal(dst, src.bfloat16())


def test_gemm_bf16_layout_identity_check():
    K = 64
    A = torch.eye(K, device="cuda")
    W = (torch.arange(96 * K, device="cuda", dtype=torch.float32) % 251).reshape(96, K)  # exact in bf16
    out = torch.zeros(K, 96, device="cuda")
    _gemm(A, W, out, K, 96, K, bf16=True)
    assert torch.equal(out, W.t().contiguous())
    # K-major operands: C = A'^T B' with A' = I stored [k][m], asymmetric B' stored [k][n]
    A2 = torch.eye(128, device="cuda")
    B2 = (torch.arange(128 * 96, device="cuda", dtype=torch.float32) % 253).reshape(128, 96)
    out2 = torch.zeros(128, 96, device="cuda")
    _gemm(A2, B2, out2, 128, 96, 128, akm=1, bkm=1, acc=1, bf16=True)
    assert torch.equal(out2, B2)
    # dgrad form: A K-contiguous, B K-major
    out3 = torch.zeros(128, 96, device="cuda")
    _gemm(A2, B2, out3, 128, 96, 128, akm=0, bkm=1, bf16=True)
    assert torch.equal(out3, B2)


def test_gemm_bf16_bias_act_accumulate_gather_scatter():
    g = torch.Generator(device="cuda").manual_seed(5)
    rows_total, M, N, K = 500, 260, 160, 96
    X = torch.randn(rows_total, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g)
    bias = torch.randn(N, device="cuda", generator=g)
    a_rows = torch.randperm(rows_total, device="cuda", generator=g)[:M].int()
    c_rows = torch.randperm(400, device="cuda", generator=g)[:M].int()
    for act in (0, 1, 2):
        out = torch.ones(400, N, device="cuda")
        _gemm(X, W, out, M, N, K, a_rows=a_rows, c_rows=c_rows, bias=bias, act=act, acc=1, bf16=True)
        y = _ref_bf16(X[a_rows.long()], W.t()) + bias
        y = [y, torch.relu(y), torch.tanh(y)][act]
        ref = torch.ones(400, N, device="cuda")
        ref[c_rows.long()] += y
        assert (out - ref).abs().max() < 2e-4


@pytest.mark.parametrize("M,N,K", [(1000, 1536, 512), (77, 2048 // 4, 100), (1024, 512, 1024)])
def test_gemm_bf16_dgrad_kmajor_b(M, N, K):
    g = torch.Generator(device="cuda").manual_seed(6)
    dY = torch.randn(M, N, device="cuda", generator=g)   # dX[M,K] = dY[M,N] W[N,K]
    W = torch.randn(N, K, device="cuda", generator=g)
    out = torch.zeros(M, K, device="cuda")
    _gemm(dY, W, out, M, K, N, akm=0, bkm=1, acc=1, bf16=True)
    ref = _ref_bf16(dY, W)
    assert (out - ref).abs().max() < 1e-3


@pytest.mark.parametrize("rows,N,K", [(8192, 1536, 512), (333, 64, 32), (5000, 32, 1536), (70, 128, 512), (9251, 2048, 512),
                                      (16421, 1000, 1000), (20000, 2048, 1024)])
def test_gemm_bf16_wgrad_splitk(rows, N, K):
    """the last two shapes (>= 16 tiles of 256 x 256, >= 256 K steps) take the 8-wave 256 x 256 weight-gradient tile
    with its XCD-major (K range, tile) workgroup order; ragged M / N / K edges on the first of them"""
    g = torch.Generator(device="cuda").manual_seed(rows)
    dY = torch.randn(rows, N, device="cuda", generator=g)
    X = torch.randn(rows + 50, K, device="cuda", generator=g)
    b_rows = torch.randperm(rows + 50, device="cuda", generator=g)[:rows].int()
    out = torch.zeros(N, K, device="cuda")
    _gemm(dY, X, out, N, K, rows, akm=1, bkm=1, b_rows=b_rows, acc=1, bf16=True)
    ref = _ref_bf16(dY.t(), X[b_rows.long()])
    assert (out - ref).abs().max() <= 1e-5 * rows ** 0.5 * 4 + 1e-4
    out2 = torch.zeros(N, K, device="cuda")
    _gemm(dY, X, out2, N, K, rows, akm=1, bkm=1, acc=1, bf16=True)
    ref2 = _ref_bf16(dY.t(), X[:rows])
    assert (out2 - ref2).abs().max() <= 1e-5 * rows ** 0.5 * 4 + 1e-4


@pytest.mark.parametrize("rows,N,K", [(16421, 1000, 1000), (20000, 2048, 1024), (8192, 1536, 512), (5000, 32, 1536), (70, 128, 512)])
@pytest.mark.parametrize("b_half", [False, True])
def test_gemm_bf16_wgrad_stored_kmajor(rows, N, K, b_half, span_limit):
    """round 3: weight gradients with a bf16-STORED K-major A (the edge predictor's gradient tables as the pair backward
    writes them, d gh / d gm as the backward tails write them) and an fp32 or bf16-stored (gathered) B, on every tile of
    gemm_bf16.hip (256 x 256 for the first two shapes, 128 x 128, 32 x 128, 64 x 64): the same numbers as the product of
    the rounded operands held in fp32"""
    g = torch.Generator(device="cuda").manual_seed(rows + 1)
    dY = torch.randn(rows, N, device="cuda", generator=g).bfloat16()
    X = torch.randn(rows + 50, K, device="cuda", generator=g)
    Xs = X.bfloat16() if b_half else X
    b_rows = torch.randperm(rows + 50, device="cuda", generator=g)[:rows].int()
    out = torch.zeros(N, K, device="cuda")
    _gemm(dY, Xs, out, N, K, rows, akm=1, bkm=1, b_rows=b_rows, acc=1, bf16=True)
    same = torch.zeros(N, K, device="cuda")
    _gemm(dY.float(), X, same, N, K, rows, akm=1, bkm=1, b_rows=b_rows, acc=1, bf16=True)
    assert (out - same).abs().max() <= 1e-5 * rows ** 0.5 * 4 + 1e-4  # split-K atomics: the order of the partial sums varies
    ref = _ref_bf16(dY.float().t(), X[b_rows.long()])
    assert (out - ref).abs().max() <= 1e-5 * rows ** 0.5 * 4 + 1e-4
    out2 = torch.zeros(N, K, device="cuda")
    _gemm(dY, Xs, out2, N, K, rows, akm=1, bkm=1, acc=1, bf16=True)
    ref2 = _ref_bf16(dY.float().t(), X[:rows])
    assert (out2 - ref2).abs().max() <= 1e-5 * rows ** 0.5 * 4 + 1e-4
    # beyond the 32-bit span of a K-major operand: accumulating K chunks, advanced in 2-byte elements (no gather: a gathered
    # operand cannot be rebased)
    if rows >= 1000:  # (a chunk is at least one 64-row K step)
        span_limit(rows * max(N * 2, K * (2 if b_half else 4)) // 3 + 4096)  # about three chunks of the wider operand
        out3 = torch.zeros(N, K, device="cuda")
        _gemm(dY, Xs, out3, N, K, rows, akm=1, bkm=1, acc=1, bf16=True)
        span_limit(0)
        assert (out3 - ref2).abs().max() <= 1e-5 * rows ** 0.5 * 4 + 1e-4
    # a small corner of it (the tail of a chunked contraction): same tile, one K range
    small = torch.zeros(64, 64, device="cuda")
    nc, kk = min(64, N, K), min(200, rows)
    _gemm(dY, Xs, small, nc, nc, kk, akm=1, bkm=1, acc=1, bf16=True)
    assert (small[:nc, :nc] - _ref_bf16(dY.float()[:kk, :nc].t(), X[:kk, :nc])).abs().max() <= 2e-4
    with pytest.raises(L.DvaeHipError, match="accumulating weight-gradient"):
        _gemm(dY, Xs, small, nc, nc, kk, akm=1, bkm=1, acc=0, bf16=True)


@pytest.mark.parametrize("rows,Hp", [(1024, 512), (100, 64), (7, 32), (333, 1024)])
def test_gru_level_fused_matches_gemm_plus_pointwise(rows, Hp):
    """chain.hip: hidden product + GRUCell tail in one launch == dvae_gemm_f32 followed by
    dvae_gru_pointwise_fwd up to the summation order over K (tolerance 2e-5 absolute on O(1)
    activations), and == torch.nn.GRUCell arithmetic in fp64 (1e-5)."""
    g = torch.Generator().manual_seed(rows + Hp)
    T = 8
    hprev = torch.randn(rows, Hp, generator=g).cuda()
    whh = (torch.randn(3 * Hp, Hp, generator=g) / Hp ** 0.5).cuda()
    gi = torch.randn(T, 3 * Hp, generator=g).cuda()
    bhn = torch.randn(Hp, generator=g).cuda()
    types = torch.randint(0, T, (rows,), generator=g).int().cuda()
    lib = L.load()
    gates_ref = torch.full((rows, 3 * Hp), float("nan")).cuda()
    _gemm(hprev, whh, gates_ref, rows, 3 * Hp, Hp)
    hout_ref = torch.zeros(rows, Hp).cuda()
    ghn_ref = torch.zeros(rows, Hp).cuda()
    L.check(lib.dvae_gru_pointwise_fwd(gates_ref.data_ptr(), gi.data_ptr(), types.data_ptr(), bhn.data_ptr(),
                                       hprev.data_ptr(), Hp, hout_ref.data_ptr(), Hp, ghn_ref.data_ptr(), rows, Hp, _s()))
    gates = torch.full((rows, 3 * Hp), float("nan")).cuda()
    hout = torch.full((rows, Hp), float("nan")).cuda()
    ghn = torch.full((rows, Hp), float("nan")).cuda()
    L.check(lib.dvae_gru_level_fwd(hprev.data_ptr(), Hp, whh.data_ptr(), gi.data_ptr(), types.data_ptr(), bhn.data_ptr(),
                                   gates.data_ptr(), ghn.data_ptr(), hout.data_ptr(), Hp, rows, Hp, _s()))
    torch.cuda.synchronize()
    for a, b in ((gates, gates_ref), (ghn, ghn_ref), (hout, hout_ref)):
        assert torch.isfinite(a).all() and (a - b).abs().max() < 2e-5
    # fp64 GRUCell arithmetic
    gh = hprev.double() @ whh.double().t()
    gsel = gi.double()[types.long()]
    r = torch.sigmoid(gsel[:, :Hp] + gh[:, :Hp])
    z = torch.sigmoid(gsel[:, Hp:2 * Hp] + gh[:, Hp:2 * Hp])
    nn_ = torch.tanh(gsel[:, 2 * Hp:] + r * (gh[:, 2 * Hp:] + bhn.double()))
    assert (hout.double() - ((1 - z) * nn_ + z * hprev.double())).abs().max() < 1e-5


@pytest.mark.parametrize("B,Hp,with_next", [(1024, 512, True), (50, 64, True), (70, 96, False)])
def test_gated_level_fused_matches_gemm_plus_gather(B, Hp, with_next):
    """chain.hip: gate/mapper product + pointwise tail + next level's predecessor sum in one launch ==
    dvae_gemm_f32 followed by dvae_gated_fwd_gather (2e-5 absolute: summation order over K)."""
    g = torch.Generator().manual_seed(B + Hp)
    levels, cur = 4, 2
    lib = L.load()
    msg = torch.randn(levels * B, Hp, generator=g).cuda()
    h = torch.randn(B, Hp, generator=g).cuda()
    wgm = (torch.randn(2 * Hp, Hp, generator=g) / Hp ** 0.5).cuda()
    gv = torch.randn(levels, 2 * Hp, generator=g).cuda()
    off, idx = [0], []
    for b in range(B):
        ks = [u for u in range(cur + 1) if torch.rand((), generator=g) < 0.6]
        idx += [u * B + b for u in ks]
        off.append(len(idx))
    off_d = torch.tensor(off, dtype=torch.int32).cuda()
    idx_d = torch.tensor(idx + [0], dtype=torch.int32).cuda()
    gm_ref = torch.full((B, 2 * Hp), float("nan")).cuda()
    _gemm(h, wgm, gm_ref, B, 2 * Hp, Hp)
    msg_ref = msg.clone()
    agg_ref = torch.full((B, Hp), float("nan")).cuda()
    L.check(lib.dvae_gated_fwd_gather(gm_ref.data_ptr(), gv[cur].data_ptr(), msg_ref.data_ptr(), Hp, cur * B,
                                      agg_ref.data_ptr() if with_next else None, Hp, off_d.data_ptr(), idx_d.data_ptr(),
                                      B, Hp, _s()))
    gm_f = torch.full((B, 2 * Hp), float("nan")).cuda()
    msg_f = msg.clone()
    agg_f = torch.full((B, Hp), float("nan")).cuda()
    L.check(lib.dvae_gated_level_fwd(h.data_ptr(), Hp, wgm.data_ptr(), gv[cur].data_ptr(), gm_f.data_ptr(),
                                     msg_f.data_ptr(), Hp, cur * B, agg_f.data_ptr() if with_next else None, Hp,
                                     off_d.data_ptr(), idx_d.data_ptr(), B, Hp, _s()))
    torch.cuda.synchronize()
    assert torch.isfinite(gm_f).all() and (gm_f - gm_ref).abs().max() < 2e-5
    assert (msg_f - msg_ref).abs().max() < 2e-5
    # rows of other levels are untouched
    assert torch.equal(msg_f[:cur * B], msg[:cur * B]) and torch.equal(msg_f[(cur + 1) * B:], msg[(cur + 1) * B:])
    if with_next:
        assert (agg_f - agg_ref).abs().max() < 5e-5


@pytest.mark.parametrize("rows,Hp,K", [(1024, 512, 1536), (1024, 512, 1024), (100, 64, 128), (7, 32, 96), (130, 96, 288)])
def test_level_dgrad_acc_matches_fp64(rows, Hp, K):
    """chain.hip: dx += dy w (w K-major) against fp64; K % 64 == 32 exercises the half step."""
    g = torch.Generator().manual_seed(rows + Hp + K)
    dy = torch.randn(rows, K, generator=g).cuda()
    w = (torch.randn(K, Hp, generator=g) / K ** 0.5).cuda()
    dx0 = torch.randn(rows, Hp, generator=g).cuda()
    dx = dx0.clone()
    L.check(L.load().dvae_level_dgrad_acc(dy.data_ptr(), K, K, w.data_ptr(), dx.data_ptr(), Hp, rows, Hp, _s()))
    torch.cuda.synchronize()
    ref = dx0.double() + dy.double() @ w.double()
    assert torch.isfinite(dx).all() and (dx.double() - ref).abs().max() < 2e-5


@pytest.mark.parametrize("rows,Hp", [(1024, 512), (100, 64), (37, 96)])
def test_gru_level_bwd_fused_matches_dgrad_plus_pointwise(rows, Hp):
    """chain.hip: dh += dgm [Wg|Wm] and the GRU pointwise backward in one launch == dvae_gemm_f32
    (dgrad) followed by dvae_gru_pointwise_bwd (2e-5 absolute: summation order over K)."""
    g = torch.Generator().manual_seed(rows * 3 + Hp)
    K = 2 * Hp
    lib = L.load()
    dgm = torch.randn(rows, K, generator=g).cuda()
    w = (torch.randn(K, Hp, generator=g) / K ** 0.5).cuda()
    dh0 = torch.randn(rows, Hp, generator=g).cuda()
    gates = torch.rand(rows, 3 * Hp, generator=g).cuda()
    gates[:, 2 * Hp:] = gates[:, 2 * Hp:] * 2 - 1
    ghn = torch.randn(rows, Hp, generator=g).cuda()
    hprev = torch.randn(rows, Hp, generator=g).cuda()
    # reference: accumulate dgrad, then pointwise backward
    dh_ref = dh0.clone()
    _gemm(dgm, w, dh_ref, rows, Hp, K, bkm=1, acc=1)
    outs_ref = [torch.full((rows, 3 * Hp), float("nan")).cuda(), torch.full((rows, Hp), float("nan")).cuda(),
                torch.full((rows, Hp), float("nan")).cuda()]
    L.check(lib.dvae_gru_pointwise_bwd(dh_ref.data_ptr(), Hp, gates.data_ptr(), ghn.data_ptr(), hprev.data_ptr(), Hp,
                                       outs_ref[0].data_ptr(), outs_ref[1].data_ptr(), outs_ref[2].data_ptr(), Hp, rows,
                                       Hp, _s()))
    dh = dh0.clone()
    outs = [torch.full((rows, 3 * Hp), float("nan")).cuda(), torch.full((rows, Hp), float("nan")).cuda(),
            torch.full((rows, Hp), float("nan")).cuda()]
    L.check(lib.dvae_gru_level_bwd(dgm.data_ptr(), K, K, w.data_ptr(), dh.data_ptr(), Hp, gates.data_ptr(),
                                   ghn.data_ptr(), hprev.data_ptr(), Hp, outs[0].data_ptr(), outs[1].data_ptr(),
                                   outs[2].data_ptr(), Hp, rows, Hp, _s()))
    torch.cuda.synchronize()
    assert (dh - dh_ref).abs().max() < 2e-5
    for a, b in zip(outs, outs_ref):
        assert torch.isfinite(a).all() and (a - b).abs().max() < 5e-5


def test_chain_level_batch_two_levels_equals_two_launches():
    """dvae_chain_level_batch with count = 2 (both encoder directions of a level in one grid) gives,
    bit for bit, what two single-level launches give (same kernel, same summation order: more rows than the
    16-row and 32-row kernels of small levels take, which walk K in another order)."""
    g = torch.Generator().manual_seed(77)
    rows, Hp, T = 300, 64, 8
    lib = L.load()

    def mk():
        return dict(hprev=torch.randn(rows, Hp, generator=g).cuda(),
                    whh=(torch.randn(3 * Hp, Hp, generator=g) / Hp ** 0.5).cuda(),
                    gi=torch.randn(T, 3 * Hp, generator=g).cuda(), bhn=torch.randn(Hp, generator=g).cuda(),
                    types=torch.randint(0, T, (rows,), generator=g).int().cuda())

    def outs():
        return [torch.full((rows, 3 * Hp), float("nan")).cuda(), torch.full((rows, Hp), float("nan")).cuda(),
                torch.full((rows, Hp), float("nan")).cuda()]

    ins = [mk(), mk()]
    single = [outs(), outs()]
    for i, o in zip(ins, single):
        L.check(lib.dvae_gru_level_fwd(i["hprev"].data_ptr(), Hp, i["whh"].data_ptr(), i["gi"].data_ptr(),
                                       i["types"].data_ptr(), i["bhn"].data_ptr(), o[0].data_ptr(), o[1].data_ptr(),
                                       o[2].data_ptr(), Hp, rows, Hp, _s()))
    both = [outs(), outs()]
    descs = (L.LevelDesc * 2)()
    for k, (i, o) in enumerate(zip(ins, both)):
        descs[k].a, descs[k].lda, descs[k].w = i["hprev"].data_ptr(), Hp, i["whh"].data_ptr()
        descs[k].gi_tab, descs[k].types, descs[k].bhn = i["gi"].data_ptr(), i["types"].data_ptr(), i["bhn"].data_ptr()
        descs[k].gates, descs[k].ghn, descs[k].hout, descs[k].ld_hout = o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), Hp
    L.check(lib.dvae_chain_level_batch(0, descs, 2, rows, Hp, _s()))
    torch.cuda.synchronize()
    for a, b in zip(single, both):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    # more than two levels per launch is refused
    assert lib.dvae_chain_level_batch(0, descs, 3, rows, Hp, _s()) != 0


@pytest.mark.parametrize("two_lists", [False, True])
def test_level_dgrad_acc_with_fused_gather_tail(two_lists):
    """DGRAD_ACC with dgm_next set == dvae_level_dgrad_acc followed by dvae_gated_bwd_gather for the
    level processed next, bit for bit (the tail sums the consumers in list order and takes this
    launch's own row from registers: the value it has just stored)."""
    g = torch.Generator().manual_seed(91 + two_lists)
    B, Hp, levels, cur = 100, 64, 4, 2          # current level 2; the next one processed is level 1
    K = 3 * Hp
    lib = L.load()
    dagg0 = torch.randn(levels * B, Hp, generator=g).cuda()
    dgh = torch.randn(B, K, generator=g).cuda()
    whh = (torch.randn(K, Hp, generator=g) / K ** 0.5).cuda()
    gm = torch.rand(B, 2 * Hp, generator=g).cuda()
    # consumers of level-1 rows: slots of the same graph in levels 2 (current) and 3 (finished)
    off, idx = [0], []
    for b in range(B):
        idx += [u * B + b for u in (2, 3) if torch.rand((), generator=g) < 0.7]
        off.append(len(idx))
    off_d = torch.tensor(off, dtype=torch.int32).cuda()
    idx_d = torch.tensor(idx + [0], dtype=torch.int32).cuda()
    dagg2 = torch.randn(30, Hp, generator=g).cuda()
    lens2 = torch.randint(0, 3, (B,), generator=g)
    off2 = torch.zeros(B + 1, dtype=torch.int32)
    off2[1:] = lens2.cumsum(0)
    off2_d = off2.cuda()
    idx2_d = torch.randint(0, 30, (int(off2[-1]) + 1,), generator=g).int().cuda()
    # reference: two launches
    dagg_ref = dagg0.clone()
    L.check(lib.dvae_level_dgrad_acc(dgh.data_ptr(), K, K, whh.data_ptr(), dagg_ref[cur * B:].data_ptr(), Hp, B, Hp, _s()))
    dgm_ref = torch.full((B, 2 * Hp), float("nan")).cuda()
    L.check(lib.dvae_gated_bwd_gather(dagg_ref.data_ptr(), Hp, off_d.data_ptr(), idx_d.data_ptr(),
                                      dagg2.data_ptr() if two_lists else None, off2_d.data_ptr() if two_lists else None,
                                      idx2_d.data_ptr() if two_lists else None, gm.data_ptr(), dgm_ref.data_ptr(), B, Hp, _s()))
    # fused
    dagg = dagg0.clone()
    dgm = torch.full((B, 2 * Hp), float("nan")).cuda()
    d = L.LevelDesc()
    d.a, d.lda, d.K, d.w = dgh.data_ptr(), K, K, whh.data_ptr()
    d.dx, d.ld_dx = dagg[cur * B:].data_ptr(), Hp
    d.gather_src, d.next_off, d.idx, d.cur_row0 = dagg.data_ptr(), off_d.data_ptr(), idx_d.data_ptr(), cur * B
    if two_lists:
        d.gather_src2, d.off2, d.idx2 = dagg2.data_ptr(), off2_d.data_ptr(), idx2_d.data_ptr()
    d.gm, d.dgm_next = gm.data_ptr(), dgm.data_ptr()
    L.check(lib.dvae_chain_level_batch(2, C.byref(d), 1, B, Hp, _s()))
    torch.cuda.synchronize()
    assert torch.equal(dagg, dagg_ref)
    assert torch.isfinite(dgm).all() and torch.equal(dgm, dgm_ref)


@pytest.mark.parametrize("M,N,K,bf16", [(700, 1024, 32, False), (50, 40, 64, False), (300, 256, 32, True)])
def test_gemm_relu_mask_epilogue(M, N, K, bf16):
    """dgrad with relu_mask: dX = (dY W) where mask > 0 else 0 (ReLU backward as the product's epilogue)."""
    g = torch.Generator().manual_seed(M + N)
    dY = torch.randn(M, K, generator=g).cuda()
    W = torch.randn(K, N, generator=g).cuda()             # K-major B operand: [K][N]
    mask = torch.randn(M, N, generator=g).clamp(min=0).cuda()  # a ReLU output: zeros and positives
    out = torch.full((M, N), float("nan")).cuda()
    d = L.GemmDesc()
    d.A, d.B, d.C = dY.data_ptr(), W.data_ptr(), out.data_ptr()
    d.lda, d.ldb, d.ldc = K, N, N
    d.M, d.N, d.K = M, N, K
    d.b_kmajor = 1
    d.relu_mask = mask.data_ptr()
    fn = L.load().dvae_gemm_bf16 if bf16 else L.load().dvae_gemm_f32
    L.check(fn(C.byref(d), _s()), "gemm")
    torch.cuda.synchronize()
    ref = torch.where(mask > 0, (dY.double() @ W.double()).float(), torch.zeros((), device="cuda"))
    tol = 0.3 if bf16 else 1e-4
    assert torch.isfinite(out).all() and (out - ref).abs().max() < tol
    assert (out[mask <= 0] == 0).all()
    # not combinable with accumulate
    d.accumulate = 1
    assert fn(C.byref(d), _s()) != 0


@pytest.mark.parametrize("rows,nb,Hp,T", [(8192, 0, 512, 8), (300, 77, 96, 7), (5, 0, 32, 3)])
def test_type_csr_and_type_colsum_match_torch(rows, nb, Hp, T):
    """rows sorted by type + one sweep == the per-type column sums (d W_ih, d b_ih of a GRUCell with one-hot
    input) and the all-rows column sum (d b_hn) computed with torch in fp64."""
    g = torch.Generator().manual_seed(rows + Hp)
    n = rows + nb
    types = torch.randint(0, T, (n,), generator=g).int().cuda()
    dgh = torch.randn(n, 3 * Hp, generator=g).cuda()
    dgin = torch.randn(n, Hp, generator=g).cuda()
    off = torch.full((T + 1,), -1, dtype=torch.int32).cuda()
    idx = torch.full((n,), -1, dtype=torch.int32).cuda()
    lib = L.load()
    tb = types[rows:] if nb else None
    L.check(lib.dvae_type_csr(types.data_ptr(), rows, tb.data_ptr() if nb else None, nb, T, off.data_ptr(), idx.data_ptr(), _s()))
    torch.cuda.synchronize()
    cnt = torch.bincount(types.long(), minlength=T).cpu()
    assert off.cpu().tolist() == [0] + cnt.cumsum(0).tolist()
    ic = idx.cpu().long()
    assert sorted(ic.tolist()) == list(range(n))
    for t in range(T):
        assert (types.cpu()[ic[off[t].item():off[t + 1].item()]] == t).all()
    gi = torch.randn(32, 3 * Hp, generator=g).cuda()
    bhn = torch.randn(Hp, generator=g).cuda()
    gi0, bhn0 = gi.clone(), bhn.clone()
    L.check(lib.dvae_type_colsum(dgh.data_ptr(), dgin.data_ptr(), off.data_ptr(), idx.data_ptr(), T, gi.data_ptr(),
                                 bhn.data_ptr(), n, Hp, _s()))
    torch.cuda.synchronize()
    oh = torch.nn.functional.one_hot(types.long(), 32).double()
    ref_gi = gi0.double().clone()
    ref_gi[:, :2 * Hp] += oh.t() @ dgh[:, :2 * Hp].double()
    ref_gi[:, 2 * Hp:] += oh.t() @ dgin.double()
    ref_bhn = bhn0.double() + dgh[:, 2 * Hp:].double().sum(0)
    scale = max(1.0, n ** 0.5)
    assert (gi.double() - ref_gi).abs().max() < 2e-5 * scale
    assert (bhn.double() - ref_bhn).abs().max() < 2e-5 * scale


def test_side_streams_are_per_caller_stream_and_concurrent_calls_are_independent():
    """The entry points fork part of their work onto an internal side stream.  It is keyed by (device, caller
    stream): two caller streams never share one (nor its fork / join events), so two host threads driving two
    streams at once get the results of the same calls made one after the other."""
    import threading
    import numpy as np
    from common import make_model
    lib = L.load()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    side = [C.c_void_p(), C.c_void_p(), C.c_void_p()]
    for st, out in ((s1, side[0]), (s2, side[1]), (s1, side[2])):
        L.check(lib.dvae_debug_side_stream(C.c_void_p(st.cuda_stream), C.byref(out)), "side stream")
    assert side[0].value != side[1].value and side[0].value == side[2].value
    assert side[0].value not in (s1.cuda_stream, s2.cuda_stream)

    ctor = dict(max_n=8, nvt=8, START_TYPE=0, END_TYPE=1, hs=72, nz=10, bidirectional=1)
    model = make_model("DVAE", ctor, 3).cuda().train()
    rows = [dvae_amd.data.synth_enas_rows(96, 6, 6, 0.3, sd) for sd in (1, 2)]
    gas = [dvae_amd.data.enas_rows_to_arrays(r, 8) for r in rows]
    batches = [model._collate_fn(g) for g in gas]
    eps = [torch.randn(96, 10, device="cuda") for _ in gas]
    model._packed_weights()
    torch.cuda.synchronize()

    def run(i, stream, out, reps):
        with torch.cuda.stream(stream):
            for _ in range(reps):
                mu, lv = model.encode(batches[i])
                loss, _, _ = model.loss(mu, lv, batches[i], eps=eps[i])
                g = torch.autograd.grad(loss, [model.grud.weight_hh, model.grue_backward.weight_hh])
            stream.synchronize()
        out[i] = (loss.item(), mu.detach().clone(), [x.clone() for x in g])

    serial, conc = {}, {}
    run(0, s1, serial, 1)
    run(1, s2, serial, 1)
    th = [threading.Thread(target=run, args=(i, st, conc, 6)) for i, st in ((0, s1), (1, s2))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in (0, 1):
        assert abs(conc[i][0] - serial[i][0]) <= 1e-6 * abs(serial[i][0])
        assert torch.equal(conc[i][1], serial[i][1])  # forward results are bit-reproducible
        for a, b in zip(conc[i][2], serial[i][2]):
            assert (a - b).abs().max() <= 1e-5 * b.abs().max()  # split-K sums arrive in any order


def test_launch_counter_counts_library_launches():
    """dvae_debug_launch_count (bench.py's config.launches_per_step): one product = one launch, a level
    operation = one launch; it never goes backwards."""
    lib = L.load()
    A = torch.randn(64, 64, device="cuda")
    W = torch.randn(64, 64, device="cuda")
    out = torch.zeros(64, 64, device="cuda")
    n0 = lib.dvae_debug_launch_count()
    _gemm(A, W, out, 64, 64, 64)
    n1 = lib.dvae_debug_launch_count()
    assert n1 == n0 + 1
    dx = torch.zeros(64, 64, device="cuda")
    L.check(lib.dvae_level_dgrad_acc(A.data_ptr(), 64, 64, W.data_ptr(), dx.data_ptr(), 64, 64, 64, _s()))
    torch.cuda.synchronize()
    assert lib.dvae_debug_launch_count() == n1 + 1
