"""dvae_decode_static (include/dvae_hip.h): generation with the workspace in a fixed device array of the
library, for callers that cannot allocate.  Checked through the C-ABI against dvae_decode (caller-provided
workspace) on the same inputs: integer outputs, so the bar is exact equality.  Two calls in a row on the same
GPU, the second with a smaller batch and another model shape: the shared array keeps the first call's
contents, which must not leak into the second call's results."""
import ctypes as C

import numpy as np
import pytest
import torch

from _dvae_pkg import dvae_amd
from common import make_model

pytestmark = pytest.mark.gpu
L = dvae_amd._lib


def _decode_both(model, B, seed, stochastic=True):
    lib = L.load()
    dev = torch.device("cuda")
    n = model.max_n
    g = torch.Generator(device="cuda").manual_seed(seed)
    z = torch.randn(B, model.nz, device=dev, generator=g)
    u_type = torch.rand(B, n, device=dev, generator=g)
    u_edge = torch.rand(B, n, n, device=dev, generator=g)
    pw = model._packed_weights()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []
    for static in (False, True):
        types = torch.full((B, n), -7, dtype=torch.int32, device=dev)
        pmask = torch.full((B, n), -7, dtype=torch.int64, device=dev)
        n_g = torch.full((B,), -7, dtype=torch.int32, device=dev)
        common = (z.data_ptr(), int(stochastic), u_type.data_ptr(), u_edge.data_ptr(), types.data_ptr(), pmask.data_ptr(),
                  n_g.data_ptr(), stream)
        if static:
            L.check(lib.dvae_decode_static(C.byref(model._desc), B, pw.data_ptr(), *common), "dvae_decode_static")
        else:
            ws = torch.full((lib.dvae_decode_ws_floats(C.byref(model._desc), B),), float("nan"), device=dev)
            L.check(lib.dvae_decode(C.byref(model._desc), B, pw.data_ptr(), ws.data_ptr(), *common), "dvae_decode")
        torch.cuda.synchronize()
        outs.append((types.cpu().numpy(), pmask.cpu().numpy(), n_g.cpu().numpy()))
    return outs


def _sharpen(model):
    with torch.no_grad():  # decisive predictors: graphs of several sizes, not coin flips at every draw
        for k, p in model.named_parameters():
            if k.startswith(("add_vertex", "add_edge", "fc3")):
                p.mul_(4.0)


def test_decode_static_equals_decode_on_two_calls_in_a_row():
    big = make_model("DVAE", dict(max_n=9, nvt=7, START_TYPE=0, END_TYPE=1, hs=72, nz=10, bidirectional=0), 4)
    small = make_model("DVAE_BN", dict(max_n=8, nvt=8, START_TYPE=0, END_TYPE=1, hs=40, nz=6, bidirectional=0), 5)
    _sharpen(big)
    _sharpen(small)
    big.cuda().eval()
    small.cuda().eval()
    for model, B, seed in ((big, 200, 1), (small, 37, 2)):  # the second call is smaller and lays the array out differently
        (t0, m0, k0), (t1, m1, k1) = _decode_both(model, B, seed)
        assert np.array_equal(k0, k1) and np.array_equal(t0, t1) and np.array_equal(m0, m1)
        assert k0.min() >= 2 and k0.max() <= model.max_n and len(set(k0.tolist())) > 1
    # deterministic decoding takes no uniforms
    (t0, m0, k0), (t1, m1, k1) = _decode_both(big, 16, 3, stochastic=False)
    assert np.array_equal(k0, k1) and np.array_equal(t0, t1) and np.array_equal(m0, m1)


def test_decode_static_rejects_what_does_not_fit():
    lib = L.load()
    model = make_model("DVAE", dict(max_n=8, nvt=8, START_TYPE=0, END_TYPE=1, hs=501, nz=56, bidirectional=0), 1).cuda().eval()
    cap = lib.dvae_decode_static_max_floats()
    B = 1
    while lib.dvae_decode_ws_floats(C.byref(model._desc), B + 1) <= cap:
        B += 1
    assert B == 263
    (t0, m0, k0), (t1, m1, k1) = _decode_both(model, B, 9)  # the largest batch that fits
    assert np.array_equal(k0, k1) and np.array_equal(t0, t1) and np.array_equal(m0, m1)
    z = torch.zeros(B + 1, 56, device="cuda")
    out = torch.zeros(B + 1, 8, dtype=torch.int64, device="cuda")
    rc = lib.dvae_decode_static(C.byref(model._desc), B + 1, model._packed_weights().data_ptr(), z.data_ptr(), 0, None, None,
                                out.data_ptr(), out.data_ptr(), out.data_ptr(),
                                C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc != 0 and b"static scratch holds" in lib.dvae_last_error()


def test_decode_static_allocates_nothing_and_refuses_overlapping_calls():
    """(a) The property the entry point exists for: after the one-time set-up (symbol look-up on the first call on a
    device) a call takes no device memory -- free memory as the driver reports it is the same before and after a call
    whose inputs and outputs were allocated beforehand.  (b) Calls that overlap on one GPU share the array: a call on
    ANOTHER stream while the previous one is still running is refused with an error instead of racing; calls on one
    stream are ordered by the stream and pass."""
    lib = L.load()
    dev = torch.device("cuda")
    model = make_model("DVAE", dict(max_n=8, nvt=8, START_TYPE=0, END_TYPE=1, hs=72, nz=10, bidirectional=0), 7).cuda().eval()
    B, n = 64, model.max_n
    z = torch.randn(B, model.nz, device=dev)
    types = torch.zeros(B, n, dtype=torch.int32, device=dev)
    pmask = torch.zeros(B, n, dtype=torch.int64, device=dev)
    n_g = torch.zeros(B, dtype=torch.int32, device=dev)
    pw = model._packed_weights()

    def call(stream):
        return lib.dvae_decode_static(C.byref(model._desc), B, pw.data_ptr(), z.data_ptr(), 0, None, None, types.data_ptr(),
                                      pmask.data_ptr(), n_g.data_ptr(), C.c_void_p(stream.cuda_stream))

    main = torch.cuda.current_stream()
    L.check(call(main), "dvae_decode_static")  # set-up happens here at the latest
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    L.check(call(main), "dvae_decode_static")
    L.check(call(main), "dvae_decode_static")  # back to back on ONE stream: ordered, accepted
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0, "dvae_decode_static changed the device's free memory by %d bytes" % (free0 - free1)
    ref = n_g.clone()

    # overlap: keep the main stream busy for a few milliseconds in front of a call, then call on a second stream
    other = torch.cuda.Stream()
    big = torch.randn(4096, 4096, device=dev)
    out = torch.empty_like(big)
    torch.cuda.synchronize()
    for _ in range(20):
        torch.mm(big, big, out=out)
    L.check(call(main), "dvae_decode_static")
    rc = call(other)
    assert rc != 0 and b"still running" in lib.dvae_last_error()
    torch.cuda.synchronize()
    L.check(call(other), "dvae_decode_static")  # the earlier call has finished: accepted
    torch.cuda.synchronize()
    assert torch.equal(n_g, ref)
