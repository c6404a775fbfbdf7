"""The RCCL exchange step on a one-GPU box: a 1-rank "nccl" (= RCCL) process group with
DVAE_FORCE_COLLECTIVE=1 runs the very calls an N-rank job makes -- the asynchronous all-reduce
of the decoder-only part of the flat gradient buffer launched from inside backward(), the
wait + all-reduce of the rest -- and a SUM over one rank must leave every gradient unchanged.
(The N > 1 arithmetic is covered by the 2-rank gloo test in test_dist_cpu.py.)"""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

from _dvae_pkg import dvae_amd
from dvae_amd import data as D

pytestmark = pytest.mark.gpu


def test_rccl_one_rank_exchange_is_identity_and_overlapped_part_is_launched():
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    import socket
    with socket.socket() as sk:  # a free port: nothing else may be listening on a fixed one
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_PORT"] = str(port)
    os.environ["DVAE_FORCE_COLLECTIVE"] = "1"
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        torch.manual_seed(2)
        ga = D.enas_rows_to_arrays(D.synth_enas_rows(64, 6, 6, 0.3, 4), 8)
        eps = torch.randn(64, 8, device="cuda")
        model = dvae_amd.DVAE(8, 8, 0, 1, hs=40, nz=8, bidirectional=True).cuda().train()
        ref = dvae_amd.DVAE(8, 8, 0, 1, hs=40, nz=8, bidirectional=True).cuda().train()
        ref.load_state_dict(model.state_dict())
        reducer = dvae_amd.dist.GradSumReducer(model)
        assert 0 < reducer.n_early < reducer.numel
        launched = []
        orig = reducer.decoder_grads_ready

        def spy():
            orig()
            launched.append(reducer._early is not None)
        reducer.decoder_grads_ready = spy
        for step in range(2):
            reducer.zero()
            mu, lv = model.encode(ga)
            loss, _, _ = model.loss(mu, lv, ga, eps=eps)
            loss.backward()
            reducer.all_reduce()
            torch.cuda.synchronize()
            assert reducer._early is None
        assert launched == [True, True]
        # the few-float logging all-reduce (loss, recon, kld) over RCCL: one rank -> unchanged, float64
        tot = reducer.reduce_scalars(loss, loss * 0.5, torch.tensor(3.0, device="cuda"))
        assert tot.dtype == torch.float64 and tot.is_cuda
        assert torch.allclose(tot, torch.stack([loss.detach().double(), loss.detach().double() * 0.5,
                                                torch.tensor(3.0, device="cuda", dtype=torch.float64)]))
        ref.zero_grad()
        mu, lv = ref.encode(ga)
        l2, _, _ = ref.loss(mu, lv, ga, eps=eps)
        l2.backward()
        rg = dict(ref.named_parameters())
        for k, p in model.named_parameters():
            if rg[k].grad is None:
                assert p.grad is None
            else:
                e = ((p.grad - rg[k].grad).norm() / (rg[k].grad.norm() + 1e-30)).item()
                assert e < 2e-5, (k, e)
    finally:
        os.environ.pop("DVAE_FORCE_COLLECTIVE", None)
        dist.destroy_process_group()
