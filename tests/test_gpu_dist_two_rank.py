"""N = 2 on the HIP path: two processes share the box's one card, each runs the real kernels on its
shard of the batch (the reference's ceil-split, util.py:566-572), the flat gradient buffer is SUMMED
between them (gloo moves it here; on a node it is RCCL, the same `GradSumReducer.all_reduce()` call),
FlatAdam steps.  After two steps both ranks must hold the parameters a single process gets from the
whole batch: the N-rank parity rule of SURVEY.md section 8(e), on the product path rather than the
oracle's gradients (test_dist_cpu.py covers those)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

B, STEPS = 96, 2


def _inputs():
    from _dvae_pkg import dvae_amd
    from dvae_amd import data as D
    ga = D.enas_rows_to_arrays(D.synth_enas_rows(B, 6, 6, 0.3, 11), 8)
    g = torch.Generator().manual_seed(5)
    eps = torch.randn(STEPS, B, 8, generator=g)
    torch.manual_seed(3)
    model = dvae_amd.DVAE(8, 8, 0, 1, hs=40, nz=8, bidirectional=True)
    return dvae_amd, ga, eps, model


def _train(dvae_amd, model, ga, eps, lo, hi, reducer_on):
    model = model.cuda().train()
    reducer = dvae_amd.dist.GradSumReducer(model)
    opt = dvae_amd.optim.FlatAdam(model, lr=1e-3, reducer=reducer)
    mine = ga[lo:hi]
    losses = []
    for s in range(STEPS):
        opt.zero_grad()
        mu, lv = model.encode(mine)
        loss, _, _ = model.loss(mu, lv, mine, eps=eps[s, lo:hi].cuda())
        loss.backward()
        if reducer_on:
            reducer.all_reduce()
        opt.step()
        losses.append(reducer.reduce_scalars(loss)[0].item() if reducer_on else loss.item())
    torch.cuda.synchronize()
    return losses, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dvae_amd, ga, eps, model = _inputs()
        lo, hi = dvae_amd.shard_bounds(len(ga), world, rank)
        assert len(dvae_amd.dist.shard(ga)) == hi - lo
        losses, sd = _train(dvae_amd, model, ga, eps, lo, hi, True)
        torch.save({"losses": losses, "sd": sd}, "%s.%d" % (out, rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_the_hip_path_equal_one_rank(tmp_path):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "rank")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    dvae_amd, ga, eps, model = _inputs()
    ref_losses, ref_sd = _train(dvae_amd, model, ga, eps, 0, len(ga), False)
    got = [torch.load("%s.%d" % (out, r), weights_only=True) for r in range(2)]
    for r in range(2):
        for a, b in zip(got[r]["losses"], ref_losses):
            assert abs(a - b) <= 2e-5 * abs(b), (r, a, b)      # fp32 sums in a different order
    for k, v in ref_sd.items():
        assert torch.equal(got[0]["sd"][k], got[1]["sd"][k]), k  # the ranks stay bit-identical replicas
        # Adam divides by sqrt(v): at lr 1e-3 a parameter moves by ~1e-3 a step whatever its gradient's size, so
        # the comparison is absolute, against the step length
        d = (got[0]["sd"][k] - v).abs().max().item()
        assert d < 2e-5, (k, d)
