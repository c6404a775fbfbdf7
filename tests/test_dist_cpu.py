"""The N > 1 path on CPU with gloo, world_size 2: batch sharding by the reference's
ceil-split rule and ONE SUM all-reduce of the flat gradient buffer must give every
rank the gradient of the whole batch (the loss is a sum over graphs).  The per-rank
gradients come from the oracle here (the HIP kernels need a GPU); the reducer, the
shard rule and the grad=None set are the product's."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import golden_files, load_case, cfg_of, make_model, state_dict_of, rel_err


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, path, out, balance):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from _dvae_pkg import dvae_amd
    import dvae_oracle as O
    d, cls, ctor = load_case(path)
    model = make_model(cls, ctor)
    model.load_state_dict(state_dict_of(d))
    reducer = dvae_amd.dist.GradSumReducer(model)
    ga = dvae_amd.GraphArrays.from_dense(d["types"], d["adj"], d["n_g"])
    if balance:   # cut by the running sum of n_g^2 instead of the graph count
        lo, hi = dvae_amd.balanced_shard_bounds(ga.n_vertices, world, rank)
    else:
        lo, hi = dvae_amd.shard_bounds(len(ga), world, rank)
    mine = dvae_amd.dist.shard(ga, balance=balance)
    assert len(mine) == hi - lo
    types, adj, n_g = (torch.from_numpy(x) for x in mine.to_dense())
    eps = torch.from_numpy(d["eps"])[lo:hi]  # rank r uses its rows of the global normal draw
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    L, _, _, _, _, grads = O.train_step_grads(cfg_of(cls, ctor), sd, types, adj, n_g, eps)
    # a parameter outside the flat buffer (the predictor head of train.py:165-172): its gradient is exchanged too
    head = torch.nn.Linear(3, 1)
    reducer.add_extra(list(head.parameters()))
    for p in head.parameters():
        p.grad = torch.full_like(p, float(rank + 1))
    reducer.zero()
    named = dict(model.named_parameters())
    for k, g in grads.items():
        if g is None:
            assert named[k].grad is None, k           # dead parameters stay grad=None on every rank
        else:
            named[k].grad += g                        # what autograd's accumulation does after backward()
    reducer.all_reduce()
    for p in head.parameters():
        assert torch.equal(p.grad, torch.full_like(p, 3.0))  # 1 + 2 over the two ranks
    loss = reducer.reduce_scalars(L)[0].float()       # the logging all-reduce
    if rank == 0:
        torch.save({"loss": loss, "grads": {k: (p.grad.clone() if p.grad is not None else None)
                                            for k, p in model.named_parameters()}}, out)
    dist.destroy_process_group()


@pytest.mark.parametrize("name,balance", [("dvae_tiny_bi_var", False), ("dvae_tiny_bi_var", True), ("bn_tiny_uni_full", False)])
def test_two_rank_sum_allreduce_equals_single_rank(tmp_path, name, balance):
    path = golden_files(name)[0]
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), path, out, balance), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    d, cls, ctor = load_case(path)
    assert abs(got["loss"].item() - d["loss3"][0]) <= 1e-5 * abs(d["loss3"][0])
    none = sorted(k for k, g in got["grads"].items() if g is None)
    assert none == sorted(x for x in d["none_grads"].tolist() if x)
    for k, g in got["grads"].items():
        if g is not None:
            assert rel_err(g, torch.from_numpy(d["grad::" + k])) < 1e-5, k
