"""The oracle (oracle/dvae_oracle.py) against golden vectors produced by the
reference itself (oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from common import golden_files, load_case, cfg_of, make_model, state_dict_of, rel_err
import dvae_oracle as O

TINY = golden_files("*tiny*")
BIG = golden_files("*cfg*")


def test_fixtures_present():
    assert len(TINY) == 9 and len(BIG) == 5


@pytest.mark.parametrize("path", TINY, ids=[os.path.basename(p)[:-4] for p in TINY])
@pytest.mark.parametrize("mode", ["dense", "refstyle"])
def test_oracle_matches_reference_tiny(path, mode):
    d, cls, ctor = load_case(path)
    cfg = cfg_of(cls, ctor)
    sd = state_dict_of(d)
    types, adj, n_g = (torch.from_numpy(d[k]) for k in ("types", "adj", "n_g"))
    L, R, K, mu, lv, grads = O.train_step_grads(cfg, sd, types, adj, n_g, torch.from_numpy(d["eps"]), mode=mode)
    ref = d["loss3"]
    assert abs(L.item() - ref[0]) <= 2e-6 * abs(ref[0])
    assert abs(R.item() - ref[1]) <= 2e-6 * abs(ref[1])
    assert abs(K.item() - ref[2]) <= 1e-5 * abs(ref[2]) + 1e-6
    assert (mu - torch.from_numpy(d["mu"])).abs().max() < 2e-6
    assert (lv - torch.from_numpy(d["logvar"])).abs().max() < 2e-6
    none = sorted(k for k, g in grads.items() if g is None)
    assert none == sorted(x for x in d["none_grads"].tolist() if x)
    for k, g in grads.items():
        if g is not None:
            assert rel_err(g, torch.from_numpy(d["grad::" + k])) < 5e-6, k
    assert abs(_adam_delta(sd, grads) - float(d["adam_delta_abs_sum"])) <= 1e-4 * float(d["adam_delta_abs_sum"])


def _adam_delta(sd, grads):
    """sum |delta parameter| of ONE torch.optim.Adam step (lr 1e-4, train.py:174) on the oracle's gradients: the
    checksum oracle/gen_golden.py stored after the reference's own optimizer step"""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    for k, p in params.items():
        p.grad = None if grads.get(k) is None else grads[k].detach().clone().to(p.dtype)
    torch.optim.Adam(list(params.values()), lr=1e-4).step()
    return sum((p.detach().double() - sd[k].double()).abs().sum().item() for k, p in params.items())


@pytest.mark.parametrize("path", BIG, ids=[os.path.basename(p)[:-4] for p in BIG])
def test_oracle_matches_reference_hs501(path):
    """BASELINE config-1 shape: weights are re-derived from the seed through the
    product module (same construction order as the reference), checked by checksums."""
    d, cls, ctor = load_case(path)
    cfg = cfg_of(cls, ctor)
    model = make_model(cls, ctor, int(d["seed"]))
    assert list(model.state_dict().keys()) == d["param_names"].tolist()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k, w in sd.items():
        s = d["psum::" + k]
        assert abs(w.double().sum().item() - s[0]) < 1e-9 and abs(w.double().abs().sum().item() - s[1]) < 1e-9, k
    types, adj, n_g = (torch.from_numpy(d[k]) for k in ("types", "adj", "n_g"))
    L, R, K, mu, lv, grads = O.train_step_grads(cfg, sd, types, adj, n_g, torch.from_numpy(d["eps"]), mode="dense")
    ref = d["loss3"]
    assert abs(L.item() - ref[0]) <= 5e-6 * abs(ref[0])
    assert abs(K.item() - ref[2]) <= 1e-4 * abs(ref[2])
    assert (mu - torch.from_numpy(d["mu"])).abs().max() < 5e-6
    for k, g in grads.items():
        if g is None:
            assert k in d["none_grads"].tolist()
            continue
        gn = d["gnorm::" + k]
        assert abs(g.double().norm().item() - gn[0]) <= 2e-5 * gn[0] + 1e-9, k
        sl = torch.from_numpy(d["gslice::" + k])
        assert (g.reshape(-1)[:sl.numel()] - sl).abs().max() <= 2e-5 * max(1e-3, sl.abs().max().item()) + 1e-7, k
    assert abs(_adam_delta(sd, grads) - float(d["adam_delta_abs_sum"])) <= 1e-3 * float(d["adam_delta_abs_sum"])


DECODE = golden_files("decode_*")


@pytest.mark.parametrize("path", DECODE, ids=[os.path.basename(p)[:-4] for p in DECODE])
def test_oracle_decode_reproduces_reference_graphs(path):
    """oracle.decode against graphs the reference itself decoded (oracle/gen_golden_decode.py:
    same weights, same latent points, the reference's random draws replayed): exact."""
    d, cls, ctor = load_case(path)
    sd = state_dict_of(d)
    types, adj, n_g = O.decode(cfg_of(cls, ctor), sd, torch.from_numpy(d["z"]), bool(int(d["stochastic"])),
                               d["u_type"], d["u_edge"])
    assert np.array_equal(n_g, d["n_g"]) and np.array_equal(types, d["types"]) and np.array_equal(adj, d["adj"])
