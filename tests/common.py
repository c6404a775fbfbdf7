"""Shared helpers for the tests."""
import glob
import os

import numpy as np
import torch

from _dvae_pkg import dvae_amd
import dvae_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files(pattern="*"):
    """train-step fixtures by default; the generation fixtures are named decode_*"""
    files = sorted(glob.glob(os.path.join(GOLDEN, pattern + ".npz")))
    if not pattern.startswith("decode_"):
        files = [f for f in files if not os.path.basename(f).startswith("decode_")]
    return files


def load_case(path):
    d = np.load(path)  # allow_pickle=False (default): plain arrays only
    ctor = dict(zip(d["ctor_keys"].tolist(), [int(x) for x in d["ctor_vals"].tolist()]))
    cls = str(d["cls"])
    return d, cls, ctor


def cfg_of(cls, ctor):
    return O.make_cfg(ctor["max_n"], ctor["nvt"], ctor["START_TYPE"], ctor["END_TYPE"], ctor["hs"], ctor["nz"],
                      ctor["bidirectional"], ctor.get("vid", 1), cls == "DVAE_BN")


def make_model(cls, ctor, seed=1):
    torch.manual_seed(seed)
    kw = dict(ctor)
    kw["bidirectional"] = bool(kw["bidirectional"])
    if "vid" in kw:
        kw["vid"] = bool(kw["vid"])
    return getattr(dvae_amd, cls)(**kw)


def state_dict_of(d, dtype=torch.float32):
    return {k[4:]: torch.from_numpy(d[k]).to(dtype) for k in d.files if k.startswith("sd::")}


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def oracle_truth(cls, ctor, sd, types, adj, n_g, eps):
    """fp64 oracle (the truth) and fp32 oracle (what the reference computes: it is
    pinned to the reference's golden vectors by tests/test_oracle_golden.py)."""
    cfg = cfg_of(cls, ctor)
    t = [torch.as_tensor(x) for x in (types, adj, n_g)]
    r32 = O.train_step_grads(cfg, {k: v.float() for k, v in sd.items()}, t[0], t[1], t[2], eps.float())
    r64 = O.train_step_grads(cfg, {k: v.double() for k, v in sd.items()}, t[0], t[1], t[2], eps.double())
    return r32, r64


def grad_error_table(model, r32, r64):
    """per parameter tensor: relative error of the HIP gradient and of the fp32 reference arithmetic, both
    against the fp64 oracle -- the two numbers assert_grads_match compares"""
    table = {}
    for k, p in model.named_parameters():
        g64 = r64[5][k]
        if g64 is None or p.grad is None:
            continue
        table[k] = {"e_hip": rel_err(p.grad, g64), "e_ref": rel_err(r32[5][k], g64), "numel": p.numel()}
    return table


def log_parity(name, payload):
    """leaves a JSON record under gpurun_out/parity/ (merged back from the GPU box; the summaries that are
    judged are copied to profiles/)"""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out", "parity")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, name + ".json"), "w") as f:
            json.dump(payload, f, indent=1, sort_keys=True)
    except OSError:
        pass  # read-only tree: the record is a by-product, never a reason to fail


_TRUTH_CACHE = {}


def oracle_truth_cached(key, cls, ctor, sd, types, adj, n_g, eps):
    """oracle_truth, computed once per fixture and process (the fp64 oracle at n = 64, hs = 1024 takes a minute)"""
    if key not in _TRUTH_CACHE:
        _TRUTH_CACHE[key] = oracle_truth(cls, ctor, sd, types, adj, n_g, eps)
    return _TRUTH_CACHE[key]


def assert_grads_match(model, r32, r64, tol=1e-4, slack=1.5):
    """Gradient parity.  The decoder's ReLU layers make the gradient discontinuous:
    a pre-activation within fp32 rounding of 0 flips its mask, and ONE flip moves
    a gradient tensor by ~1e-3 relative.  The reference's own fp32 run therefore
    sits up to ~2e-3 away from its exact (fp64) value at hs=501 (measured:
    DESIGN.md 'Parity').  The bar per parameter is
        rel_err(HIP, fp64 oracle) <= tol + slack * rel_err(fp32 oracle, fp64 oracle)
    i.e. the HIP gradient is as close to the exact gradient as the reference's is;
    where no flip occurs this is the plain `tol` = 1e-4 of north_star."""
    worst = 0.0
    for k, p in model.named_parameters():
        g64 = r64[5][k]
        if g64 is None:
            assert p.grad is None, "%s must have no gradient" % k
            continue
        assert p.grad is not None, k
        e_ref = rel_err(r32[5][k], g64)
        e_hip = rel_err(p.grad, g64)
        worst = max(worst, e_hip)
        assert e_hip <= tol + slack * e_ref, "%s: hip-vs-exact %.3e, reference-vs-exact %.3e" % (k, e_hip, e_ref)
    return worst
