"""Shared helpers for the tests."""
import glob
import os

import numpy as np
import torch

from _dvae_pkg import dvae_amd
import dvae_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files(pattern="*"):
    return sorted(glob.glob(os.path.join(GOLDEN, pattern + ".npz")))


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
