"""Golden vectors for decode() produced by the REFERENCE itself (build container only).

Run as:   PYTHONDONTWRITEBYTECODE=1 python -B oracle/gen_golden_decode.py

Imports /root/reference/models.py unmodified (oracle/igraph_standin stands in for
python-igraph) and calls its `decode(z, stochastic)` (models.py:747-804).  For the stochastic
case the two random sources of that function are fed from recorded uniforms so that the run is
replayable: `np.random.choice(range(nvt), p=probs)` (models.py:766) is answered with numpy's own
inverse-CDF rule applied to u_type[b, idx], and `torch.rand_like(score)` (models.py:783) returns
u_edge[:, idx, vi].  Writes tests/golden/decode_*.npz: weights, z, the uniforms, and the decoded
graphs (types, adjacency, sizes).  Data only; nothing of the reference's source travels.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "igraph_standin"))
sys.path.insert(0, HERE)
sys.path.append("/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import graph_rows  # noqa: E402
import dvae_oracle as O  # noqa: E402
import models as ref_models  # noqa: E402  (the reference)

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


class Replay(object):
    """answers the reference's random draws from recorded uniforms, in call order"""

    def __init__(self, u_type, u_edge, B, n):
        self.u_type, self.u_edge, self.B, self.n = u_type, u_edge, B, n
        self.k_type = 0
        self.pairs = [(idx, vi) for idx in range(1, n) for vi in range(idx - 1, -1, -1)]
        self.k_edge = 0

    def choice(self, a, p=None):
        idx = 1 + self.k_type // self.B
        b = self.k_type % self.B
        self.k_type += 1
        return O.choice_from_uniform(p, self.u_type[b, idx])

    def rand_like(self, t):
        idx, vi = self.pairs[self.k_edge]
        self.k_edge += 1
        return torch.from_numpy(self.u_edge[:, idx, vi].copy()).to(t.dtype).reshape(t.shape)


def run_case(name, cls, ctor, B, stochastic, seed, weight_scale=1.0):
    torch.manual_seed(seed)
    model = getattr(ref_models, cls)(**ctor)
    model.eval()
    if weight_scale != 1.0:  # sharper predictors -> more varied graphs than a fresh init gives
        with torch.no_grad():
            for k, p in model.named_parameters():
                if k.startswith("add_vertex") or k.startswith("add_edge") or k.startswith("fc3"):
                    p.mul_(weight_scale)
    n = ctor["max_n"]
    rng = np.random.default_rng(seed)
    z = torch.from_numpy(rng.standard_normal((B, ctor["nz"])).astype(np.float32))
    u_type = rng.random((B, n)).astype(np.float32)
    u_edge = rng.random((B, n, n)).astype(np.float32)
    rep = Replay(u_type, u_edge, B, n)
    keep = (ref_models.np.random.choice, torch.rand_like)
    ref_models.np.random.choice = rep.choice
    torch.rand_like = rep.rand_like
    try:
        with torch.no_grad():
            G = model.decode(z, stochastic=stochastic)
    finally:
        ref_models.np.random.choice, torch.rand_like = keep
    types, adj, n_g = graph_rows.igraphs_to_arrays(G, n)
    out = {
        "cls": np.array(cls), "ctor_keys": np.array(list(ctor.keys())),
        "ctor_vals": np.array([int(v) for v in ctor.values()], np.int64),
        "stochastic": np.array(int(stochastic)), "z": z.numpy(), "u_type": u_type, "u_edge": u_edge,
        "types": types, "adj": adj, "n_g": n_g,
    }
    for k, w in model.state_dict().items():
        out["sd::" + k] = w.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    # the oracle must reproduce the reference exactly on the same draws
    cfg = O.make_cfg(ctor["max_n"], ctor["nvt"], 0, 1, ctor["hs"], ctor["nz"], ctor["bidirectional"],
                     ctor.get("vid", 1), cls == "DVAE_BN")
    t2, a2, n2 = O.decode(cfg, model.state_dict(), z, stochastic, u_type, u_edge)
    same = np.array_equal(n2, n_g) and np.array_equal(t2, types) and np.array_equal(a2, adj)
    print("%-26s B=%d sizes=%s edges=%d oracle_matches=%s" % (name, B, np.bincount(n_g, minlength=n + 1)[2:].tolist(),
                                                             int(adj.sum()), same))
    assert same, name


def main():
    os.makedirs(OUT, exist_ok=True)
    tiny = dict(max_n=8, nvt=8, START_TYPE=0, END_TYPE=1, hs=20, nz=6, bidirectional=1)
    tiny_bn = dict(max_n=10, nvt=10, START_TYPE=0, END_TYPE=1, hs=20, nz=6, bidirectional=0)
    for stoch in (0, 1):
        tag = "stoch" if stoch else "det"
        run_case("decode_dvae_tiny_%s" % tag, "DVAE", tiny, 24, bool(stoch), 3, 6.0)
        run_case("decode_bn_tiny_%s" % tag, "DVAE_BN", tiny_bn, 24, bool(stoch), 4, 6.0)
    run_case("decode_dvae_tiny_stoch_init", "DVAE", tiny, 16, True, 5, 1.0)


if __name__ == "__main__":
    main()
