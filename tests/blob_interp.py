"""CPU interpreter of the packed batch tables (TEST HELPER).

Evaluates encode + loss forward in plain torch by walking exactly the index
tables that dvae_batch_export() hands to the HIP kernels (same slot layout, same
de-duplicated partial-state rows, same pair / vertex-row tables).  Comparing its
result with the oracle checks the host-side packer without a GPU.
"""
import sys, os
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dvae_oracle as O  # noqa: E402

ORDER = ["types", "pred_off", "pred_idx", "succ_off", "succ_idx", "hgf_off", "hgf_idx", "hgb_off", "hgb_idx",
         "p_off", "p_idx", "p_type", "pt_off", "pt_idx", "rowsA", "rowsB", "pair_a", "pair_h", "pair_c", "pair_t",
         "ca_off", "ca_row", "ca_pairs", "ch_off", "ch_row", "ch_pairs", "cc_off", "cc_row",
         "vs_off", "vs_idx", "vst_off", "vst_idx", "ce_off", "ce_tgt"]


def unpack_blob(batch):
    c = batch.counts
    sizes = [c.nB, c.nB + 1, c.E, c.nB + 1, c.E, c.B + 1, c.HGF, c.B + 1, c.HGB, c.PT + 1, c.PL, c.PT, c.nB + 1, c.PL,
             c.NA, c.NB, c.NP, c.NP, c.NP, (c.NP + 3) // 4, c.CA + 1, c.CA, c.NP, c.CH + 1, c.CH, c.NP, c.CC + 1, c.CC,
             c.R + 1, c.RL, c.nB + 1, c.RL, c.R + 1, c.NCE]
    blob = batch.blob_host.numpy()
    out, o = {}, 0
    for name, n in zip(ORDER, sizes):
        a = blob[o:o + n]
        if name == "pair_t":
            a = a.view(np.uint8)[:c.NP]
        out[name] = a.copy()
        o += (n + 3) & ~3
    assert o == c.total_ints
    return out


def _csr_sum(src, off, idx, rows):
    out = torch.zeros(rows, src.shape[1], dtype=src.dtype)
    for r in range(rows):
        for k in range(off[r], off[r + 1]):
            out[r] = out[r] + src[idx[k]]
    return out


def forward(cfg, sd, batch, eps, beta=0.005):
    """-> mu, logvar, (loss, res, kld) evaluated through the tables."""
    t = unpack_blob(batch)
    c = batch.counts
    B, n, nB, T = c.B, c.n, c.nB, c.T
    dt = sd["fc1.weight"].dtype
    H = cfg.hs
    types = torch.from_numpy(t["types"]).long()

    def onehot(x):
        return torch.nn.functional.one_hot(x, T).to(dt)

    def gated(direction, h, lvl):
        return O._gated(sd, direction, O._msg_input(cfg, h, lvl, None, dt))

    gtab = {}
    if cfg.is_bn:
        for d_ in ("forward", "backward"):
            gtab[d_] = O._gated(sd, d_, torch.eye(T, dtype=dt))

    def chain(cell, direction, off, idx, h0=None):
        Hs = torch.zeros(nB, H, dtype=dt)
        msg = torch.zeros(nB, H, dtype=dt)
        order = range(n) if direction == "forward" else range(n - 1, -1, -1)
        for step, v in enumerate(order):
            r0 = v * B
            if cfg.is_bn:
                agg = _csr_sum(gtab[direction], off[r0:], idx, B)
            else:
                agg = _csr_sum(msg, off[r0:], idx, B)
            if h0 is not None and step == 0:
                agg = h0
            Hs[r0:r0 + B] = O._gru(sd, cell, onehot(types[r0:r0 + B]), agg)
            if not cfg.is_bn:
                msg[r0:r0 + B] = gated(direction, Hs[r0:r0 + B], v)
        return Hs, msg

    Hf, _ = chain("grue_forward", "forward", t["pred_off"], t["pred_idx"])
    Hg = _csr_sum(Hf, t["hgf_off"], t["hgf_idx"], B)
    if cfg.bidir:
        Hb, _ = chain("grue_backward", "backward", t["succ_off"], t["succ_idx"])
        Hg = O._linear(sd, "hg_unify.0", torch.cat([Hg, _csr_sum(Hb, t["hgb_off"], t["hgb_idx"], B)], 1))
    mu, logvar = O._linear(sd, "fc1", Hg), O._linear(sd, "fc2", Hg)

    z = O.reparameterize(mu, logvar, eps, True)
    H0 = torch.tanh(O._linear(sd, "fc3", z))
    Hd, msg = chain("grud", "forward", t["pred_off"], t["pred_idx"], h0=H0)
    # partial rows
    src = gtab["forward"] if cfg.is_bn else msg
    aggp = _csr_sum(src, t["p_off"], t["p_idx"], c.PT)
    Hp_ = O._gru(sd, "grud", onehot(torch.from_numpy(t["p_type"]).long()), aggp)
    ST = torch.cat([Hd, Hp_], 0)
    W1 = sd["add_edge.0.weight"]
    b1 = sd["add_edge.0.bias"]
    EA = torch.cat([ST[torch.from_numpy(t["rowsA"]).long()] @ W1[:, :H].t(), torch.zeros(1, W1.shape[0], dtype=dt)], 0)
    EB = torch.cat([ST[torch.from_numpy(t["rowsB"]).long()] @ W1[:, H:2 * H].t(), torch.zeros(1, W1.shape[0], dtype=dt)], 0)
    pre = EA[torch.from_numpy(t["pair_a"]).long()] + EB[torch.from_numpy(t["pair_h"]).long()] + b1
    if cfg.is_bn:
        EC = H0 @ W1[:, 2 * H:].t()
        pre = pre + EC[torch.from_numpy(t["pair_c"]).long()]
    s = O._linear(sd, "add_edge.2", torch.relu(pre))
    prob = torch.sigmoid(s)[:, 0]
    tgt = torch.from_numpy(t["pair_t"].astype(np.float32)).to(dt)
    res = torch.nn.functional.binary_cross_entropy(prob, tgt, reduction="sum")
    VS = _csr_sum(ST, t["vs_off"], t["vs_idx"], c.R)
    logits = O._linear(sd, "add_vertex.2", torch.relu(O._linear(sd, "add_vertex.0", VS)))
    lsm = torch.log_softmax(logits, 1)
    for r in range(c.R):
        for k in range(t["ce_off"][r], t["ce_off"][r + 1]):
            res = res - lsm[r, t["ce_tgt"][k]]
    kld = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())
    return mu, logvar, (res + beta * kld, res, kld)
