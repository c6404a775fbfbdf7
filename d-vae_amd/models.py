"""Drop-in `DVAE` / `DVAE_BN` modules whose encode / loss / backward run on
hand-written gfx950 HIP kernels (libdvae_hip.so, C-ABI in include/dvae_hip.h).

Mirrors the class API of the reference (`/root/reference/models.py`):
  DVAE.__init__            models.py:537-604   same sub-modules, same order ->
                                               identical state_dict and seeded init
  DVAE_BN.__init__         models.py:980-1000
  _collate_fn              models.py:631-632   (returns a packed DAGBatch)
  encode                   models.py:720-731   (BN 1068-1080)
  reparameterize           models.py:733-740
  loss                     models.py:806-860
  forward                  models.py:867-870
  get_device               models.py:606-609

There is no PyTorch/CPU fallback for the compute: without the HIP library and a
HIP device, encode()/loss() raise.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .batching import DAGBatch, GraphArrays, graphs_to_arrays

# state_dict order == C-ABI parameter order (include/dvae_hip.h, DVAE_NUM_PARAMS)
PARAM_NAMES = [
    "grue_forward.weight_ih", "grue_forward.weight_hh", "grue_forward.bias_ih", "grue_forward.bias_hh",
    "grue_backward.weight_ih", "grue_backward.weight_hh", "grue_backward.bias_ih", "grue_backward.bias_hh",
    "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias",
    "grud.weight_ih", "grud.weight_hh", "grud.bias_ih", "grud.bias_hh",
    "fc3.weight", "fc3.bias",
    "add_vertex.0.weight", "add_vertex.0.bias", "add_vertex.2.weight", "add_vertex.2.bias",
    "add_edge.0.weight", "add_edge.0.bias", "add_edge.2.weight", "add_edge.2.bias",
    "gate_forward.0.weight", "gate_forward.0.bias", "gate_backward.0.weight", "gate_backward.0.bias",
    "mapper_forward.0.weight", "mapper_backward.0.weight",
    "hv_unify.0.weight", "hv_unify.0.bias", "hg_unify.0.weight", "hg_unify.0.bias",
]
assert len(PARAM_NAMES) == _lib.NUM_PARAMS
_IDX = {k: i for i, k in enumerate(PARAM_NAMES)}

_ENC_FWD = ["grue_forward.weight_ih", "grue_forward.weight_hh", "grue_forward.bias_ih", "grue_forward.bias_hh",
            "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias",
            "gate_forward.0.weight", "gate_forward.0.bias", "mapper_forward.0.weight"]
_ENC_BWD = ["grue_backward.weight_ih", "grue_backward.weight_hh", "grue_backward.bias_ih", "grue_backward.bias_hh",
            "gate_backward.0.weight", "gate_backward.0.bias", "mapper_backward.0.weight",
            "hg_unify.0.weight", "hg_unify.0.bias"]
_DEC = ["grud.weight_ih", "grud.weight_hh", "grud.bias_ih", "grud.bias_hh", "fc3.weight", "fc3.bias",
        "add_vertex.0.weight", "add_vertex.0.bias", "add_vertex.2.weight", "add_vertex.2.bias",
        "add_edge.0.weight", "add_edge.0.bias", "add_edge.2.weight", "add_edge.2.bias",
        "gate_forward.0.weight", "gate_forward.0.bias", "mapper_forward.0.weight"]


def _stream():
    # (torch.cuda.current_stream() builds a Stream object: ~8 us a call, six calls per train step; the raw handle is 0.3 us)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def _require_gpu(t, what):
    if not t.is_cuda:
        raise _lib.DvaeHipError(
            "%s: parameters are on %s; the D-VAE hot path only runs on a HIP device "
            "(move the model with .to('cuda')). There is no CPU fallback." % (what, t.device))


class _EncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, batch, *params):
        lib = _lib.load()
        dev = params[0].device
        pw = model._packed_weights()
        batch.wait_ready()
        B = batch.B
        n_ws = lib.dvae_encode_ws_floats(C.byref(model._desc), C.byref(batch.counts))
        if n_ws < 0:
            _lib.check(-1, "dvae_encode_ws_floats")
        ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
        mu = torch.empty(B, model.nz, dtype=torch.float32, device=dev)
        logvar = torch.empty(B, model.nz, dtype=torch.float32, device=dev)
        _lib.check(lib.dvae_encode_forward(C.byref(model._desc), C.byref(batch.counts), batch.blob.data_ptr(),
                                           pw.data_ptr(), ws.data_ptr(), mu.data_ptr(), logvar.data_ptr(), _stream()),
                   "dvae_encode_forward")
        ctx.model, ctx.batch, ctx.ws, ctx.pw = model, batch, ws, pw
        ctx.n_params = len(params)
        model._pw_hold(pw, ctx)  # this graph reads pw again in its backward pass
        return mu, logvar

    @staticmethod
    def backward(ctx, dmu, dlogvar):
        lib = _lib.load()
        model, batch = ctx.model, ctx.batch
        if ctx.ws is None:
            raise RuntimeError("encode(): backward through the graph a second time (the workspace of the forward pass "
                               "is released by the first backward; retain_graph is not supported)")
        dev = ctx.ws.device
        dmu = torch.zeros(batch.B, model.nz, device=dev) if dmu is None else dmu.contiguous().float()
        dlogvar = torch.zeros(batch.B, model.nz, device=dev) if dlogvar is None else dlogvar.contiguous().float()
        pg = torch.empty(model._n_packed, dtype=torch.float32, device=dev)
        _lib.check(lib.dvae_encode_backward(C.byref(model._desc), C.byref(batch.counts), batch.blob.data_ptr(),
                                            ctx.pw.data_ptr(), ctx.ws.data_ptr(), dmu.data_ptr(), dlogvar.data_ptr(),
                                            pg.data_ptr(), _stream()), "dvae_encode_backward")
        grads = model._unpack(ctx.pw, pg, model._enc_names)
        model._pw_release(ctx)
        ctx.ws = ctx.pw = None  # release the workspace and the packed weights (which can then be re-packed in place)
        return (None, None) + tuple(grads)


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, batch, eps, eps_scale, training, beta, mu, logvar, *params):
        lib = _lib.load()
        dev = params[0].device
        pw = model._packed_weights()
        batch.wait_ready()
        mu_c, lv_c = mu.contiguous().float(), logvar.contiguous().float()
        n_ws = lib.dvae_loss_ws_floats(C.byref(model._desc), C.byref(batch.counts))
        if n_ws < 0:
            _lib.check(-1, "dvae_loss_ws_floats")
        ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
        out3 = torch.empty(3, dtype=torch.float32, device=dev)
        _lib.check(lib.dvae_loss_forward(C.byref(model._desc), C.byref(batch.counts), batch.blob.data_ptr(),
                                         pw.data_ptr(), ws.data_ptr(), mu_c.data_ptr(), lv_c.data_ptr(),
                                         eps.data_ptr() if eps is not None else None, float(eps_scale), int(training),
                                         float(beta), out3.data_ptr(), _stream()), "dvae_loss_forward")
        ctx.model, ctx.batch, ctx.ws, ctx.pw = model, batch, ws, pw
        ctx.args = (mu_c, lv_c, eps, float(eps_scale), int(training), float(beta))
        model._pw_hold(pw, ctx)
        return out3

    @staticmethod
    def backward(ctx, g3):
        lib = _lib.load()
        model, batch = ctx.model, ctx.batch
        mu_c, lv_c, eps, eps_scale, training, beta = ctx.args
        if ctx.ws is None:
            raise RuntimeError("loss(): backward through the graph a second time (the workspace of the forward pass "
                               "is released by the first backward; retain_graph is not supported)")
        dev = ctx.ws.device
        g3 = g3.contiguous().float()
        pg = torch.empty(model._n_packed, dtype=torch.float32, device=dev)
        dmu = torch.empty_like(mu_c)
        dlogvar = torch.empty_like(lv_c)
        _lib.check(lib.dvae_loss_backward(C.byref(model._desc), C.byref(batch.counts), batch.blob.data_ptr(),
                                          ctx.pw.data_ptr(), ctx.ws.data_ptr(), mu_c.data_ptr(), lv_c.data_ptr(),
                                          eps.data_ptr() if eps is not None else None, eps_scale, training, beta,
                                          g3.data_ptr(), dmu.data_ptr(), dlogvar.data_ptr(), pg.data_ptr(), _stream()),
                   "dvae_loss_backward")
        grads = model._unpack(ctx.pw, pg, model._dec_names)
        model._pw_release(ctx)
        ctx.ws = ctx.pw = None
        if model._grad_sink and getattr(model, "_reducer", None) is not None:
            model._reducer.decoder_grads_ready()
        return (None, None, None, None, None, None, dmu, dlogvar) + tuple(grads)


class DVAE(nn.Module):
    """DAG variational autoencoder (reference `models.py:536-875`)."""
    _is_bn = False

    def __init__(self, max_n, nvt, START_TYPE, END_TYPE, hs=501, nz=56, bidirectional=False, vid=True,
                 mfma_dtype="fp32"):
        super(DVAE, self).__init__()
        self.mfma_dtype = mfma_dtype
        self.max_n = max_n
        self.nvt = nvt
        self.START_TYPE = START_TYPE
        self.END_TYPE = END_TYPE
        self.hs = hs
        self.nz = nz
        self.gs = hs
        self.bidir = bidirectional
        self.vid = vid
        self.device = None
        self.vs = hs + max_n if self.vid else hs

        # same construction order as the reference (identical RNG consumption)
        self.grue_forward = nn.GRUCell(nvt, hs)
        self.grue_backward = nn.GRUCell(nvt, hs)
        self.fc1 = nn.Linear(self.gs, nz)
        self.fc2 = nn.Linear(self.gs, nz)
        self.grud = nn.GRUCell(nvt, hs)
        self.fc3 = nn.Linear(nz, hs)
        self.add_vertex = nn.Sequential(nn.Linear(hs, hs * 2), nn.ReLU(), nn.Linear(hs * 2, nvt))
        self.add_edge = nn.Sequential(nn.Linear(hs * 2, hs * 4), nn.ReLU(), nn.Linear(hs * 4, 1))
        self.gate_forward = nn.Sequential(nn.Linear(self.vs, hs), nn.Sigmoid())
        self.gate_backward = nn.Sequential(nn.Linear(self.vs, hs), nn.Sigmoid())
        self.mapper_forward = nn.Sequential(nn.Linear(self.vs, hs, bias=False))
        self.mapper_backward = nn.Sequential(nn.Linear(self.vs, hs, bias=False))
        if self.bidir:
            self.hv_unify = nn.Sequential(nn.Linear(hs * 2, hs))
            self.hg_unify = nn.Sequential(nn.Linear(self.gs * 2, self.gs))
        self.relu = nn.ReLU()
        self.sigmoid = nn.Sigmoid()
        self.tanh = nn.Tanh()
        self.logsoftmax1 = nn.LogSoftmax(1)
        self._finish_init()

    # ------------------------------------------------------------------ plumbing
    def _finish_init(self):
        if self.mfma_dtype not in ("fp32", "bf16"):
            raise ValueError("mfma_dtype must be 'fp32' or 'bf16'")
        self._desc = _lib.ModelDesc(self.max_n, self.nvt, self.hs, self.nz, int(bool(self.bidir)),
                                    int(bool(self.vid) and not self._is_bn), int(self._is_bn),
                                    int(self.START_TYPE), int(self.END_TYPE), int(self.mfma_dtype == "bf16"))
        if self.max_n > 64 or self.max_n < 2:
            raise ValueError("max_n must be in 2..64")
        if self.nvt > 32:
            raise ValueError("nvt must be <= 32")
        self._n_packed = None
        self._pw = None
        self._pw_key = None
        self._pw_dirty = False
        self._pw_holders = None  # weak set of the autograd contexts that still read self._pw in their backward pass
        self._named_cache = None
        self._host_cache = {}  # derived from _named_cache: parameter lists per name list, validated .grad tensors (dropped with it)
        self._grad_sink = getattr(self, "_grad_sink", False)
        self._reducer = getattr(self, "_reducer", None)
        self._enc_names = _ENC_FWD + (_ENC_BWD if self.bidir else [])
        self._dec_names = list(_DEC)
        self.eps_override = None  # test hook: fixed normal draw for reparameterize

    def __setattr__(self, name, value):
        if isinstance(value, (nn.Module, nn.Parameter)):
            self.__dict__["_named_cache"] = None  # a (re)assigned sub-module changes the parameter set
            self.__dict__["_host_cache"] = {}
        super(DVAE, self).__setattr__(name, value)

    def _named(self):
        """name -> Parameter, cached (walking the module tree costs ~100 us per call, several
        times per step); dropped when modules are converted / re-assigned."""
        if self._named_cache is None:
            self._named_cache = dict(self.named_parameters())
            self._host_cache = {}
        return self._named_cache

    def _param_table(self):
        named = self._named()
        t = self._host_cache.get("table")
        if t is None:
            t = self._host_cache["table"] = [named.get(k) for k in PARAM_NAMES]
        return t

    def _hot_params(self, names):
        named = self._named()
        key = ("hot", id(names))
        t = self._host_cache.get(key)
        if t is None:
            t = self._host_cache[key] = [named[k] for k in names]
        return t

    def _pw_hold(self, pw, ctx):
        """`ctx` (an autograd context of _EncodeFn / _LossFn) reads the packed buffer `pw` again in its backward
        pass: until that has run -- or the graph is dropped and the context collected -- the buffer must not be
        re-packed in place.  Held weakly: a graph that is thrown away without backward() releases by itself."""
        import weakref
        if pw is not self._pw:
            return
        if self._pw_holders is None:
            self._pw_holders = weakref.WeakSet()
        self._pw_holders.add(ctx)

    def _pw_release(self, ctx):
        if self._pw_holders is not None:
            self._pw_holders.discard(ctx)

    def _packed_weights(self):
        """Zero-padded device copy of the weights, rebuilt when any parameter changed (optimizer steps bump
        tensor versions; FlatAdam, which writes through raw pointers, sets `_pw_dirty`).  The buffer is kept
        and re-packed in place (one launch, padding stays zero) unless a live autograd graph still holds it
        for its backward pass (`_pw_hold`): then a fresh one is built."""
        lib = _lib.load()
        table = self._param_table()
        live = self._host_cache.get("live")
        if live is None:
            live = self._host_cache["live"] = [p for p in table if p is not None]
        first = live[0]
        _require_gpu(first, "DVAE")
        # (two flat lists, compared as lists: this runs twice per train step and the host is the bound at batch 32)
        key = ([p.data_ptr() for p in live], [p._version for p in live], [p is None for p in table])
        if self._pw is not None and key == self._pw_key and not self._pw_dirty and self._pw.device == first.device:
            return self._pw
        if self._n_packed is None:
            self._n_packed = int(lib.dvae_packed_floats(C.byref(self._desc)))
            if self._n_packed < 0:
                _lib.check(-1, "dvae_packed_floats")
        for p in table:
            if p is not None and (p.dtype != torch.float32 or not p.is_contiguous()):
                raise _lib.DvaeHipError("parameters must be contiguous float32")
        arr = (C.c_void_p * _lib.NUM_PARAMS)(*[p.data_ptr() if p is not None else None for p in table])
        # same set of parameters as the buffer was packed from (absent ones leave zeros), and no autograd graph that
        # was built on it still waits for its backward pass
        reuse = (self._pw is not None and self._pw.device == first.device and self._pw_key is not None and
                 key[2] == self._pw_key[2] and not self._pw_holders)
        if reuse:
            _lib.check(lib.dvae_repack_weights(C.byref(self._desc), arr, self._pw.data_ptr(), _stream()), "dvae_repack_weights")
        else:
            pw = torch.empty(self._n_packed, dtype=torch.float32, device=first.device)
            _lib.check(lib.dvae_pack_weights(C.byref(self._desc), arr, pw.data_ptr(), _stream()), "dvae_pack_weights")
            self._pw = pw
            self._pw_holders = None  # contexts of the old buffer keep it alive themselves (ctx.pw)
        self._pw_key, self._pw_dirty = key, False
        return self._pw

    def _unpack(self, pw, pg, names):
        """Packed gradient buffer -> gradients of `names`.  Default: fresh tensors, returned to
        autograd.  With a gradient sink attached (`attach_grad_sink`): added in place to the
        parameters' existing .grad storage and None is returned for each, so autograd runs no
        accumulation kernels of its own."""
        lib = _lib.load()
        named = self._named()
        sink = self._grad_sink
        if sink:
            # fast path: the same .grad tensors as at the last call (identity), validated then: reuse the pointer arrays
            c = self._host_cache.get(("unpack", id(names)))
            if c is not None and c[4] == pg.device:
                params, gs = c[0], c[1]
                for i in range(len(params)):
                    if params[i].grad is not gs[i]:
                        c = None
                        break
            if c is not None and c[4] == pg.device:
                if self._reducer is not None:
                    self._reducer.clean = False  # the flat gradient buffer is about to be added to
                _lib.check(lib.dvae_unpack_grads(C.byref(self._desc), pw.data_ptr(), pg.data_ptr(), c[2], c[3], 1, _stream()),
                           "dvae_unpack_grads")
                return c[5]
        ptrs = [None] * _lib.NUM_PARAMS
        want = [0] * _lib.NUM_PARAMS
        grads = []
        for k in names:
            if sink:
                g = named[k].grad
                if g is None or not g.is_contiguous() or g.dtype != torch.float32 or g.device != pg.device:
                    raise _lib.DvaeHipError("gradient sink: %s has no contiguous float32 .grad on %s" % (k, pg.device))
                grads.append(None)
            else:
                g = torch.empty_like(named[k])
                grads.append(g)
            ptrs[_IDX[k]] = g.data_ptr()
            want[_IDX[k]] = 1
        if sink and self._reducer is not None:
            self._reducer.clean = False  # the flat gradient buffer is about to be added to
        arr = (C.c_void_p * _lib.NUM_PARAMS)(*ptrs)
        w = (C.c_uint8 * _lib.NUM_PARAMS)(*want)
        _lib.check(lib.dvae_unpack_grads(C.byref(self._desc), pw.data_ptr(), pg.data_ptr(), arr, w, int(bool(sink)),
                                         _stream()), "dvae_unpack_grads")
        if sink:
            params = [named[k] for k in names]
            self._host_cache[("unpack", id(names))] = (params, [p.grad for p in params], arr, w, pg.device, grads)
        return grads

    def attach_grad_sink(self, on=True, reducer=None):
        """Gradient sink: backward() adds the gradients of the hot-path parameters straight into
        their preallocated .grad tensors (e.g. the views of `dist.GradSumReducer`'s flat buffer)
        instead of returning them to autograd.  Same .grad contents after loss.backward(); the
        difference is only visible to torch.autograd.grad(), which then sees no gradient."""
        self._grad_sink = bool(on)
        self._reducer = reducer if on else None  # told when the decoder's gradients are final
        return self

    def _as_batch(self, G, caps=None):
        if isinstance(G, DAGBatch) or getattr(G, "is_dag_batch", False):
            return G
        if not isinstance(G, (list, tuple, GraphArrays)):
            G = [G]  # single graph, as models.py:722-723
        arrays = graphs_to_arrays(G, self.max_n)
        return DAGBatch(arrays, self.nvt, self.START_TYPE, self._is_bn, self.get_device(), caps=caps)

    def set_mfma_dtype(self, mfma_dtype):
        """'fp32' (default: every contraction exact fp32, the parity mode) or 'bf16' (matrix-unit
        operands rounded to bf16, fp32 accumulation; weights and activations stay fp32)."""
        self.mfma_dtype = mfma_dtype
        keep = self.eps_override
        self._finish_init()
        self.eps_override = keep
        return self

    # ------------------------------------------------------------------ reference API
    def get_device(self):
        if self.device is None:
            self.device = next(self.parameters()).device
        return self.device

    def _apply(self, fn, *a, **kw):
        self.device = None
        self._pw = None
        self._pw_holders = None
        self._named_cache = None
        return super(DVAE, self)._apply(fn, *a, **kw)

    def _collate_fn(self, G, caps=None):
        """Packs the batch once; encode() and loss() both accept the result.  `caps` (batching.padding_caps): pack
        shape-stable, for harness.GraphedTrainStep."""
        return self._as_batch(G, caps)

    def encode(self, G):
        batch = self._as_batch(G)
        params = self._hot_params(self._enc_names)
        return _EncodeFn.apply(self, batch, *params)

    def reparameterize(self, mu, logvar, eps_scale=0.01):
        if self.training:
            std = logvar.mul(0.5).exp_()
            eps = torch.randn_like(std) * eps_scale
            return eps.mul(std).add_(mu)
        return mu

    def loss(self, mu, logvar, G_true, beta=0.005, eps=None):
        """(res + beta*kld, res, kld).  `eps` optionally injects the standard-normal
        draw of reparameterize (shape [B, nz]); default: torch.randn on the device."""
        batch = self._as_batch(G_true)
        if mu.shape[0] != batch.B:
            raise ValueError("mu has %d rows, batch has %d graphs" % (mu.shape[0], batch.B))
        if eps is None:
            eps = self.eps_override
        if self.training:
            if eps is None:
                eps = torch.randn(batch.B, self.nz, device=mu.device, dtype=torch.float32)
            eps = eps.to(device=mu.device, dtype=torch.float32).contiguous()
        else:
            eps = None
        params = self._hot_params(self._dec_names)
        out3 = _LossFn.apply(self, batch, eps, 0.01, self.training, beta, mu, logvar, *params)
        return out3[0], out3[1], out3[2]

    def forward(self, G):
        batch = self._as_batch(G)
        mu, logvar = self.encode(batch)
        loss, _, _ = self.loss(mu, logvar, batch)
        return loss

    def decode_arrays(self, z, stochastic=True, uniforms=None):
        """decode() without the host-side graph objects: returns a GraphArrays (types,
        predecessor masks, sizes).  `uniforms` = (u_type [B, max_n], u_edge [B, max_n, max_n])
        injects the random draws; default: torch.rand on the device."""
        lib = _lib.load()
        pw = self._packed_weights()
        dev = pw.device
        z = z.detach().to(device=dev, dtype=torch.float32).contiguous()
        if z.dim() != 2 or z.shape[1] != self.nz:
            raise ValueError("z must be [B, %d]" % self.nz)
        B, n = z.shape[0], self.max_n
        if B == 0:
            return GraphArrays(torch.zeros(0, n).numpy(), torch.zeros(0, n).numpy(), torch.zeros(0).numpy(), n)
        u_type = u_edge = None
        if stochastic:
            if uniforms is None:
                u_type = torch.rand(B, n, device=dev, dtype=torch.float32)
                u_edge = torch.rand(B, n, n, device=dev, dtype=torch.float32)
            else:
                u_type = torch.as_tensor(uniforms[0]).to(device=dev, dtype=torch.float32).contiguous()
                u_edge = torch.as_tensor(uniforms[1]).to(device=dev, dtype=torch.float32).contiguous()
                if tuple(u_type.shape) != (B, n) or tuple(u_edge.shape) != (B, n, n):
                    raise ValueError("uniforms must be ([B, max_n], [B, max_n, max_n])")
        n_ws = lib.dvae_decode_ws_floats(C.byref(self._desc), B)
        if n_ws < 0:
            _lib.check(-1, "dvae_decode_ws_floats")
        ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
        types = torch.empty(B, n, dtype=torch.int32, device=dev)
        pmask = torch.empty(B, n, dtype=torch.int64, device=dev)
        n_g = torch.empty(B, dtype=torch.int32, device=dev)
        _lib.check(lib.dvae_decode(C.byref(self._desc), B, pw.data_ptr(), ws.data_ptr(), z.data_ptr(), int(bool(stochastic)),
                                   u_type.data_ptr() if stochastic else None, u_edge.data_ptr() if stochastic else None,
                                   types.data_ptr(), pmask.data_ptr(), n_g.data_ptr(), _stream()), "dvae_decode")
        return GraphArrays(types.cpu().numpy(), pmask.cpu().numpy().view("uint64"), n_g.cpu().numpy(), n)

    def decode(self, z, stochastic=True, uniforms=None):
        """Latent points -> list of graphs (reference `models.py:747-804`).  The graphs are
        `graph.DAG` objects (igraph's surface: vcount, vs['type'], get_adjlist, ...)."""
        return self.decode_arrays(z, stochastic, uniforms).to_graphs()

    def encode_decode(self, G):
        """reference `models.py:862-865`"""
        mu, logvar = self.encode(G)
        z = self.reparameterize(mu, logvar)
        return self.decode(z)

    def generate_sample(self, n):
        """reference `models.py:872-875`"""
        sample = torch.randn(n, self.nz).to(self.get_device())
        return self.decode(sample)


class DVAE_BN(DVAE):
    """D-VAE for Bayesian networks (reference `models.py:979-1085`)."""
    _is_bn = True

    def __init__(self, max_n, nvt, START_TYPE, END_TYPE, hs=501, nz=56, bidirectional=False, mfma_dtype="fp32"):
        super(DVAE_BN, self).__init__(max_n, nvt, START_TYPE, END_TYPE, hs, nz, bidirectional, mfma_dtype=mfma_dtype)
        # re-created after the base constructor, in the reference's order (extra RNG draws)
        self.mapper_forward = nn.Sequential(nn.Linear(self.nvt, hs, bias=False))
        self.mapper_backward = nn.Sequential(nn.Linear(self.nvt, hs, bias=False))
        self.gate_forward = nn.Sequential(nn.Linear(self.nvt, hs), nn.Sigmoid())
        self.gate_backward = nn.Sequential(nn.Linear(self.nvt, hs), nn.Sigmoid())
        self.add_edge = nn.Sequential(nn.Linear(hs * 3, hs), nn.ReLU(), nn.Linear(hs, 1))
        self._finish_init()
