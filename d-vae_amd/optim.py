"""Adam for the train step: one HIP launch over a flat parameter buffer.

Same update as `torch.optim.Adam(model.parameters(), lr=1e-4)` of the reference
(`/root/reference/train.py:174`): the parameters the step touches become views of one
flat fp32 buffer (their values, names and `state_dict` are unchanged), gradients are
the views of `dist.GradSumReducer`'s flat buffer, and `step()` is `dvae_adam_step_f32`.
Parameters the step never touches (grad None in the reference: `hv_unify`, the backward
encoder's when unidirectional) are left alone, exactly as torch's Adam skips them.
"""
import ctypes as C

import torch

from . import _lib
from .dist import GradSumReducer


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, reducer=None, extra_params=None, capturable=False):
        """extra_params: parameters outside the hot path (e.g. the optional `model.predictor` head of
        train.py:165-172); they are stepped by a torch.optim.Adam with the same hyper-parameters.
        capturable: the learning rate and the step count live in device memory (dvae_adam_step_dev_f32; the
        counterpart of torch.optim.Adam(capturable=True)), so that a captured train step can be replayed
        (harness.GraphedTrainStep); same arithmetic."""
        self.model = model
        extra = [p for p in (extra_params or [])]
        self.extra = torch.optim.Adam(extra, lr=lr, betas=betas, eps=eps) if extra else None
        self.reducer = reducer if reducer is not None else GradSumReducer(model)
        r = self.reducer
        r.add_extra(extra)  # their gradients join the exchange step (the head's loss is a sum over the rank's graphs too)
        if not r.flat.is_cuda:
            raise _lib.DvaeHipError("FlatAdam runs on a HIP device; move the model first (no CPU fallback)")
        super(FlatAdam, self).__init__(r.params, dict(lr=lr, betas=betas, eps=eps))
        self.flat_p = torch.zeros_like(r.flat)
        with torch.no_grad():
            for p, o in zip(r.params, r.offsets):
                view = self.flat_p[o:o + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view
        self.exp_avg = torch.zeros_like(r.flat)
        self.exp_avg_sq = torch.zeros_like(r.flat)
        self.steps = 0
        self.capturable = bool(capturable)
        self._lr_dev = self._step_dev = None   # device copies of lr / steps (capturable)
        self._dev_mirror = (None, None)        # the (lr, steps) the device copies are known to hold
        model._pw = None  # parameter storage moved: pack afresh

    def zero_grad(self, set_to_none=False):
        self.reducer.zero()
        if self.extra is not None:
            self.extra.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        r = self.reducer
        base_p, base_g = self.flat_p.data_ptr(), r.flat.data_ptr()
        for p, o in zip(r.params, r.offsets):
            # .to() / .cuda() / .float() on the model re-allocate parameters: the views would be gone
            pg = p.grad
            if pg is None or p.data_ptr() != base_p + 4 * o or pg.data_ptr() != base_g + 4 * o:
                raise _lib.DvaeHipError("FlatAdam: a parameter (or its .grad) no longer lives in the flat buffers "
                                        "(was the model moved or its gradients set to None after the optimizer "
                                        "was built?); build the optimizer after the last .to()")
        lib = _lib.load()
        # zero_grad: with the gradient sink attached every write to the flat gradient buffer goes through the model
        # (which clears `reducer.clean`), so the kernel can leave the buffer zeroed and the next zero_grad() is free
        zero = int(bool(getattr(self.model, "_grad_sink", False)) and getattr(self.model, "_reducer", None) is r)
        stream = C.c_void_p(torch._C._cuda_getCurrentRawStream(self.flat_p.device.index))
        if self.capturable:
            self.sync_device_state()
            _lib.check(lib.dvae_adam_step_dev_f32(self.flat_p.data_ptr(), r.flat.data_ptr(), self.exp_avg.data_ptr(),
                                                  self.exp_avg_sq.data_ptr(), self.flat_p.numel(), self._lr_dev.data_ptr(),
                                                  float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                                  self._step_dev.data_ptr(), zero, stream), "dvae_adam_step_dev_f32")
            self.steps += 1
            self._dev_mirror = (self._dev_mirror[0], self.steps)  # the kernel pair advanced the device count
        else:
            self.steps += 1
            _lib.check(lib.dvae_adam_step_f32(self.flat_p.data_ptr(), r.flat.data_ptr(), self.exp_avg.data_ptr(),
                                              self.exp_avg_sq.data_ptr(), self.flat_p.numel(), float(g["lr"]),
                                              float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), self.steps, zero,
                                              stream), "dvae_adam_step_f32")
        # (set either way: with the sink detached autograd's own accumulation writes into the .grad views behind the
        # reducer's back, so the buffer is only known to be zero when THIS step's kernel zeroed it)
        r.clean = bool(zero)
        # the kernel wrote through raw pointers: tensor versions did not move, so the packed copy is stale
        self.model._pw_dirty = True
        if self.extra is not None:
            for eg in self.extra.param_groups:
                eg["lr"] = g["lr"]  # follows a scheduler attached to this optimizer
            self.extra.step()
        return loss

    def sync_device_state(self):
        """capturable: brings the device copies of (lr, steps) up to the host's values where they differ (a scheduler
        changed lr, a checkpoint was loaded).  Two tiny fills at most; nothing when they agree."""
        dev = self.flat_p.device
        if self._lr_dev is None:
            self._lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)
            self._step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
            self._dev_mirror = (None, None)
        lr = float(self.param_groups[0]["lr"])
        if self._dev_mirror[0] != lr:
            self._lr_dev.fill_(lr)
        if self._dev_mirror[1] != self.steps:
            self._step_dev.fill_(self.steps)
        self._dev_mirror = (lr, self.steps)

    def replayed(self):
        """a captured step() was replayed: the device count advanced by one, the host's follows"""
        self.steps += 1
        self._dev_mirror = (self._dev_mirror[0], self.steps)

    # ------------------------------------------------------------------ checkpoints
    # The layout written and read is torch.optim.Adam's own (what the reference saves as
    # optimizer_checkpoint<epoch>.pth, train.py:705-712, and loads back with --continue-from, train.py:185-192):
    #   {"state": {i: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [{..., "params": [0 .. N-1]}]}
    # with i numbering `model.parameters()` in order (the reference builds optim.Adam(model.parameters())),
    # parameters that never received a gradient having no entry.  So a reference run resumes here and a run
    # of this package resumes under torch.optim.Adam.
    def _ordered_params(self):
        params = list(self.model.parameters())
        seen = {id(p) for p in params}
        if self.extra is not None:  # extra parameters that are not sub-modules of the model come last
            params += [p for g in self.extra.param_groups for p in g["params"] if id(p) not in seen]
        return params

    def state_dict(self):
        r = self.reducer
        slot = {id(p): (o, p.numel()) for p, o in zip(r.params, r.offsets)}
        state = {}
        extra_state = {id(k): v for k, v in self.extra.state.items()} if self.extra is not None else {}
        for i, p in enumerate(self._ordered_params()):
            if id(p) in slot:
                if self.steps > 0:
                    o, n = slot[id(p)]
                    state[i] = {"step": torch.tensor(float(self.steps)),
                                "exp_avg": self.exp_avg[o:o + n].view_as(p).clone(),
                                "exp_avg_sq": self.exp_avg_sq[o:o + n].view_as(p).clone()}
            elif id(p) in extra_state:
                state[i] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in extra_state[id(p)].items()}
        probe = torch.zeros(1, requires_grad=True)
        group = dict(torch.optim.Adam([probe]).state_dict()["param_groups"][0])  # this torch version's keys and defaults
        mine = self.param_groups[0]
        group.update(lr=mine["lr"], betas=tuple(mine["betas"]), eps=mine["eps"])
        for k in ("initial_lr",):  # what a scheduler attached to this optimizer left in the group
            if k in mine:
                group[k] = mine[k]
        group["params"] = list(range(len(self._ordered_params())))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        if "names" in sd and "state" not in sd:
            return self._load_flat_state(sd)  # files written by the first version of this class
        if "state" not in sd or "param_groups" not in sd:
            raise ValueError("not an Adam state dict (expected the keys 'state' and 'param_groups')")
        params = self._ordered_params()
        ids = [i for g in sd["param_groups"] for i in g["params"]]
        if len(ids) != len(params):
            raise ValueError("optimizer state covers %d parameters, the model has %d" % (len(ids), len(params)))
        r = self.reducer
        slot = {id(p): (o, p.numel()) for p, o in zip(r.params, r.offsets)}
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = 0
        extra_state = {}
        for i, p in zip(ids, params):
            st = sd["state"].get(i, sd["state"].get(str(i)))
            if st is None:
                continue
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError("optimizer state %d has shape %s, the parameter %s" % (i, tuple(st["exp_avg"].shape), tuple(p.shape)))
            if id(p) in slot:
                o, n = slot[id(p)]
                self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps = max(steps, int(float(st["step"])))
            else:
                extra_state[id(p)] = st  # a parameter outside the hot path (predictor head), or one the step never touches
        self.steps = steps
        saved = sd["param_groups"][0]
        g = self.param_groups[0]
        g.update(lr=saved["lr"], betas=tuple(saved["betas"]), eps=saved["eps"])
        if "initial_lr" in saved:
            g["initial_lr"] = saved["initial_lr"]
        if self.extra is not None:
            own = self.extra.state_dict()
            eparams = [p for eg in self.extra.param_groups for p in eg["params"]]
            own["state"] = {j: extra_state[id(p)] for j, p in enumerate(eparams) if id(p) in extra_state}
            for eg in own["param_groups"]:
                eg.update(lr=saved["lr"], betas=tuple(saved["betas"]), eps=saved["eps"])
            self.extra.load_state_dict(own)

    def _load_flat_state(self, sd):
        if list(sd["names"]) != list(self.reducer.names):
            raise ValueError("optimizer state belongs to a different parameter set")
        self.steps = int(sd["steps"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            g.update(saved)
        if self.extra is not None and sd.get("extra") is not None:
            self.extra.load_state_dict(sd["extra"])
