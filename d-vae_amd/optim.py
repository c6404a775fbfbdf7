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
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, reducer=None, extra_params=None):
        """extra_params: parameters outside the hot path (e.g. the optional `model.predictor` head of
        train.py:165-172); they are stepped by a torch.optim.Adam with the same hyper-parameters."""
        self.model = model
        extra = [p for p in (extra_params or [])]
        self.extra = torch.optim.Adam(extra, lr=lr, betas=betas, eps=eps) if extra else None
        self.reducer = reducer if reducer is not None else GradSumReducer(model)
        r = self.reducer
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
        model._pw = None  # parameter storage moved: repack

    def zero_grad(self, set_to_none=False):
        self.reducer.zero()
        if self.extra is not None:
            self.extra.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        r = self.reducer
        for p, o in zip(r.params, r.offsets):
            # .to() / .cuda() / .float() on the model re-allocate parameters: the views would be gone
            if p.data_ptr() != self.flat_p.data_ptr() + 4 * o or p.grad is None or \
                    p.grad.data_ptr() != r.flat.data_ptr() + 4 * o:
                raise _lib.DvaeHipError("FlatAdam: a parameter (or its .grad) no longer lives in the flat buffers "
                                        "(was the model moved or its gradients set to None after the optimizer "
                                        "was built?); build the optimizer after the last .to()")
        self.steps += 1
        lib = _lib.load()
        _lib.check(lib.dvae_adam_step_f32(self.flat_p.data_ptr(), self.reducer.flat.data_ptr(), self.exp_avg.data_ptr(),
                                          self.exp_avg_sq.data_ptr(), self.flat_p.numel(), float(g["lr"]),
                                          float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), self.steps,
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)), "dvae_adam_step_f32")
        # the kernel wrote through raw pointers: tensor versions did not move, so drop the packed copy
        self.model._pw = None
        if self.extra is not None:
            for eg in self.extra.param_groups:
                eg["lr"] = g["lr"]  # follows a scheduler attached to this optimizer
            self.extra.step()
        return loss

    def state_dict(self):
        return {"steps": self.steps, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "names": list(self.reducer.names), "offsets": list(self.reducer.offsets),
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
                "extra": self.extra.state_dict() if self.extra is not None else None}

    def load_state_dict(self, sd):
        if list(sd["names"]) != list(self.reducer.names):
            raise ValueError("optimizer state belongs to a different parameter set")
        self.steps = int(sd["steps"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            g.update(saved)
        if self.extra is not None and sd.get("extra") is not None:
            self.extra.load_state_dict(sd["extra"])
