"""``FusedAdam``: ``torch.optim.Adam`` (amsgrad off) with the step as one streaming HIP kernel per 48 tensors (include/ag_optim.h, csrc/ag_optim.hip).

The reference builds ``torch.optim.Adam(avatar_net.parameters(), lr)`` and steps it once per iteration (main_avatar.py:60-66, 248-250).  Same
hyper-parameters, same update, same ``state_dict()`` layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter, so ``optm.pt`` written by either loads in
the other: main_avatar.py:777-813); torch's own fused kernel moves the step's 28 bytes per parameter at 3.6 TB/s on MI355X, this one at ~5.5.
fp32 parameters on one GPU only -- anything else raises (there is no fallback)."""
from __future__ import annotations

import ctypes
import math

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, maximize=False):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("FusedAdam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, maximize=maximize))
        self._calls = {}        # per parameter group: the ALLOCATED argument structures (every field is refilled on every step)

    def _state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            for p in ps:
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_cuda or p.device != dev or p.grad.is_sparse:
                    raise RuntimeError("FusedAdam: dense fp32 parameters and gradients on one GPU only")
                if not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be contiguous")
            states = [self._state(p) for p in ps]
            b1, b2 = group["betas"]
            # steps are counted per parameter, as torch.optim.Adam does (the reference's pretraining pass leaves the colour network without gradients)
            bcs = {}
            for st in states:
                t = float(st["step"]) + 1.0
                if t not in bcs:
                    bcs[t] = (1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t))
            # The argument structures are allocated once per (group, tensor count); every pointer in them is REFILLED from the live tensors on
            # every step.  (Round 5 cached the filled structures keyed on id(p): a parameter whose storage was replaced -- p.data = ...,
            # module.to() -- or an optimizer state replaced without load_state_dict would have been updated through stale addresses.)
            n_calls = (len(ps) + _lib.AG_ADAM_MAX_TENSORS - 1) // _lib.AG_ADAM_MAX_TENSORS
            cached = self._calls.get(gi)
            if cached is None or len(cached) != n_calls:
                cached = self._calls[gi] = [_lib.AgAdamArgs() for _ in range(n_calls)]
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            with _lib.on_device(dev):
                for ci, a in enumerate(cached):
                    c0 = ci * _lib.AG_ADAM_MAX_TENSORS
                    chunk, sts = ps[c0:c0 + _lib.AG_ADAM_MAX_TENSORS], states[c0:c0 + _lib.AG_ADAM_MAX_TENSORS]
                    a.n = len(chunk)
                    a.maximize = int(bool(group["maximize"]))
                    a.lr, a.beta1, a.beta2, a.eps, a.weight_decay = float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"])
                    a.one_minus_beta1, a.one_minus_beta2 = 1.0 - float(b1), 1.0 - float(b2)          # in double, rounded once
                    keep = []                       # contiguous copies must outlive the launch (the allocator would hand their block to the next copy)
                    for i, (p, st) in enumerate(zip(chunk, sts)):
                        m, v = st["exp_avg"], st["exp_avg_sq"]
                        if not (m.is_contiguous() and v.is_contiguous() and m.device == dev and v.device == dev
                                and m.dtype == torch.float32 and v.dtype == torch.float32 and m.numel() == p.numel() and v.numel() == p.numel()):
                            raise RuntimeError("FusedAdam: optimizer state must be contiguous fp32 of the parameter's size on the parameters' device")
                        a.param[i], a.exp_avg[i], a.exp_avg_sq[i], a.numel[i] = p.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
                        a.bias_correction1[i], a.bias_correction2_sqrt[i] = bcs[float(st["step"]) + 1.0]
                        g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                        keep.append(g)
                        a.grad[i] = g.data_ptr()
                    _lib.check(L.ag_adam_step(ctypes.byref(a), stream), "ag_adam_step")
                    del keep
            for st in states:
                st["step"] += 1.0
            # the kernel wrote the parameters behind autograd's back: bump their version counters, as torch.optim.Adam's in-place ops do (anything
            # keyed on a parameter's version -- the frozen-weight cache of inference, grouped.py -- must see the update)
            torch.autograd.graph.increment_version(ps)
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._calls = {}
        for st in self.state.values():            # torch.optim.Adam(fused=True) keeps `step` on the device: bring it to the host
            if "step" in st and torch.is_tensor(st["step"]):
                st["step"] = st["step"].detach().to("cpu", torch.float32).reshape(())
