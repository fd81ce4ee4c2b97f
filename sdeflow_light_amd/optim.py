"""Fused Adam on flat parameter buckets (K13) — drop-in for the reference's
``torch.optim.Adam(gen_sde.parameters(), lr=lr)`` (MSGM_higherDim.py:792).

Parameters that are consecutive views of one flat buffer (what
``FlatParamMixin`` produces) are updated by ONE kernel launch per step; the
state is exposed per parameter as ``step / exp_avg / exp_avg_sq`` so
``state_dict()`` is wire-compatible with ``torch.optim.Adam`` checkpoints
(NN.py:13-42).  The step counter lives on the device so a captured hipGraph can
replay the update.
"""
from __future__ import annotations

from typing import List

import torch

from . import ops


class _Run:
    __slots__ = ("params", "numel", "m", "v", "group")

    def __init__(self, params, group):
        self.params: List[torch.nn.Parameter] = params
        self.group = group                     # the param_group whose lr / betas / eps this run is updated with
        self.numel = sum(p.numel() for p in params)
        dev = params[0].device
        self.m = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.numel, dtype=torch.float32, device=dev)


def _flat_view(first: torch.Tensor, numel: int) -> torch.Tensor:
    return first.as_strided((numel,), (1,), first.storage_offset())


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale: float = 1.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.grad_scale = grad_scale          # 1/world_size after a sum all-reduce
        self._runs = None
        self._step_dev = None
        self._step_host = 0

    def _build_runs(self):
        runs = []
        for group in self.param_groups:        # a run never spans two groups: each keeps its own hyper-parameters
            cur = []
            for p in group["params"]:
                if not p.requires_grad:
                    continue      # e.g. the horizon T, an nn.Parameter(requires_grad=False) upstream
                if cur and p.data_ptr() == cur[-1].data_ptr() + cur[-1].numel() * 4 and p.device == cur[-1].device:
                    cur.append(p)
                else:
                    if cur:
                        runs.append(_Run(cur, group))
                    cur = [p]
            if cur:
                runs.append(_Run(cur, group))
        if not runs:
            raise ValueError("FusedAdam got no trainable parameters")
        self._runs = runs
        dev = runs[0].params[0].device
        self._step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        for r in runs:
            off = 0
            for p in r.params:
                k = p.numel()
                self.state[p] = {"step": torch.tensor(float(self._step_host)),
                                 "exp_avg": r.m[off:off + k].view(p.shape),
                                 "exp_avg_sq": r.v[off:off + k].view(p.shape)}
                off += k

    @torch.no_grad()
    def step(self, closure=None):
        if self._runs is None:
            self._build_runs()
        ops.counter_inc(self._step_dev)
        self._step_host += 1
        for r in self._runs:
            g = r.group
            lr, (b1, b2), eps = g["lr"], g["betas"], g["eps"]
            p0 = r.params[0]
            contiguous = p0.grad is not None
            if contiguous:
                ptr = p0.grad.data_ptr()
                for p in r.params:
                    if p.grad is None or p.grad.data_ptr() != ptr or not p.grad.is_contiguous():
                        contiguous = False
                        break
                    ptr += p.numel() * 4
            if contiguous:
                ops.adam_step(_flat_view(p0.data, r.numel), _flat_view(p0.grad, r.numel), r.m, r.v, step=0, lr=lr,
                              beta1=b1, beta2=b2, eps=eps, gscale=self.grad_scale, step_dev=self._step_dev)
            else:
                off = 0
                for p in r.params:
                    k = p.numel()
                    if p.grad is not None:
                        ops.adam_step(p.data.view(-1), p.grad.contiguous().view(-1), r.m[off:off + k], r.v[off:off + k],
                                      step=0, lr=lr, beta1=b1, beta2=b2, eps=eps, gscale=self.grad_scale,
                                      step_dev=self._step_dev)
                    off += k
            for p in r.params:
                self.state[p]["step"] = torch.tensor(float(self._step_host))
        return None

    def load_state_dict(self, state_dict):
        if self._runs is None:
            self._build_runs()
        views = {p: (self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]) for r in self._runs for p in r.params}
        super().load_state_dict(state_dict)
        step = 0
        for p, (m, v) in views.items():
            st = self.state[p]
            if "exp_avg" not in st:            # checkpoint of a never-stepped optimizer: empty state
                m.zero_(); v.zero_()
                st["step"] = torch.tensor(0.0)
            else:
                m.copy_(st["exp_avg"]); v.copy_(st["exp_avg_sq"])
                step = int(float(st["step"]))
            st["exp_avg"], st["exp_avg_sq"] = m, v
        self._step_host = step
        self._step_dev.fill_(step)
