"""Oracle: sliced-score-matching loss, parameter gradients and Adam.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Two statements of the same loss are kept:
* ``ssm_loss_double_backward`` follows the reference literally
  (SDEs.py:616-646: ``autograd.grad(mu_to_div, y, v, create_graph=True)``);
* ``ssm_loss_jvp`` is the forward-mode form the HIP path computes
  (``v^T (d mu/dy) v = (J_mu v).v``); tests assert both agree, and the
  golden fixtures pin both against the imported reference.
"""
from __future__ import annotations

import math
from typing import Callable, Dict

import torch

from . import sde_ref as S

Params = Dict[str, torch.Tensor]
ScoreFn = Callable[[Params, torch.Tensor, torch.Tensor], torch.Tensor]   # (params, y, t)->a


def _mu_to_div(spec: S.SdeSpec, score: ScoreFn, params: Params, t, y):
    """a, and mu - 1/2 divSigma with mu = g a - f + divSigma (lambda=0).
    SDEs.py:624-632,560-561."""
    a = score(params, y, t.squeeze())
    g = S.diffusion_g(spec, t, y)
    mu = S.apply_sigma(spec, g, a) - S.drift_f(spec, t, y) + S.div_sigma(spec, t, y)
    return a, mu - 0.5 * S.div_sigma(spec, t, y)


def ssm_loss_double_backward(spec, score: ScoreFn, params: Params, t, y, v, create_graph=True):
    """Per-sample loss exactly as upstream.  SDEs.py:616-646."""
    y = y.detach().clone().requires_grad_(True)
    a, m = _mu_to_div(spec, score, params, t, y)
    Jtv = torch.autograd.grad(m, y, v, create_graph=create_graph)[0]
    mMu = (Jtv * v).reshape(y.shape[0], -1).sum(1)
    mNu = (a ** 2).reshape(y.shape[0], -1).sum(1) / 2
    return mMu + mNu


def ssm_loss_jvp(spec, score: ScoreFn, params: Params, t, y, v):
    """Same scalar via forward mode; differentiable once w.r.t. params."""
    def f(yy):
        a, m = _mu_to_div(spec, score, params, t, yy)
        return m, a
    m, Jv, a = torch.func.jvp(f, (y,), (v,), has_aux=True)
    mMu = (Jv * v).reshape(y.shape[0], -1).sum(1)
    mNu = (a ** 2).reshape(y.shape[0], -1).sum(1) / 2
    return mMu + mNu


def ssm_mean_and_grads(spec, score: ScoreFn, params: Params, t, y, v, form="jvp"):
    """loss.mean() and d/dparams — the quantities one training step needs
    (MSGM_higherDim.py:807-808).  Returns (loss_scalar, per_sample, grads)."""
    leaf = {k: p.detach().clone().requires_grad_(True) for k, p in params.items()}
    fn = ssm_loss_jvp if form == "jvp" else ssm_loss_double_backward
    per = fn(spec, score, leaf, t, y, v)
    loss = per.mean()
    names = list(leaf)
    gs = torch.autograd.grad(loss, [leaf[k] for k in names], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(leaf[k])) for k, g in zip(names, gs)}
    return loss.detach(), per.detach(), grads


def adam_step(p, g, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam defaults as used upstream (MSGM_higherDim.py:792):
    no weight decay, no amsgrad.  ``step`` is the 1-based step count after
    the increment.  Returns new (p, m, v)."""
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v
