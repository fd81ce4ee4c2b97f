"""ORACLE (test infrastructure only — never imported by the product): CPU restatement of the reference's
reporting metrics next to the hot path.

* RBF-kernel MMD between two sample sets — quantitative_comparison.py:22-46:
  k(x, y) = exp(-mean_d((x - y)^2) / d),  mmd = mean(Kxx) + mean(Kyy) - 2 mean(Kxy).
* Gaussian latent log-density of the additive SDE — SDEs.py:209-215.
* ELBO slice estimate — SDEs.py:708-721, NN.py:124-129 (SGM; the multiplicative SDE's branch needs a sklearn
  KDE, SDEs.py:503-509, and is out of scope).

Pinned by tests/golden/g15_metrics.npz (tools/make_golden.py, generated from the imported reference).
"""
from __future__ import annotations

import math

import torch

from . import sde_ref as S
from . import ssm_ref as LR

LOG2PI = float(math.log(2 * math.pi))


def rbf_kernel(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """(Nx, Ny) matrix exp(-mean_d((x_i - y_j)^2) / d) — compute_kernel, quantitative_comparison.py:22-36."""
    d = x.shape[1]
    diff = x.unsqueeze(1) - y.unsqueeze(0)
    return torch.exp(-(diff.pow(2).mean(2) / float(d)))


def mmd(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """compute_mmd, quantitative_comparison.py:38-46."""
    return rbf_kernel(x, x).mean() + rbf_kernel(y, y).mean() - 2 * rbf_kernel(x, y).mean()


def log_normal(x, mean, log_var, eps=0.00001):
    """SDEs.py:213-215."""
    return -(x - mean) ** 2 / (2.0 * torch.exp(log_var) + eps) - log_var / 2.0 - 0.5 * LOG2PI


def sgm_log_latent_pdf(yT):
    """SDEs.py:209-211: standard normal log-density per element (with the reference's +1e-5 in the denominator)."""
    return log_normal(yT, torch.zeros_like(yT), torch.zeros_like(yT))


def elbo_sgm(spec: S.SdeSpec, score, params, x, u_t, eps, u_v, eps_T):
    """elbo_random_t_slice for the additive SDE (SDEs.py:708-721) with its draws injected:
    u_t, eps, u_v = the three draws of ssm(x); eps_T = the noise of cond_latent_sample (sample at t = T).
    (The second sample_txy of :717 only provides the shape of t_; its draws do not enter the value.)"""
    t = S.clamp_time(spec, u_t)
    y = S.vp_perturb(spec, t, x, eps)
    v = S.rademacher_from_uniform(u_v)
    per = LR.ssm_loss_jvp(spec, score, params, t, y, v)
    qt = 1.0 / spec.T
    yT = S.vp_perturb(spec, torch.ones_like(t) * spec.T, x, eps_T)
    lp = sgm_log_latent_pdf(yT).view(x.size(0), -1).sum(1)
    return lp - per / qt
