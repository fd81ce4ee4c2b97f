"""Oracle: SDE schedules, drift/diffusion and integrators (CPU, plain torch).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  A functional
restatement of /root/reference/SDEs.py and /root/reference/sde_scheme.py
with every random draw turned into an explicit argument so that the HIP
kernels can be fed the very same noise.

Reference citations are ``file:line`` relative to /root/reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Optional

import torch

# --------------------------------------------------------------------------
# SDE description
# --------------------------------------------------------------------------

SGM = "sgm"                  # SGMsde                      SDEs.py:161-215
MSGM_DENSE = "msgm_dense"    # MSGMsde(denseTensor=True)   SDEs.py:221-251
MSGM_SPARSE = "msgm_sparse"  # MSGMsde(denseTensor=False)  SDEs.py:247-251,369-399


@dataclass
class SdeSpec:
    """Plain-data description of a base SDE (the reference keeps these as
    attributes of ``SDE``/``SGMsde``/``MSGMsde`` objects, SDEs.py:54-64)."""
    kind: str = SGM
    beta_min: float = 0.1
    beta_max: float = 20.0
    T: float = 1.0
    t_epsilon: float = 1e-3
    num_steps_forward: int = 16
    n: int = 0                                  # state dimension (MSGM)
    G: Optional[torch.Tensor] = None            # (n,n,n) dense tensor, SDEs.py:315-341
    L_G: Optional[torch.Tensor] = field(default=None)  # (n,n) Ito correction, SDEs.py:246

    def __post_init__(self):
        if self.kind == MSGM_DENSE and self.G is not None and self.L_G is None:
            # L_G[i,m] = 1/2 sum_{j,k} G[i,j,k] G[j,m,k]        SDEs.py:246
            self.L_G = 0.5 * torch.einsum("ijk,jmk->im", self.G, self.G)


def beta(spec: SdeSpec, t: torch.Tensor) -> torch.Tensor:
    """Linear schedule beta(t) = b0 + (b1-b0) t.  SDEs.py:72-73."""
    return spec.beta_min + (spec.beta_max - spec.beta_min) * t


def vp_mean_weight(spec: SdeSpec, t: torch.Tensor) -> torch.Tensor:
    """exp(-1/4 t^2 (b1-b0) - 1/2 t b0).  SDEs.py:177-178."""
    db = spec.beta_max - spec.beta_min
    return torch.exp(-0.25 * t ** 2 * db - 0.5 * t * spec.beta_min)


def vp_var(spec: SdeSpec, t: torch.Tensor) -> torch.Tensor:
    """1 - exp(-1/2 t^2 (b1-b0) - t b0).  SDEs.py:180-181."""
    db = spec.beta_max - spec.beta_min
    return 1.0 - torch.exp(-0.5 * t ** 2 * db - t * spec.beta_min)


def vp_perturb(spec: SdeSpec, t: torch.Tensor, x0: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
    """y_t = eps*sqrt(var(t)) + mean_weight(t)*x0 (closed form, SGM).
    SDEs.py:134-146 with ``randn_like`` replaced by the ``eps`` argument."""
    return eps * vp_var(spec, t) ** 0.5 + vp_mean_weight(spec, t) * x0


def clamp_time(spec: SdeSpec, u: torch.Tensor) -> torch.Tensor:
    """t = u*T, then rows with t <= t_eps are set to t_eps by mask
    arithmetic.  SDEs.py:688-693 with ``torch.rand`` replaced by ``u``."""
    t = u * spec.T
    m = (t <= spec.t_epsilon).to(t.dtype)
    return m * spec.t_epsilon + (1.0 - m) * t


def forward_step_index(spec: SdeSpec, t: torch.Tensor) -> torch.Tensor:
    """Per-row stop index k = trunc((nsf*t)/T) as int32, rows with t>=T are
    forced to nsf.  SDEs.py:89-101 (include_t0=True branch).  Bit-exact
    integer output; the product is formed in the dtype of ``t`` (fp32)."""
    k = torch.trunc(spec.num_steps_forward * t / spec.T).to(torch.int32).reshape(-1)
    k = torch.where(t.reshape(-1) >= spec.T, torch.full_like(k, spec.num_steps_forward), k)
    return k


def rademacher_from_uniform(u: torch.Tensor) -> torch.Tensor:
    """v = 2*[u >= 1/2] - 1.  SDEs.py:514-515."""
    return (u >= 0.5).to(u.dtype) * 2 - 1


def unit_sphere_from_normal(z: torch.Tensor) -> torch.Tensor:
    """z / ||z|| row-wise.  SDEs.py:520-526."""
    return z / torch.linalg.norm(z, dim=1, keepdim=True)


# --------------------------------------------------------------------------
# sparse nearest-neighbour rotation tensor
# --------------------------------------------------------------------------

def sparse_ijkv(n: int):
    """Index/value lists of the sparse MSGM tensor: for each k two entries
    (i,j)=(k,k+1) with +sqrt(2)/2 and (k+1,k) with -sqrt(2)/2 (circular).
    SDEs.py:369-383.  Returns I,J,K (int64) and V (fp32), each (2n,)."""
    k = torch.arange(n, dtype=torch.int64)
    kp = (k + 1) % n
    I = torch.stack([k, kp], dim=1).reshape(-1)
    J = torch.stack([kp, k], dim=1).reshape(-1)
    K = torch.stack([k, k], dim=1).reshape(-1)
    c = 0.5 * torch.sqrt(torch.tensor(2.0, dtype=torch.float32))
    V = torch.stack([c.expand(n), (-c).expand(n)], dim=1).reshape(-1).contiguous()
    return I, J, K, V


def make_dense_G(n: int, gen_normals: torch.Tensor) -> torch.Tensor:
    """Dense skew tensor from n Gaussian matrices F_k (``gen_normals`` is
    (n,n,n) with gen_normals[k] the k-th ``randn(n,n)`` draw):
    G[:,:,k] = (F_k-F_k^T)/2, then scaled so that trace(L_G) = -n/2.
    SDEs.py:315-326."""
    G = torch.zeros(n, n, n, dtype=gen_normals.dtype)
    for k in range(n):
        F = gen_normals[k]
        G[:, :, k] = 0.5 * (F - F.T)
    L = 0.5 * torch.einsum("ijk,jmk->im", G, G)
    return torch.sqrt(-0.5 * n / torch.trace(L)) * G


# --------------------------------------------------------------------------
# drift / diffusion of the base (forward) SDE
# --------------------------------------------------------------------------

def drift_f(spec: SdeSpec, t, y):
    """Ito drift f(t,y).  SGM: -1/2 beta y (SDEs.py:183-184).  MSGM sparse:
    +1/2 beta y; dense: L_G (beta y) (SDEs.py:410-415)."""
    b = beta(spec, t)
    if spec.kind == SGM:
        return -0.5 * b * y
    if spec.kind == MSGM_SPARSE:
        return 0.5 * b * y
    return torch.einsum("ij,bj->bi", spec.L_G, b * y)


def drift_f_strato(spec: SdeSpec, t, y):
    """Stratonovich drift.  SGM: same as f (SDEs.py:186-187); MSGM: 0 (:417-418)."""
    if spec.kind == SGM:
        return -0.5 * beta(spec, t) * y
    return torch.zeros_like(y)


def div_sigma(spec: SdeSpec, t, y):
    """SGM: 0 (SDEs.py:189-190); MSGM: 2 f (:420-421)."""
    if spec.kind == SGM:
        return torch.zeros_like(y)
    return 2 * drift_f(spec, t, y)


def diffusion_g(spec: SdeSpec, t, y):
    """Diffusion in the reference's three layouts:
    SGM (B,n): sqrt(beta)*1 (SDEs.py:192-194); dense (B,n,n):
    einsum('ijk,bj->bik', G, sqrt(beta) y) (:432); sparse (B,2n):
    V * sqrt(beta) * y[:,J] (:425-430)."""
    b = beta(spec, t)
    if spec.kind == SGM:
        return torch.ones_like(y) * b ** 0.5
    if spec.kind == MSGM_DENSE:
        return torch.einsum("ijk,bj->bik", spec.G, (b ** 0.5) * y)
    I, J, K, V = sparse_ijkv(y.shape[1])
    return V.to(y).unsqueeze(0) * ((b ** 0.5) * y[:, J])


def apply_sigma(spec: SdeSpec, sigma, w):
    """sigma . w for the three layouts — the ``dx`` part of EMstep
    (sde_scheme.py:27-38) and of PluginReverseSDE.ga (SDEs.py:565-578)."""
    if spec.kind == MSGM_SPARSE:
        n = w.shape[1]
        I, J, K, V = sparse_ijkv(n)
        prod = sigma * w[:, K]
        dx = torch.zeros_like(w)
        dx.scatter_add_(1, I.unsqueeze(0).expand(w.shape[0], -1), prod)
        return dx
    if sigma.dim() > 2:
        return torch.einsum("bij,bj->bi", sigma, w)
    return sigma * w


def em_increment(spec: SdeSpec, mu, delta, sigma, dW):
    """mu*delta + sigma.dW.  sde_scheme.py:18-40."""
    return mu * delta + apply_sigma(spec, sigma, dW)


# --------------------------------------------------------------------------
# processes seen by the integrators
# --------------------------------------------------------------------------

class ForwardProcess:
    """forward_SDE adaptor.  SDEs.py:30-47."""

    def __init__(self, spec: SdeSpec):
        self.spec = spec

    def drift_strato(self, t, x):
        return drift_f_strato(self.spec, t, x)

    def drift(self, t, x):
        return self.drift_strato(t, x) + 0.5 * div_sigma(self.spec, t, x)

    def sigma(self, t, x):
        return diffusion_g(self.spec, t, x)


class ReverseProcess:
    """Plug-in reverse SDE.  SDEs.py:556-588: mu(t,y) = (1-l/2) g a - f +
    (1-l) divSigma evaluated at s = T - t; sigma = sqrt(1-l) g(s,y);
    mu_Strato = mu - (1-l)/2 divSigma."""

    def __init__(self, spec: SdeSpec, score: Callable, lmbd: float = 0.0):
        self.spec, self.score, self.lmbd = spec, score, lmbd

    def ga(self, s, y):
        g = diffusion_g(self.spec, s, y)
        a = self.score(y, s.squeeze())
        return apply_sigma(self.spec, g, a)

    def ga_m_drift(self, s, y):
        l = self.lmbd
        return (1.0 - 0.5 * l) * self.ga(s, y) - drift_f(self.spec, s, y) \
            + (1.0 - l) * div_sigma(self.spec, s, y)

    def drift(self, t, x):
        return self.ga_m_drift(self.spec.T - t, x)

    def drift_strato(self, t, x):
        return self.drift(t, x) - 0.5 * (1.0 - self.lmbd) * div_sigma(self.spec, self.spec.T - t, x)

    def sigma(self, t, x):
        return (1.0 - self.lmbd) ** 0.5 * diffusion_g(self.spec, self.spec.T - t, x)


# --------------------------------------------------------------------------
# integrators (noise injected: ``noise[i]`` replaces the i-th randn_like)
# --------------------------------------------------------------------------

def _time_grid(T_, num_steps):
    """delta (python float) and fp32 grid ts = linspace(0,1,N+1)*T_.
    sde_scheme.py:58-59."""
    delta = T_ / num_steps
    ts = torch.linspace(0, 1, num_steps + 1) * T_
    return delta, ts


def _renorm(x, norm0):
    """norm_correction.  sde_scheme.py:85-86."""
    return x * (norm0 / torch.norm(x, dim=1))[:, None]


def _integrate(step_fn, proc, x0, num_steps, noise, T_, norm_correction,
               keep_all, include_t0, stop_index):
    spec = proc.spec
    T_ = float(spec.T if T_ is None else T_)
    delta, ts = _time_grid(T_, num_steps)
    x = x0.detach().clone()
    norm0 = torch.norm(x, dim=1) if norm_correction else None
    B = x.shape[0]
    traj = [x.clone()] if (keep_all and include_t0) else []
    kept = torch.zeros_like(x) if stop_index is not None else None
    t = torch.zeros(B, 1, dtype=x.dtype)
    for i in range(num_steps):
        t.fill_(ts[i].item())
        x = step_fn(proc, t, x, delta, noise[i])
        if norm_correction:
            x = _renorm(x, norm0)
        if keep_all:
            traj.append(x.clone())
        elif stop_index is not None:
            sel = (stop_index.reshape(-1) == (i + int(include_t0)))
            kept[sel] = x[sel]
    if keep_all:
        return torch.stack(traj, 0)                 # (N[+1], B, n)   sde_scheme.py:94-95
    if stop_index is not None:
        return kept
    return x


def _em_step(proc, t, x, delta, z):
    mu = proc.drift(t, x)
    sg = proc.sigma(t, x)
    return x + em_increment(proc.spec, mu, delta, sg, delta ** 0.5 * z)     # sde_scheme.py:82-84


def _heun_step(proc, t, x, delta, z):
    sp = proc.spec
    mu1 = proc.drift_strato(t, x)
    s1 = proc.sigma(t, x)
    dW = delta ** 0.5 * z
    xp = x + em_increment(sp, mu1, delta, s1, dW)                            # sde_scheme.py:140-148
    mu2 = proc.drift_strato(t + delta, xp)
    s2 = proc.sigma(t + delta, xp)
    return x + em_increment(sp, mu1 + mu2, delta / 2, s1 + s2, dW / 2)       # :150-156


def _rk4_step(proc, t, x, delta, z):
    sp = proc.spec
    dW = delta ** 0.5 * z                                                    # sde_scheme.py:223-227
    k1 = em_increment(sp, proc.drift_strato(t, x), delta, proc.sigma(t, x), dW)
    xm = x + k1 / 2
    k2 = em_increment(sp, proc.drift_strato(t + delta / 2, xm), delta, proc.sigma(t + delta / 2, xm), dW)
    xm = x + k2 / 2
    k3 = em_increment(sp, proc.drift_strato(t + delta / 2, xm), delta, proc.sigma(t + delta / 2, xm), dW)
    xe = x + k3
    k4 = em_increment(sp, proc.drift_strato(t + delta, xe), delta, proc.sigma(t + delta, xe), dW)
    return x + (k1 + 2 * k2 + 2 * k3 + k4) / 6                               # :229-251


def euler_maruyama(proc, x0, num_steps, noise, T_=None, norm_correction=False,
                   keep_all=False, include_t0=False, stop_index=None):
    """sde_scheme.py:43-99 with ``noise`` (num_steps,B,n) injected."""
    return _integrate(_em_step, proc, x0, num_steps, noise, T_, norm_correction,
                      keep_all, include_t0, stop_index)


def heun(proc, x0, num_steps, noise, T_=None, norm_correction=False,
         keep_all=False, include_t0=False, stop_index=None):
    """sde_scheme.py:101-172."""
    return _integrate(_heun_step, proc, x0, num_steps, noise, T_, norm_correction,
                      keep_all, include_t0, stop_index)


def rk4_stratonovich(proc, x0, num_steps, noise, T_=None, norm_correction=False,
                     keep_all=False, include_t0=False, stop_index=None):
    """sde_scheme.py:174-269 (one shared dW for the four stages)."""
    return _integrate(_rk4_step, proc, x0, num_steps, noise, T_, norm_correction,
                      keep_all, include_t0, stop_index)


# --------------------------------------------------------------------------
# MSGM forward perturbation (no closed form)
# --------------------------------------------------------------------------

def msgm_forward_perturb(spec: SdeSpec, t, x0, noise_main, noise_short):
    """y_t | x0 for the multiplicative SDE.  SDEs.py:78-122,434-436.

    All rows are RK4-integrated with the forward process over the nsf-step
    grid on [0,T]; row b keeps the state after k_b = forward_step_index
    steps.  Rows with k_b == 0 are instead advanced by a single RK4 step of
    length t_b from x0 (SDEs.py:112-117) using ``noise_short[b]``.
    ``noise_main`` is (nsf,B,n).  Returns (y, k)."""
    proc = ForwardProcess(spec)
    k = forward_step_index(spec, t)
    y = rk4_stratonovich(proc, x0, spec.num_steps_forward, noise_main,
                         include_t0=True, stop_index=k)
    for b in range(x0.shape[0]):
        if int(k[b]) == 0:
            yb = rk4_stratonovich(proc, x0[b:b + 1], 1, noise_short[b:b + 1].unsqueeze(0),
                                  T_=float(t.reshape(-1)[b]))
            y[b] = yb[0]
    return y, k


def msgm_latent(r_T: torch.Tensor, u: torch.Tensor, z: torch.Tensor, log_map: bool):
    """Latent sample r*s: r = quantile(r_T,u) (exp(.)-1e-6 under the log
    map), s uniform on the sphere.  SDEs.py:438-451,467-471."""
    r = torch.quantile(r_T, u).reshape(-1, 1)
    if log_map:
        r = torch.exp(r) - 1e-6
    return r * unit_sphere_from_normal(z)
