"""SDE definitions and the sliced-score-matching loss on HIP kernels — host
mirror of the reference's ``SDEs.py`` (same class / method names and argument
meaning: ``forward_SDE``, ``SDE``, ``SGMsde``, ``MSGMsde``, ``PluginReverseSDE``).

What runs where
---------------
* hot path (``ssm``, ``mu``, ``mu_Strato``, ``sample``, the integrators in
  ``sde_scheme``): hand-written gfx950 kernels through the C ABI;
* schedule accessors kept for API compatibility (``beta``, ``mean_weight``,
  ``var``, ``f``, ``g`` ...) are one-line tensor expressions evaluated on the
  tensor's own device — the hot path never calls them;
* out of scope: the sklearn-KDE latent density
  (``log_latent_pdf``), plotting / validation branches, ``ssm_intT``.

RNG: the reference draws from torch's global generator (CPU mt19937 for t,
device generator for the rest, SDEs.py:688,141,515).  Here every draw comes
from a device-resident Philox stream owned by the object (``.rng``); parity
tests inject explicit noise instead.
"""
from __future__ import annotations

import math

from typing import Optional

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from ._lib import MsgmError, PhiloxState


class forward_SDE(nn.Module):
    """Adaptor exposing the forward (noising) process to the integrators
    (SDEs.py:30-47): mu = f_strato + 1/2 divSigma, sigma = g."""

    def __init__(self, base_sde, T):
        super().__init__()
        self.base_sde = base_sde
        self.T = T

    def mu(self, s, y, lmbd=0.):
        return self.mu_Strato(s, y) + 0.5 * self.base_sde.div_Sigma(s, y)

    def mu_Strato(self, s, y, lmbd=0.):
        return self.base_sde.f_strato(s, y)

    def sigma(self, s, y, lmbd=0., sparse=False):
        return self.base_sde.g(s, y, sparse=sparse)


class SDE(nn.Module):
    """Base class (SDEs.py:49-76): linear beta schedule and bookkeeping."""
    kind = L.SDE_SGM

    def __init__(self, beta_min=0.1, beta_max=20.0, T=1.0, t_epsilon=0.001, num_steps_forward=100, device="cpu"):
        super().__init__()
        self.device = torch.device(device)
        self.T = T
        self.beta_min, self.beta_max = beta_min, beta_max
        self.t_epsilon = t_epsilon
        self.num_steps_forward = num_steps_forward
        self.norm_correction = False
        self.sparseTensor = False
        self.rng: Optional[PhiloxState] = None
        self._shard = None           # (first global row, n) of this rank's rows in a data-parallel run

    def to(self, device):
        new = super().to(device)
        new.device = torch.device(device)
        if torch.is_tensor(self.T):
            new.T = self.T.to(device)
        return new

    def T_float(self) -> float:
        return L.host_scalar(self.T)          # cached: no device synchronisation per call (graph-capturable)

    def struct(self) -> L.SdeT:
        """msgm_sde_t for the kernels."""
        return L.sde_struct(self.kind, self.beta_min, self.beta_max, self.T_float(), self.t_epsilon,
                            getattr(self, "G", None) if self.kind == L.SDE_MSGM_DENSE else None,
                            getattr(self, "L_G", None) if self.kind == L.SDE_MSGM_DENSE else None)

    def philox(self, device) -> PhiloxState:
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if self.rng is None or self.rng.state.device != device:
            seed = int(torch.initial_seed()) ^ 0x5DE5
            if self._shard is None:
                # no shard placement given: at least give every rank of a distributed run its own stream
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                    seed = (seed + 0x9E3779B97F4A7C15 * dist.get_rank()) & 0x7FFFFFFFFFFFFFFF
            self.rng = PhiloxState(seed, device)
            if self._shard is not None:
                self.rng.set_shard(*self._shard)
        return self.rng

    def set_shard(self, row_base: int, n: int) -> None:
        """Place this rank's rows at [row_base, ...) of the GLOBAL batch / sample set: with the same seed on every
        rank the sharded run then draws (t, eps, v, dW, latent) exactly as the single-GPU run does for those rows
        (``parallel.shard_rows`` gives row_base).  Without it, ranks of a distributed run get decorrelated seeds."""
        self._shard = (int(row_base), int(n))
        if self.rng is not None:
            self.rng.set_shard(*self._shard)

    def beta(self, t):
        return self.beta_min + (self.beta_max - self.beta_min) * t

    def IJK(self):
        return None, None, None

    # -- forward-process samplers of the base class (SDEs.py:78-146), same names and arguments ----------------
    @torch.no_grad()
    def sample_scheme(self, t, y0, keep_all_samples=False, return_noise=False):
        """y_t | y_0 by RK4 on the forward process (SDEs.py:78-122) — one device-resident masked multi-stop
        integration instead of the per-row Python loop; ``keep_all_samples`` only chose the capture buffer upstream."""
        if return_noise:
            raise NotImplementedError('See the official repository.')
        from .sde_scheme import msgm_forward_perturb
        return msgm_forward_perturb(self, t, y0)

    @torch.no_grad()
    def sample_scheme_allt(self, y0, include_t0=True, keep_all_samples=True, samplesToKeep=None):
        """y_0, y_{t_1}, ..., y_T | y_0 (SDEs.py:124-132)."""
        from .sde_scheme import rk4_stratonovich_sampler
        return rk4_stratonovich_sampler(forward_SDE(self, self.T).to(self.device), y0, num_steps=self.num_steps_forward, lmbd=0,
                                        keep_all_samples=keep_all_samples, samplesToKeep=samplesToKeep, include_t0=include_t0)

    def sample_Song_et_al(self, t, y0, return_noise=False, eps=None):
        """Closed-form y_t | y_0 of the VP SDE (SDEs.py:134-146) in one kernel; with ``return_noise`` also
        (epsilon, std, g(t, y_t)) as upstream."""
        if self.kind != L.SDE_SGM:
            raise MsgmError("sample_Song_et_al is the closed form of the additive (VP) SDE")
        y, e = ops.perturb_vp_at(y0.contiguous().float(), self.struct(), t, eps=eps,
                                 rng=None if eps is not None else self.philox(y0.device), return_eps=True)
        if eps is None:
            self.rng.advance(1)
        if not return_noise:
            return y
        return y, e, self.var(t) ** 0.5, self.g(t, y)

    def sample_debiasing_t(self, shape):
        raise NotImplementedError('See the official repository.')


class SGMsde(SDE):
    """Variance-preserving SDE (SDEs.py:161-215)."""
    kind = L.SDE_SGM

    def __init__(self, beta_min=0.1, beta_max=20.0, T=1.0, t_epsilon=0.001, num_steps_forward=100, device='cpu'):
        super().__init__(beta_min, beta_max, T, t_epsilon, num_steps_forward, device)
        self.name_SDE = "SGM"

    # -- schedule accessors (API compatibility; not on the hot path) ---------
    def mean_weight(self, t):
        return torch.exp(-0.25 * t ** 2 * (self.beta_max - self.beta_min) - 0.5 * t * self.beta_min)

    def var(self, t):
        return 1. - torch.exp(-0.5 * t ** 2 * (self.beta_max - self.beta_min) - t * self.beta_min)

    def f(self, t, y):
        return -0.5 * self.beta(t) * y

    def f_strato(self, t, y):
        return -0.5 * self.beta(t) * y

    def div_Sigma(self, t, y):
        return torch.zeros_like(y)

    def g(self, t, y, sparse=False):
        return torch.ones_like(y) * self.beta(t) ** 0.5

    # -- hot path --------------------------------------------------------------
    @torch.no_grad()
    def sample(self, t, y0, return_noise=False, eps=None):
        """y_t | y_0 in closed form — ONE kernel (K1) instead of ~8 eager ops.
        ``t`` (B,1) is used exactly as given (no clamp, as upstream: SDEs.py:134-146);
        ``eps`` injects the Gaussian draw (parity tests)."""
        if return_noise:
            raise NotImplementedError('See the official repository.')
        y = ops.perturb_vp_at(y0.contiguous().float(), self.struct(), t, eps=eps,
                              rng=None if eps is not None else self.philox(y0.device))
        if eps is None:
            self.rng.advance(1)
        return y

    def latent_sample(self, num_samples, n):
        dev = self.device
        if dev.type != "cuda":
            raise MsgmError("latent_sample draws on the GPU (Philox); move the SDE to a cuda device")
        x = torch.empty(num_samples, n, dtype=torch.float32, device=dev)
        rng = self.philox(dev)
        ops.fill_normal(x, rng, L.RNG_STREAM_USER)
        rng.advance(1)
        return x

    def cond_latent_sample(self, t_, T, x, eps=None):
        return self.sample(torch.ones_like(t_) * T, x, eps=eps)

    @property
    def logvar_mean_T(self):
        """SDEs.py:171-175."""
        return torch.zeros(1), torch.zeros(1)

    def log_normal(self, x, mean, log_var, eps=0.00001):
        """SDEs.py:213-215."""
        return -(x - mean) ** 2 / (2. * torch.exp(log_var) + eps) - log_var / 2. - 0.5 * math.log(2 * math.pi)

    def log_latent_pdf(self, yT):
        """Standard-normal latent log-density per element (SDEs.py:209-211), on the device."""
        return self.log_normal(yT, torch.zeros_like(yT), torch.zeros_like(yT))


class MSGMsde(SDE):
    """Multiplicative SDE dY = G(Y) o dB (SDEs.py:221-509), dense rank-3
    tensor or sparse nearest-neighbour rotation tensor.  The KDE of the radial
    law (sklearn, SDEs.py:240) is out of scope; the ECDF latent sampler is kept."""

    def __init__(self, y0, beta_min=0.1, beta_max=20.0, T=1.0, t_epsilon=0.001, denseTensor=True,
                 norm_sampler="ecdf", norm_map=None, kernel='gaussian', plot_validate=False,
                 num_steps_forward=100, device='cpu', estim_cst_norm_dens_r_T=True, G=None):
        super().__init__(beta_min, beta_max, T, t_epsilon, num_steps_forward, device)
        if norm_sampler != "ecdf":
            raise MsgmError("only norm_sampler='ecdf' is built (the KDE branch is broken upstream, SDEs.py:444)")
        self.sparseTensor = not denseTensor
        self.kind = L.SDE_MSGM_DENSE if denseTensor else L.SDE_MSGM_SPARSE
        self.norm_correction = True
        self.r_T = torch.linalg.norm(y0, dim=1)
        self.norm_map = norm_map
        if norm_map == "log":
            self.r_T = torch.log(self.r_T + 1e-6)
        self.r_T = self.r_T.to(self.device)
        self.norm_sampler = norm_sampler
        self.dim = y0.shape[1]
        self.name_SDE = "MSGM"
        if denseTensor:
            self.G = (G if G is not None else self.new_G(self.dim)).to(self.device).contiguous()
            self.L_G = (0.5 * torch.einsum('ijk, jmk -> im', self.G, self.G)).contiguous()      # SDEs.py:246
        else:
            self.name_SDE += "_sparseTens"
            self.sparse_G(self.dim)
            self.L_G = 0.5 * torch.eye(self.dim, device=self.device)
        if norm_map == "log":
            self.name_SDE += "logNorm"
        self.cst_log_dens = 0

    def to(self, device):
        new = super().to(device)
        new.r_T = self.r_T.to(device)
        if self.sparseTensor:
            for k in ("G_I", "G_J", "G_K", "G_V"):
                setattr(new, k, getattr(self, k).to(device))
        else:
            new.G = self.G.to(device).contiguous()
        new.L_G = self.L_G.to(device).contiguous()
        return new

    @staticmethod
    def new_G(n):
        """n skew-symmetrised Gaussian matrices scaled so trace(L_G) = -n/2
        (SDEs.py:315-341); drawn from torch's CPU generator."""
        G = torch.zeros(n, n, n)
        for k in range(n):
            F = torch.randn(n, n)
            G[:, :, k] = 0.5 * (F - F.T)
        L_G = 0.5 * torch.einsum('ijk, jmk -> im', G, G)
        return torch.sqrt(-0.5 * n / torch.trace(L_G)) * G

    def sparse_G(self, n):
        """Index lists of the sparse tensor (SDEs.py:369-399) — kept for API
        compatibility; the stencil kernel never reads them."""
        k = torch.arange(n, dtype=torch.int64)
        kp = (k + 1) % n
        self.G_I = torch.stack([k, kp], 1).reshape(-1).to(self.device)
        self.G_J = torch.stack([kp, k], 1).reshape(-1).to(self.device)
        self.G_K = torch.stack([k, k], 1).reshape(-1).to(self.device)
        c = 0.5 * torch.sqrt(torch.tensor(2, dtype=torch.float32))
        self.G_V = torch.stack([c.expand(n), (-c).expand(n)], 1).reshape(-1).contiguous().to(self.device)

    def IJK(self):
        return (self.G_I, self.G_J, self.G_K) if self.sparseTensor else (None, None, None)

    def sparse_G_full(self, n):
        """Dense (n,n,n) image of the sparse nearest-neighbour rotation tensor (SDEs.py:343-367), for checks."""
        k = torch.arange(n)
        kp = (k + 1) % n
        c = 0.5 * math.sqrt(2.0)
        G = torch.zeros(n, n, n, device=self.device)
        G[k, kp, k] = c
        G[kp, k, k] = -c
        self.G = G
        return G

    # -- accessors (API compatibility) ----------------------------------------
    def f(self, t, y):
        b = self.beta(t)
        return 0.5 * b * y if self.sparseTensor else torch.einsum('ij, bj -> bi', self.L_G, b * y)

    def f_strato(self, t, y):
        return torch.zeros_like(y)

    def div_Sigma(self, t, y):
        return 2 * self.f(t, y)

    def g(self, t, y, sparse=False):
        b = self.beta(t)
        if sparse:
            return self.G_V.unsqueeze(0) * ((b ** 0.5) * y[:, self.G_J])
        return torch.einsum('ijk, bj -> bik', self.G, (b ** 0.5) * y)

    # -- hot path --------------------------------------------------------------
    @torch.no_grad()
    def sample(self, t, y0, return_noise=False, noise_main=None, noise_short=None):
        """y_t | y_0 by RK4 on the forward process, entirely on the device:
        masked multi-stop integration replaces the per-row Python loop and the
        D2H/H2D round trips of SDEs.py:78-122."""
        if return_noise:
            raise NotImplementedError('See the official repository.')
        from .sde_scheme import msgm_forward_perturb
        return msgm_forward_perturb(self, t, y0, noise_main=noise_main, noise_short=noise_short)

    def gen_radial_distribution(self, num_samples, u=None):
        if u is None:
            u = torch.empty(num_samples, dtype=torch.float32, device=self.device)
            rng = self.philox(self.device)
            ops.fill_uniform(u, rng, L.RNG_STREAM_ROWS)
            rng.advance(1)
        r = torch.quantile(self.r_T, u).reshape(num_samples, 1)                # SDEs.py:442
        if self.norm_map == "log":
            r = torch.exp(r) - 1e-6
        return r

    def latent_sample(self, num_samples, n, u=None, z=None):
        r = self.gen_radial_distribution(num_samples, u)
        s = randu_on_sphere((num_samples, self.dim), self.device, z=z, rng=self.philox(self.device))
        return r * s

    def cond_latent_sample(self, t_, T, x):
        r_x = torch.linalg.norm(x.detach().to(self.device), dim=1).reshape(x.shape[0], 1)
        return r_x * randu_on_sphere((x.shape[0], self.dim), self.device, rng=self.philox(self.device))

    def log_latent_pdf(self, yT):
        raise MsgmError("log_latent_pdf (sklearn KDE) is outside the accelerated hot path")


# ---------------------------------------------------------------------------
def sample_rademacher(shape, device, rng: Optional[PhiloxState] = None, u=None):
    """2[U>=1/2]-1 (SDEs.py:514-515) drawn in one kernel."""
    return ops.rademacher(tuple(shape), device, u=u, rng=rng)


def sample_gaussian(shape, device, rng: PhiloxState):
    return ops.fill_normal(torch.empty(tuple(shape), dtype=torch.float32, device=device), rng, L.RNG_STREAM_USER)


def randu_on_sphere(shape, device, z=None, rng: Optional[PhiloxState] = None):
    """Uniform direction: Gaussian / its norm (SDEs.py:520-526)."""
    if z is None:
        z = torch.empty(tuple(shape), dtype=torch.float32, device=device)
        ops.fill_normal(z, rng, L.RNG_STREAM_USER + 1)
        rng.advance(1)
    return z / ops.row_norm(z.contiguous()).reshape(shape[0], 1)


def sample_v(shape, device, vtype='rademacher', rng: Optional[PhiloxState] = None):
    if vtype == 'rademacher':
        return sample_rademacher(shape, device, rng)
    if vtype in ('normal', 'gaussian'):
        return sample_gaussian(shape, device, rng)
    if vtype == 'uniform':
        return randu_on_sphere(shape, device, rng=rng)
    raise MsgmError(f'vtype {vtype} not supported')


class PluginReverseSDE(nn.Module):
    """Plug-in reverse SDE (SDEs.py:538-729): f <- g a - f, time reversed.

    ``ssm(x)`` is the training entry point.  With an ``MLP`` score net on an
    ``SGMsde`` it runs the fully fused path: perturbation (K1), probe (K14)
    and ONE persistent kernel for forward + tangent + loss + backward (K5),
    which writes the parameter gradients of ``ssm(x).mean()`` straight into
    the flat ``.grad`` bucket.  The returned per-sample tensor carries a
    backward hook that rescales those gradients by ``grad_output.sum()``,
    so the reference loop ``gen_sde.ssm(x).mean().backward()`` works as is
    (uniform reductions only — which is all the driver uses,
    MSGM_higherDim.py:807)."""

    def __init__(self, base_sde, drift_a, T, vtype='rademacher', debias=False, ssm_intT=False, deviceReverseSDE='cpu'):
        super().__init__()
        if ssm_intT:
            raise MsgmError("ssm_intT=True is dead code upstream (undefined global at SDEs.py:700); not built")
        self.base_sde = base_sde.to(deviceReverseSDE)
        self.a = drift_a
        self.T = T.to(deviceReverseSDE)
        self.vtype = vtype
        self.ssm_intT = ssm_intT
        self.debias = debias
        self.deviceReverseSDE = deviceReverseSDE
        self._ws = None

    # ---- drift / diffusion (integrators call the stage kernel directly; these
    # keep the reference's call surface) ---------------------------------------
    def _stage(self, t, y, lmbd, strato):
        T = self.base_sde.T_float()
        B = y.shape[0]
        if torch.is_tensor(t) and t.is_cuda:
            # the reference passes t as a (B,1) device tensor filled with one value (sde_scheme.py:81): keep it on the device
            # — the stage kernel reads its clock from t_dev — instead of a .item() round trip per mu() call
            t0 = t.reshape(-1)[:1].to(device=y.device, dtype=torch.float32).contiguous()
            s = (T - t0).expand(B).contiguous()
            tt, t_dev = 0.0, t0
        else:
            tt, t_dev = (float(t.reshape(-1)[0]) if torch.is_tensor(t) else float(t)), None
            s = torch.full((B,), T - tt, dtype=torch.float32, device=y.device)
        a = self.a(y, s)
        out = torch.empty_like(y)
        zero = torch.zeros_like(y)
        ops.sde_stage(out, None, 1.0, y.contiguous(), a, self.base_sde.struct(), L.PROC_REVERSE, strato, tt, 1.0,
                      lmbd, dW=zero, t_dev=t_dev)
        return out

    def mu(self, t, y, lmbd=0.):
        """Ito drift at reverse time t (batch-uniform t, as the integrators use it)."""
        return self._stage(t, y, lmbd, False)

    def mu_Strato(self, t, y, lmbd=0.):
        return self._stage(t, y, lmbd, True)

    def sigma(self, t, y, lmbd=0., sparse=False):
        return (1. - lmbd) ** 0.5 * self.base_sde.g(self.T - t, y, sparse)

    def ga(self, s, y):
        """g(s, y) . a(y, s) with a per-row forward time s (SDEs.py:563-580).  Accessor-level (the integrators use
        the fused stage kernel through ``mu`` / ``mu_Strato`` instead, which takes a batch-uniform time)."""
        base = self.base_sde
        a = self.a(y, s.reshape(-1))
        if base.kind == L.SDE_SGM:
            return base.g(s, y) * a
        if base.sparseTensor:
            I, _, K = base.IJK()
            return torch.zeros_like(y).scatter_add_(1, I.unsqueeze(0).expand(y.shape[0], -1), base.g(s, y, True) * a[:, K])
        return torch.einsum('bij, bj -> bi', base.g(s, y), a)

    def ga_m_drift(self, s, y, lmbd=0.):
        """Drift of the reverse generative SDE at forward time s (SDEs.py:560-561); mu(t) = ga_m_drift(T - t)."""
        base = self.base_sde
        return (1. - 0.5 * lmbd) * self.ga(s, y) - base.f(s, y) + (1. - lmbd) * base.div_Sigma(s, y)

    def sample_t_linspace(self, x):
        """Gridded t in (0, T] with the entries <= t_epsilon dropped (SDEs.py:695-706); returns (t, mask)."""
        nsf = self.base_sde.num_steps_forward
        T = self.base_sde.T_float()
        t_ = torch.linspace(T / nsf, T, nsf).to(x.device)
        mask_le_t_eps = (t_ <= self.base_sde.t_epsilon)
        return t_[~mask_le_t_eps], mask_le_t_eps

    # ---- SSM loss -------------------------------------------------------------
    def sample_t(self, x, u=None):
        """t ~ U(0,T) clamped at t_epsilon (SDEs.py:684-693) — returned by
        ``sample_txy``; fused into the perturbation kernel."""
        return self.sample_txy(x, u=u)[0]

    @torch.no_grad()
    def sample_txy(self, x, u=None, eps=None):
        base = self.base_sde
        if base.kind != L.SDE_SGM:
            rng = base.philox(x.device)
            if u is None:
                u = ops.fill_uniform(torch.empty(x.shape[0], device=x.device), rng, L.RNG_STREAM_T)
            T = base.T_float()
            t = u.reshape(-1, 1) * T
            m = (t <= base.t_epsilon).float()
            t = m * base.t_epsilon + (1. - m) * t
            nm, ns = eps if isinstance(eps, (tuple, list)) else (None, None)     # (noise_main, noise_short)
            return t, x, base.sample(t, x, noise_main=nm, noise_short=ns)
        rng = None if (u is not None and eps is not None) else base.philox(x.device)
        y, t = ops.perturb_vp(x.contiguous(), base.struct(), u=u, eps=eps, rng=rng)
        return t.reshape(-1, 1), x, y

    def ssm(self, x, u=None, eps=None, u_v=None, y=None, t_given=None, v=None):
        """Per-sample SSM loss (B,), gradients of its mean accumulated into the
        score net's ``.grad`` (see class docstring).  ``u``/``eps``/``u_v``
        inject the three draws of SDEs.py:688,141,515 (parity tests); ``v`` injects
        the probe itself (any ``vtype``).  ``vtype`` 'gaussian' / 'uniform' (sphere)
        probes (SDEs.py:517-536) are drawn from the Philox stream and go through the
        general form of the loss (u = (dmu/da)^T v, cst = v^T (d(-f)/dy) v)."""
        from .NN import MLP
        base = self.base_sde
        net = self.a
        if not (isinstance(net, MLP) or hasattr(net, "ssm_grad")):
            raise MsgmError(f"no HIP SSM path for score net {type(net).__name__}")
        if self.vtype not in ('rademacher', 'normal', 'gaussian', 'uniform'):
            raise MsgmError(f'vtype {self.vtype} not supported')
        x = x.contiguous().float()
        B, d = x.shape
        dev = x.device
        rng = base.philox(dev)
        if y is None:
            t, _, y = self.sample_txy(x, u=u, eps=eps)
        else:                                        # (t, y) given: the ssm_loss(t_, x, y) entry of the reference
            t = t_given.reshape(-1, 1).contiguous().float()
            y = y.contiguous().float()
        pm1 = v is None and self.vtype == 'rademacher'          # probe entries are +-1: |v|^2 = d is baked into the fused kernel
        drew_v = v is None and not (pm1 and u_v is not None)
        if v is not None:
            v = v.contiguous().float()
        elif pm1:
            v = ops.rademacher((B, d), dev, u=u_v, rng=None if u_v is not None else rng)
        else:
            v = sample_v((B, d), dev, self.vtype, rng=rng).contiguous()
        if drew_v or (t_given is None and (u is None or eps is None)):
            rng.advance(1)
        st = base.struct()
        uu = cst = None
        if base.kind != L.SDE_SGM or not isinstance(net, MLP) or not pm1:
            # general form of the loss: loss_b = adot.u + cst + |a|^2/2 (u = G(y)^T v for the multiplicative SDE)
            uu, cst = ops.ssm_terms(y, v, t.reshape(-1).contiguous(), st)
        flat, gflat = net.flat_parameters()
        if isinstance(net, MLP):
            if self._ws is None or self._ws.device != dev:
                self._ws = ops.mlp_ssm_workspace(d, net.pre is not None, dev)
            per = torch.empty(B, dtype=torch.float32, device=dev)
            gtmp = torch.empty_like(gflat)
            ops.mlp_ssm_grad(net.kernel_params(), y, t.reshape(-1), v, st, 1.0 / B, gtmp, self._ws, loss_per=per, u=uu, cst=cst)
        else:
            # U-Nets: dual-number forward + hand-written backward; the net writes its flat .grad bucket
            # (what was accumulated so far is put back afterwards; a parameter whose .grad is None — zero_grad() —
            # has accumulated nothing, whatever its slice of the bucket still holds)
            keep = gflat.clone()
            off = 0
            for p_ in net.parameters():
                if p_.grad is None:
                    keep[off:off + p_.numel()].zero_()
                off += p_.numel()
            per = net.ssm_grad(y, t.reshape(-1), v, uu, cst, 1.0 / B)
            flat, gflat = net.flat_parameters()
            gtmp = gflat.clone()
            gflat.copy_(keep)
        return _SSMGradBridge.apply(per, gtmp, gflat, net, B, next(net.parameters()))

    def ssm_loss(self, t_, x, y, u_v=None):
        """Reference entry point (SDEs.py:616-646) with (t, y) given: same fused forward-mode kernels as ``ssm``."""
        return self.ssm(x, u_v=u_v, y=y, t_given=t_)

    def latent_sample(self, num_samples, n):
        return self.base_sde.latent_sample(num_samples, n)

    def cond_latent_sample(self, t_, T, x):
        return self.base_sde.cond_latent_sample(t_, T, x)

    def elbo_random_t_slice(self, x, u=None, eps=None, u_v=None, eps_T=None):
        """ELBO slice estimate lp(y_T) - ssm(x) T (SDEs.py:708-721) for the additive SDE; the loss term runs on the
        same fused kernels as training.  ``u``/``eps``/``u_v`` inject the draws of ``ssm`` and ``eps_T`` the noise
        of ``cond_latent_sample`` (parity tests).  The multiplicative SDE's latent density is a sklearn KDE
        (SDEs.py:503-509) and stays out of scope."""
        if self.base_sde.kind != L.SDE_SGM:
            raise MsgmError("ELBO evaluation of the multiplicative SDE needs its KDE latent density (out of scope)")
        x = x.contiguous().float()
        qt = 1.0 / self.base_sde.T_float()
        with torch.no_grad():
            loss_ssm = self.ssm(x, u=u, eps=eps, u_v=u_v).detach() / qt
            t_ = torch.empty(x.shape[0], 1, device=x.device)
            yT = self.base_sde.cond_latent_sample(t_, self.base_sde.T_float(), x, eps=eps_T)
            lp = self.base_sde.log_latent_pdf(yT).view(x.size(0), -1).sum(1)
        return lp - loss_ssm


class _SSMGradBridge(torch.autograd.Function):
    """Lets ``ssm(x).mean().backward()`` deliver the kernel-computed parameter
    gradients: backward adds ``sum(grad_output) * grad_of_mean`` to the flat
    gradient bucket (exact for uniform reductions: sum_b c dL_b = c B grad_of_mean)."""

    @staticmethod
    def forward(ctx, per, gmean, gflat, net, B, anchor):
        # `anchor` (a parameter that requires grad) only makes autograd record this node
        ctx.gmean, ctx.gflat, ctx.net, ctx.B = gmean, gflat, net, B
        return per.clone()

    @staticmethod
    def backward(ctx, grad_out):
        scale = grad_out.sum()          # = c*B for a uniform weight c; gmean already carries 1/B
        net = ctx.net
        flat, gflat = net.flat_parameters()
        off = 0
        for p in net.parameters():
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != gflat[off:off + k].data_ptr():
                gflat[off:off + k].zero_()
                p.grad = gflat[off:off + k].view(p.shape)
            off += k
        gflat.add_(ctx.gmean * scale)
        return None, None, None, None, None, None
