"""SDE integrators on HIP kernels — host mirror of the reference's
``sde_scheme.py`` (same function names, argument lists, return layout and
error behaviour: sde_scheme.py:44-46,73-75,94-99).

Per step the reference launches ~20 eager kernels plus a host ``.item()`` and,
with ``keep_all_samples``, a blocking D2H copy (sde_scheme.py:80-92).  Here a
step is ONE fused stage kernel after the score net (or one kernel in total for
MLP + SGM: ``msgm_mlp_em_step``), the time grid is computed on the host exactly
as upstream (fp32 ``linspace``) and passed by value, trajectories are captured
into a device buffer and copied once at the end, and the whole fused loop can
be replayed as a single hipGraph (``GraphedEMSampler``).

Extra keyword ``noise=`` (steps,B,n) injects the standard-normal draws
(parity tests); otherwise each step uses the Philox stream of the base SDE.
"""
from __future__ import annotations

import os

from typing import Optional

import torch

from . import _lib as L
from . import ops
from ._lib import MsgmError


def EMstep(mu, delta, sigma, dW, sparse=False, I=None, K=None):
    """Kept for API compatibility (sde_scheme.py:18-40); the integrators below
    never materialise mu / sigma — they call the fused stage kernel."""
    if sparse:
        prod = sigma * dW[:, K]
        dx = torch.zeros_like(dW)
        dx.scatter_add_(1, I.unsqueeze(0).expand(dW.size(0), -1), prod)
    elif sigma.dim() > 2:
        dx = torch.einsum('bij, bj -> bi', sigma, dW)
    else:
        dx = sigma * dW
    return mu * delta + dx


class _Run:
    """Shared set-up of the three samplers (sde_scheme.py:50-78)."""

    def __init__(self, sde, x_0, num_steps, lmbd, keep_all_samples, samplesToKeep, include_t0, T_, norm_correction,
                 noise):
        from .SDEs import PluginReverseSDE, forward_SDE
        self.sde = sde
        self.base = sde.base_sde
        self.device = sde.T.device
        if self.device.type != "cuda":
            raise MsgmError("the HIP integrators need the SDE on a cuda device; there is no CPU fallback")
        if x_0.dim() != 2:
            raise MsgmError("state must be 2-D (B,n)")
        self.B, self.n = x_0.shape
        self.T_ = L.host_scalar(sde.T) if (not torch.is_tensor(T_) and T_ == -1) else L.host_scalar(T_)   # sde_scheme.py:54-57
        self.N = num_steps
        self.delta = self.T_ / num_steps
        self.ts = torch.linspace(0, 1, num_steps + 1) * self.T_               # fp32, host (sde_scheme.py:59)
        self.lmbd = float(lmbd)
        self.reverse = isinstance(sde, PluginReverseSDE)
        if not self.reverse and not isinstance(sde, forward_SDE):
            raise MsgmError("sde must be a PluginReverseSDE or a forward_SDE")
        self.proc = L.PROC_REVERSE if self.reverse else L.PROC_FORWARD
        self.struct = self.base.struct()
        if x_0.is_cuda and x_0.dtype == torch.float32 and x_0.is_contiguous() and x_0.device == self.device:
            # device-to-device copy as a KERNEL (captured steps hold kernel nodes only: see msgm_zero_async in csrc/common.h)
            self.x = ops.lincomb(torch.empty_like(x_0), x_0.detach(), 1.0)
        else:
            self.x = x_0.detach().clone().to(self.device).float().contiguous()
        self.norm0 = ops.row_norm(self.x) if norm_correction else None
        self.include_t0 = bool(include_t0)
        self.keep_all = keep_all_samples
        self.keep = None
        self.traj = None
        if keep_all_samples:
            self.traj = torch.zeros((num_steps + int(self.include_t0), self.B, self.n), device=self.device)
            if self.include_t0:
                self.traj[0].copy_(self.x)
        elif samplesToKeep is not None:
            if not (len(samplesToKeep) == self.B):
                raise ValueError('Error: len(samplesToKeep) must correspond to batch size.')
            self.keep = torch.as_tensor(samplesToKeep).reshape(-1).to(torch.int32).to(self.device).contiguous()
            self.kept = torch.zeros((self.B, self.n), device=self.device)
        self.noise = noise
        if noise is not None and tuple(noise.shape) != (num_steps, self.B, self.n):
            raise MsgmError(f"noise must be (num_steps,B,n) = {(num_steps, self.B, self.n)}")
        self.rng = None if noise is not None else self.base.philox(self.device)
        self.T = self.base.T_float()

    def t(self, i):
        return self.ts[i].item()

    def z(self, i):
        return None if self.noise is None else self.noise[i].to(self.device).contiguous()

    def score(self, x, t):
        """a(x, T - t) for the reverse process (SDEs.py:556-557,569); None for the forward one."""
        if not self.reverse:
            return None
        s = torch.full((self.B,), self.T - t, dtype=torch.float32, device=self.device)
        return self.sde.a(x, s).contiguous()

    def after_step(self, i):
        if self.keep_all:
            self.traj[i + int(self.include_t0)].copy_(self.x)
        elif self.keep is not None:
            ops.keep_rows(self.kept, self.x, self.keep, i + int(self.include_t0))

    def result(self, advance=True):
        if self.rng is not None and advance:
            self.rng.advance(self.N)
        if self.keep_all:
            return self.traj.to('cpu')                                        # (N[+1],B,n)  sde_scheme.py:94-95
        if self.keep is not None:
            return self.kept.to('cpu')
        return self.x.to('cpu')


@torch.no_grad()
def euler_maruyama_sampler(sde, x_0, num_steps=1000, lmbd=0., keep_all_samples=True, samplesToKeep=None,
                           include_t0=False, T_=-1, norm_correction=False, noise=None):
    """Euler–Maruyama (sde_scheme.py:43-99)."""
    return _em(sde, x_0, num_steps, lmbd, keep_all_samples, samplesToKeep, include_t0, T_, norm_correction,
               noise).result()


def _em(sde, x_0, num_steps, lmbd, keep_all_samples, samplesToKeep, include_t0, T_, norm_correction, noise):
    from .NN import MLP
    r = _Run(sde, x_0, num_steps, lmbd, keep_all_samples, samplesToKeep, include_t0, T_, norm_correction, noise)
    fused = r.reverse and isinstance(sde.a, MLP) and r.base.kind == L.SDE_SGM and not norm_correction
    P = sde.a.kernel_params() if fused else None
    if fused and not r.keep_all and r.keep is None and noise is None and r.x.shape[0] > 32:
        # final state only: the whole loop in ONE launch, same time grid and Philox draws as the per-step kernels
        ops.mlp_em_loop(P, r.x, r.struct, r.ts.to(r.device), r.delta, r.lmbd, r.rng, 0)
        return r
    other = torch.empty_like(r.x)
    for i in range(num_steps):
        t = r.t(i)
        if fused:                     # score net + update in ONE kernel
            ops.mlp_em_step(P, r.x, r.struct, t, r.delta, r.lmbd, z=r.z(i), rng=r.rng, rng_step=i)
        else:
            a = r.score(r.x, t)
            inplace = r.base.kind == L.SDE_SGM
            out = r.x if inplace else other
            ops.sde_stage(out, r.x, 1.0, r.x, a, r.struct, r.proc, False, t, r.delta, r.lmbd, z=r.z(i), rng=r.rng,
                          rng_step=i, norm0=r.norm0)
            if not inplace:
                r.x, other = out, r.x
        r.after_step(i)
    return r


@torch.no_grad()
def heun_sampler(sde, x_0, num_steps=1000, lmbd=0., keep_all_samples=True, samplesToKeep=None,
                 include_t0=False, T_=-1, norm_correction=False, noise=None):
    """Heun / RK2 in Stratonovich form (sde_scheme.py:101-172):
    x <- x + 1/2 (K1 + K2), K = mu_Strato*delta + sigma.dW, one shared dW."""
    r = _Run(sde, x_0, num_steps, lmbd, keep_all_samples, samplesToKeep, include_t0, T_, norm_correction, noise)
    dW, k1, xp, xn = (torch.empty_like(r.x) for _ in range(4))
    for i in range(num_steps):
        t = r.t(i)
        # predictor x + K1 (also keeps K1 and the Wiener increment)
        ops.sde_stage(xp, r.x, 1.0, r.x, r.score(r.x, t), r.struct, r.proc, True, t, r.delta, r.lmbd, z=r.z(i),
                      rng=r.rng, rng_step=i, dW_out=dW, inc_out=k1)
        ops.lincomb(xn, r.x, 1.0, k1, 0.5)                                    # x + K1/2
        t2 = float(torch.tensor(t, dtype=torch.float32) + r.delta)            # fp32 t + delta as upstream
        # corrector: (x + K1/2) + K2/2, K2 evaluated at the predictor (out aliases base, not x_eval)
        ops.sde_stage(xn, xn, 0.5, xp, r.score(xp, t2), r.struct, r.proc, True, t2, r.delta, r.lmbd, dW=dW,
                      norm0=r.norm0)
        r.x, xn = xn, r.x
        r.after_step(i)
    return r.result()


@torch.no_grad()
def rk4_stratonovich_sampler(sde, x_0, num_steps=1000, lmbd=0., keep_all_samples=True, samplesToKeep=None,
                             include_t0=False, T_=-1, norm_correction=False, noise=None):
    """RK4 for Stratonovich SDEs, one dW per step (sde_scheme.py:174-269)."""
    return _rk4(sde, x_0, num_steps, lmbd, keep_all_samples, samplesToKeep, include_t0, T_, norm_correction,
                noise).result()


def _rk4(sde, x_0, num_steps, lmbd, keep_all_samples, samplesToKeep, include_t0, T_, norm_correction, noise):
    r = _Run(sde, x_0, num_steps, lmbd, keep_all_samples, samplesToKeep, include_t0, T_, norm_correction, noise)
    dW, k1, k2, k3, k4, xm, xn = (torch.empty_like(r.x) for _ in range(7))
    for i in range(num_steps):
        t = r.t(i)
        tf = torch.tensor(t, dtype=torch.float32)
        th, te = float(tf + r.delta / 2), float(tf + r.delta)
        ops.sde_stage(xm, r.x, 0.5, r.x, r.score(r.x, t), r.struct, r.proc, True, t, r.delta, r.lmbd, z=r.z(i),
                      rng=r.rng, rng_step=i, dW_out=dW, inc_out=k1)           # x + K1/2
        ops.sde_stage(xn, r.x, 0.5, xm, r.score(xm, th), r.struct, r.proc, True, th, r.delta, r.lmbd, dW=dW, inc_out=k2)
        ops.sde_stage(xm, r.x, 1.0, xn, r.score(xn, th), r.struct, r.proc, True, th, r.delta, r.lmbd, dW=dW, inc_out=k3)
        ops.sde_stage(k4, None, 1.0, xm, r.score(xm, te), r.struct, r.proc, True, te, r.delta, r.lmbd, dW=dW)
        ops.rk4_combine(xn, r.x, k1, k2, k3, k4, norm0=r.norm0)
        r.x, xn = xn, r.x
        r.after_step(i)
    return r


@torch.no_grad()
def msgm_forward_perturb(base, t, y0, noise_main=None, noise_short=None):
    """y_t | y_0 for the multiplicative SDE (SDE.sample_scheme, SDEs.py:78-122)
    kept on the device: all rows are RK4-integrated over the nsf-step grid and
    row b is captured when the step count reaches k_b = trunc(nsf t_b / T)
    (bit-exact int32 index kernel); rows with k_b = 0 take one RK4 step of
    length t_b from y_0 (per-row delta inside the stage kernel) — no Python
    loop over rows, no D2H/H2D round trips."""
    from .SDEs import forward_SDE
    dev = y0.device
    B, n = y0.shape
    nsf = base.num_steps_forward
    T = base.T_float()
    tt = t.reshape(-1).contiguous().float()
    k = ops.forward_step_index(tt, nsf, T)
    fwd = forward_SDE(base, base.T)
    kept = _rk4(fwd, y0, nsf, 0., False, k, True, -1, False, noise_main).kept
    # rows with k == 0: one RK4 step, delta_b = t_b, stage times 0, t_b/2, t_b/2, t_b
    st = base.struct()
    x = y0.contiguous().float()
    dW, k1, k2, k3, k4, xm, xn = (torch.empty_like(x) for _ in range(7))
    rng = None if noise_short is not None else base.philox(dev)
    z = None if noise_short is None else noise_short.to(dev).contiguous()
    kw = dict(delta_rows=tt)
    ops.sde_stage(xm, x, 0.5, x, None, st, L.PROC_FORWARD, True, 0.0, 0.0, 0.0, z=z, rng=rng, rng_step=nsf + 1,
                  dW_out=dW, inc_out=k1, t_frac=0.0, **kw)
    ops.sde_stage(xn, x, 0.5, xm, None, st, L.PROC_FORWARD, True, 0.0, 0.0, 0.0, dW=dW, inc_out=k2, t_frac=0.5, **kw)
    ops.sde_stage(xm, x, 1.0, xn, None, st, L.PROC_FORWARD, True, 0.0, 0.0, 0.0, dW=dW, inc_out=k3, t_frac=0.5, **kw)
    ops.sde_stage(k4, None, 1.0, xm, None, st, L.PROC_FORWARD, True, 0.0, 0.0, 0.0, dW=dW, t_frac=1.0, **kw)
    ops.rk4_combine(xn, x, k1, k2, k3, k4)
    ops.keep_rows(kept, xn, k, 0)
    base.philox(dev).advance(2 * nsf + 2)
    return kept


class GraphedEMSampler:
    """The whole fused EM loop (MLP + SGM) captured as ONE hipGraph: N kernel
    nodes with the time grid and Philox step index baked in by value; the state
    buffer and the Philox state live at fixed device addresses, so a replay is a
    single host call.  (hipGraph instead of a tracing compiler.)"""

    def __init__(self, sde, B, num_steps, lmbd=0.0, T_=None):
        from .NN import MLP
        if not (isinstance(sde.a, MLP) and sde.base_sde.kind == L.SDE_SGM):
            raise MsgmError("GraphedEMSampler is built for MLP + SGMsde")
        self.sde, self.N = sde, num_steps
        dev = sde.T.device
        base = sde.base_sde
        self.T_ = base.T_float() if T_ is None else float(T_)
        self.x = torch.zeros(B, sde.a.input_dim, device=dev)
        self.rng = base.philox(dev)
        P = sde.a.kernel_params()
        st = base.struct()
        ts = torch.linspace(0, 1, num_steps + 1) * self.T_
        delta = self.T_ / num_steps
        self.ts_dev = ts.to(dev)
        self._keep = (P, st)
        # B > 32: the whole loop is ONE launch (msgm_mlp_em_loop — rows never interact, so a workgroup carries its rows
        # through all steps); otherwise N single-step kernel nodes.  Same Philox numbers either way.
        self.one_launch = B > 32 and not os.environ.get("MSGM_EM_PER_STEP")

        def body():
            if self.one_launch:
                ops.mlp_em_loop(P, self.x, st, self.ts_dev, delta, lmbd, self.rng, 0)
            else:
                for i in range(num_steps):
                    ops.mlp_em_step(P, self.x, st, ts[i].item(), delta, lmbd, rng=self.rng, rng_step=i)
            self.rng.advance(num_steps)

        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            body()                                   # warm-up (loads code objects) before capture
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        self.graph = ops.new_graph()
        with torch.cuda.graph(self.graph):
            body()

    @torch.no_grad()
    def run(self, x_0):
        self.x.copy_(x_0)
        self.graph.replay()
        return self.x


class GraphedStepSampler:
    """Euler-Maruyama, Heun or RK4-Stratonovich with ANY HIP score net (U-Nets): ONE step — device clock ticks, the
    score-net forward(s) (hundreds of kernels each), the fused stage kernels, step-counter bump — is captured as a
    hipGraph and replayed num_steps times.  The time grid (fp32 ``linspace`` as upstream, sde_scheme.py:59), the stage
    times t + delta/2, t + delta (fp32 adds as upstream, sde_scheme.py:150,233-247) and the step index live on the
    device, so the replays need no host scalar (the reference does ``ts[i].item()`` + ``fill_`` every step,
    sde_scheme.py:81).  ``method='rk4'`` is what the driver generates with (MSGM_higherDim.py:903).  Only kernel nodes
    are captured (no D2D memcpy / memset nodes — see msgm_zero_async in csrc/common.h)."""

    def __init__(self, sde, B, n, num_steps, lmbd=0.0, norm_correction=False, method="em"):
        from .SDEs import PluginReverseSDE
        if not isinstance(sde, PluginReverseSDE):
            raise MsgmError("GraphedStepSampler integrates a PluginReverseSDE")
        if method not in ("em", "heun", "rk4"):
            raise MsgmError(f"unknown integrator {method}")
        base = sde.base_sde
        dev = sde.T.device
        self.sde, self.N, self.B, self.n, self.method = sde, num_steps, B, n, method
        T = base.T_float()
        self.ts = (torch.linspace(0, 1, num_steps + 1) * T).to(dev)
        self.step = torch.zeros(1, dtype=torch.int64, device=dev)
        nstage = {"em": 1, "heun": 2, "rk4": 3}[method]                   # distinct stage times
        self.t_dev = [torch.zeros(1, device=dev) for _ in range(nstage)]
        self.s = [torch.zeros(B, device=dev) for _ in range(nstage)]
        self.x = torch.zeros(B, n, device=dev)
        self.norm0 = torch.zeros(B, device=dev) if norm_correction else None
        self.rng = base.philox(dev)
        st, delta = base.struct(), T / num_steps
        self._keep = st
        inplace = base.kind == L.SDE_SGM
        buf = lambda: torch.zeros(B, n, device=dev)
        if method == "em":
            other = None if inplace else buf()
        elif method == "heun":
            dW, k1, xp, xn = buf(), buf(), buf(), buf()
        else:
            dW, k1, k2, k3, k4, xm, xn = (buf() for _ in range(7))
        f32 = lambda v: float(torch.tensor(v, dtype=torch.float32))       # the fp32 value upstream adds to the fp32 t
        R = L.PROC_REVERSE

        def score(xx, k):
            return sde.a(xx, self.s[k]).contiguous()

        def body():
            x = self.x
            ops.time_tick(self.ts, self.step, T, self.t_dev[0], self.s[0])
            if method == "em":
                out = x if inplace else other
                ops.sde_stage(out, x, 1.0, x, score(x, 0), st, R, False, 0.0, delta, lmbd, rng=self.rng, norm0=self.norm0,
                              t_dev=self.t_dev[0], step_dev=self.step)
                if not inplace:
                    ops.lincomb(x, out, 1.0)
            elif method == "heun":                                        # sde_scheme.py:101-172
                ops.time_tick(self.ts, self.step, T, self.t_dev[1], self.s[1], t_add=f32(delta))
                ops.sde_stage(xp, x, 1.0, x, score(x, 0), st, R, True, 0.0, delta, lmbd, rng=self.rng, dW_out=dW, inc_out=k1,
                              t_dev=self.t_dev[0], step_dev=self.step)
                ops.lincomb(xn, x, 1.0, k1, 0.5)
                ops.sde_stage(xn, xn, 0.5, xp, score(xp, 1), st, R, True, 0.0, delta, lmbd, dW=dW, norm0=self.norm0,
                              t_dev=self.t_dev[1])
                ops.lincomb(x, xn, 1.0)
            else:                                                         # sde_scheme.py:174-269
                ops.time_tick(self.ts, self.step, T, self.t_dev[1], self.s[1], t_add=f32(delta / 2))
                ops.time_tick(self.ts, self.step, T, self.t_dev[2], self.s[2], t_add=f32(delta))
                ops.sde_stage(xm, x, 0.5, x, score(x, 0), st, R, True, 0.0, delta, lmbd, rng=self.rng, dW_out=dW, inc_out=k1,
                              t_dev=self.t_dev[0], step_dev=self.step)
                ops.sde_stage(xn, x, 0.5, xm, score(xm, 1), st, R, True, 0.0, delta, lmbd, dW=dW, inc_out=k2, t_dev=self.t_dev[1])
                ops.sde_stage(xm, x, 1.0, xn, score(xn, 1), st, R, True, 0.0, delta, lmbd, dW=dW, inc_out=k3, t_dev=self.t_dev[1])
                ops.sde_stage(k4, None, 1.0, xm, score(xm, 2), st, R, True, 0.0, delta, lmbd, dW=dW, t_dev=self.t_dev[2])
                ops.rk4_combine(xn, x, k1, k2, k3, k4, norm0=self.norm0)
                ops.lincomb(x, xn, 1.0)
            ops.counter_inc(self.step)

        self._body = body
        self.graph = None
        if os.environ.get("MSGM_NO_GRAPH_SAMPLER"):          # diagnostic: the same step enqueued eagerly every time
            return
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            body()                                   # warm-up outside capture
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = ops.new_graph()
        with torch.cuda.graph(self.graph):
            body()

    @torch.no_grad()
    def run(self, x_0):
        ops.lincomb(self.x, x_0.contiguous().float().view(self.B, self.n), 1.0)
        if self.norm0 is not None:
            ops.lincomb(self.norm0, ops.row_norm(self.x), 1.0)
        self.step.zero_()
        for _ in range(self.N):
            if self.graph is None:
                self._body()
            else:
                self.graph.replay()
        self.rng.advance(self.N)
        return self.x
