"""Thin tensor-level wrappers over the C ABI (one function per entry point).

Every function validates shapes on the host (a faulting kernel can reset the
GPU for everyone), passes raw device pointers through ctypes, and raises
``MsgmError`` on a non-zero status.  No arithmetic happens here.
"""
from __future__ import annotations

import ctypes as C_
import math
from typing import Optional

import os

import torch

from . import _lib as L
from ._lib import MsgmError, PhiloxState, check, f32, lib, ptr, stream


def _rng_ptr(rng: Optional[PhiloxState]):
    return None if rng is None else rng.ptr()


def _same_shape(a: Optional[torch.Tensor], shape, name: str):
    if a is not None and tuple(a.shape) != tuple(shape):
        raise MsgmError(f"{name} has shape {tuple(a.shape)}, expected {tuple(shape)}")


def fill_uniform(out: torch.Tensor, rng: PhiloxState, stream_id: int = L.RNG_STREAM_USER) -> torch.Tensor:
    check(lib().msgm_fill_uniform(ptr(f32(out)), out.numel(), rng.ptr(), stream_id, stream()), "msgm_fill_uniform")
    return out


def fill_normal(out: torch.Tensor, rng: PhiloxState, stream_id: int = L.RNG_STREAM_USER) -> torch.Tensor:
    check(lib().msgm_fill_normal(ptr(f32(out)), out.numel(), rng.ptr(), stream_id, stream()), "msgm_fill_normal")
    return out


def perturb_vp(x0: torch.Tensor, sde: L.SdeT, u: Optional[torch.Tensor] = None, eps: Optional[torch.Tensor] = None,
               rng: Optional[PhiloxState] = None, return_eps: bool = False):
    """(y, t[, eps]) — K1.  x0 (B,d); u (B,) raw uniforms; eps (B,d)."""
    if x0.dim() != 2:
        raise MsgmError("x0 must be (B,d)")
    B, d = x0.shape
    if u is not None:
        u = u.reshape(-1)
        _same_shape(u, (B,), "u")
    _same_shape(eps, (B, d), "eps")
    y = torch.empty_like(x0)
    t = torch.empty(B, dtype=torch.float32, device=x0.device)
    eo = torch.empty_like(x0) if return_eps else None
    check(lib().msgm_perturb_vp(ptr(f32(x0)), ptr(y), ptr(t), ptr(eo), B, d, sde, ptr(u), ptr(eps), _rng_ptr(rng),
                                stream()), "msgm_perturb_vp")
    return (y, t, eo) if return_eps else (y, t)


def perturb_vp_at(x0: torch.Tensor, sde: L.SdeT, t: torch.Tensor, eps: Optional[torch.Tensor] = None,
                  rng: Optional[PhiloxState] = None, return_eps: bool = False):
    """y_t | y_0 at GIVEN times t (B,) — used as passed, no clamp (SGMsde.sample, SDEs.py:134-146)."""
    if x0.dim() != 2:
        raise MsgmError("x0 must be (B,d)")
    B, d = x0.shape
    t = t.reshape(-1).contiguous().float()
    _same_shape(t, (B,), "t")
    _same_shape(eps, (B, d), "eps")
    y = torch.empty_like(x0)
    eo = torch.empty_like(x0) if return_eps else None
    check(lib().msgm_perturb_vp_at(ptr(f32(x0)), ptr(y), ptr(eo), B, d, sde, ptr(t), ptr(eps), _rng_ptr(rng), stream()),
          "msgm_perturb_vp_at")
    return (y, eo) if return_eps else y


def forward_step_index(t: torch.Tensor, nsf: int, T: float) -> torch.Tensor:
    t = t.reshape(-1)
    k = torch.empty(t.numel(), dtype=torch.int32, device=t.device)
    check(lib().msgm_forward_step_index(ptr(f32(t)), ptr(k), t.numel(), int(nsf), float(T), stream()),
          "msgm_forward_step_index")
    return k


def rademacher(shape, device, u: Optional[torch.Tensor] = None, rng: Optional[PhiloxState] = None) -> torch.Tensor:
    v = torch.empty(shape, dtype=torch.float32, device=device)
    _same_shape(u, v.shape, "u")
    check(lib().msgm_rademacher(ptr(v), v.numel(), ptr(u), _rng_ptr(rng), stream()), "msgm_rademacher")
    return v


def sde_stage(out: torch.Tensor, base: Optional[torch.Tensor], c_out: float, x: torch.Tensor,
              a: Optional[torch.Tensor], sde: L.SdeT, proc: int, strato: bool, t: float, delta: float, lmbd: float = 0.0,
              dW: Optional[torch.Tensor] = None, z: Optional[torch.Tensor] = None, rng: Optional[PhiloxState] = None,
              rng_step: int = 0, dW_out: Optional[torch.Tensor] = None, norm0: Optional[torch.Tensor] = None,
              inc_out: Optional[torch.Tensor] = None, delta_rows: Optional[torch.Tensor] = None,
              t_frac: float = 0.0, t_dev: Optional[torch.Tensor] = None,
              step_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One integrator stage (K2/K3/K4): out = base + c_out*(drift*delta + sigma.dW)."""
    if x.dim() != 2:
        raise MsgmError("state must be 2-D (B,n)")
    B, n = x.shape
    for nm, tt in (("out", out), ("base", base), ("a", a), ("dW", dW), ("z", z), ("dW_out", dW_out), ("inc_out", inc_out)):
        _same_shape(tt, (B, n), nm)
    _same_shape(norm0, (B,), "norm0")
    _same_shape(delta_rows, (B,), "delta_rows")
    check(lib().msgm_sde_stage(ptr(f32(out)), ptr(base), float(c_out), ptr(f32(x)), ptr(a), ptr(dW), ptr(z),
                               float(delta ** 0.5), _rng_ptr(rng), int(rng_step), ptr(dW_out), ptr(inc_out), B, n, sde,
                               int(proc), int(bool(strato)), float(t), float(delta), float(lmbd), ptr(norm0),
                               ptr(delta_rows), float(t_frac), ptr(t_dev), ptr(step_dev), stream()),
          "msgm_sde_stage")
    return out


def ssm_loss_diag(out: torch.Tensor, v: torch.Tensor, t: torch.Tensor, sde: L.SdeT, inv_batch: float):
    """(per[B], g[2B][n]) from the stacked (primal | tangent) net output — K12."""
    B, n = v.shape
    if out.numel() != 2 * B * n or t.numel() != B:
        raise MsgmError("ssm_loss_diag: out must be [2B][n], t [B]")
    per = torch.empty(B, dtype=torch.float32, device=v.device)
    g = torch.empty(2 * B * n, dtype=torch.float32, device=v.device)
    check(lib().msgm_ssm_loss_diag(ptr(f32(out)), ptr(f32(v)), ptr(f32(t)), ptr(per), ptr(g), B, n, sde, float(inv_batch),
                                   stream()), "msgm_ssm_loss_diag")
    return per, g


def lincomb(out: torch.Tensor, a: torch.Tensor, c0: float, b: Optional[torch.Tensor] = None, c1: float = 0.0,
            c: Optional[torch.Tensor] = None, c2: float = 0.0) -> torch.Tensor:
    n = a.numel()
    for nm, tt in (("out", out), ("b", b), ("c", c)):
        if tt is not None and tt.numel() != n:
            raise MsgmError(f"{nm} has {tt.numel()} elements, expected {n}")
    check(lib().msgm_lincomb(ptr(f32(out)), ptr(f32(a)), float(c0), ptr(b), float(c1), ptr(c), float(c2), n, stream()),
          "msgm_lincomb")
    return out


def rk4_combine(out, x, k1, k2, k3, k4, norm0: Optional[torch.Tensor] = None):
    B, n = x.shape
    for nm, tt in (("out", out), ("k1", k1), ("k2", k2), ("k3", k3), ("k4", k4)):
        _same_shape(tt, (B, n), nm)
    _same_shape(norm0, (B,), "norm0")
    check(lib().msgm_rk4_combine(ptr(out), ptr(x), ptr(k1), ptr(k2), ptr(k3), ptr(k4), B, n, ptr(norm0), stream()),
          "msgm_rk4_combine")
    return out


def row_norm(x: torch.Tensor) -> torch.Tensor:
    B, n = x.shape
    out = torch.empty(B, dtype=torch.float32, device=x.device)
    check(lib().msgm_row_norm(ptr(f32(x)), ptr(out), B, n, stream()), "msgm_row_norm")
    return out


def keep_rows(kept: torch.Tensor, x: torch.Tensor, stop: torch.Tensor, index: int):
    B, n = x.shape
    _same_shape(kept, (B, n), "kept")
    if stop.dtype != torch.int32 or stop.numel() != B:
        raise MsgmError("stop must be int32 of length B")
    check(lib().msgm_keep_rows(ptr(kept), ptr(x), ptr(stop), int(index), B, n, stream()), "msgm_keep_rows")
    return kept


def adam_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float = 1e-3,
              beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, gscale: float = 1.0,
              step_dev: Optional[torch.Tensor] = None):
    n = p.numel()
    for nm, tt in (("g", g), ("m", m), ("v", v)):
        if tt.numel() != n:
            raise MsgmError(f"{nm} has {tt.numel()} elements, expected {n}")
    check(lib().msgm_adam_step(ptr(f32(p)), ptr(f32(g)), ptr(f32(m)), ptr(f32(v)), n, lr, beta1, beta2, eps,
                               float(gscale), int(step), ptr(step_dev), stream()), "msgm_adam_step")


def time_tick(ts: torch.Tensor, step: torch.Tensor, T: float, t_dev: torch.Tensor, s_out: torch.Tensor, t_add: float = 0.0):
    """Device clock of a replayed sampler step: t_dev = ts[step] (+ t_add in fp32 for the later Heun / RK4 stages),
    s_out[:] = T - t_dev."""
    if t_add == 0.0:
        check(lib().msgm_time_tick(ptr(f32(ts)), ptr(step), ts.numel(), float(T), ptr(f32(t_dev)), ptr(f32(s_out)),
                                   s_out.numel(), stream()), "msgm_time_tick")
    else:
        check(lib().msgm_time_tick_stage(ptr(f32(ts)), ptr(step), ts.numel(), float(T), float(t_add), ptr(f32(t_dev)),
                                         ptr(f32(s_out)), s_out.numel(), stream()), "msgm_time_tick_stage")


def counter_inc(ctr: torch.Tensor):
    check(lib().msgm_counter_inc(ptr(ctr), stream()), "msgm_counter_inc")


# ---------------------------------------------------------------- fused MLP
def mlp_params(W1, b1, W2, b2, W3, b3, W4, b4, premodule: bool) -> L.MlpParamsT:
    d = W4.shape[0]
    in_dim = d + 1 + (1 if premodule else 0)
    exp = {"W1": (128, in_dim), "b1": (128,), "W2": (128, 128), "b2": (128,), "W3": (128, 128), "b3": (128,),
           "W4": (d, 128), "b4": (d,)}
    for nm, tt in (("W1", W1), ("b1", b1), ("W2", W2), ("b2", b2), ("W3", W3), ("b3", b3), ("W4", W4), ("b4", b4)):
        _same_shape(tt, exp[nm], nm)
        f32(tt, nm)
    return L.MlpParamsT(ptr(W1), ptr(b1), ptr(W2), ptr(b2), ptr(W3), ptr(b3), ptr(W4), ptr(b4), d, int(bool(premodule)))


def mlp_num_params(d: int, premodule: bool) -> int:
    return int(lib().msgm_mlp_num_params(d, int(bool(premodule))))


def mlp_forward(P: L.MlpParamsT, y: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    B, d = y.shape
    if d != P.d:
        raise MsgmError(f"input has d={d}, the MLP was built for d={P.d}")
    t = t.reshape(-1)
    _same_shape(t, (B,), "t")
    a = torch.empty_like(y)
    check(lib().msgm_mlp_forward(P, ptr(f32(y)), ptr(f32(t)), ptr(a), B, stream()), "msgm_mlp_forward")
    return a


def mlp_em_step(P: L.MlpParamsT, x: torch.Tensor, sde: L.SdeT, t: float, delta: float, lmbd: float = 0.0,
                z: Optional[torch.Tensor] = None, rng: Optional[PhiloxState] = None, rng_step: int = 0):
    B, d = x.shape
    if d != P.d:
        raise MsgmError(f"state has d={d}, the MLP was built for d={P.d}")
    _same_shape(z, (B, d), "z")
    check(lib().msgm_mlp_em_step(P, ptr(f32(x)), B, sde, float(t), float(delta), float(lmbd), ptr(z), _rng_ptr(rng),
                                 int(rng_step), stream()), "msgm_mlp_em_step")
    return x


def mlp_em_loop(P: L.MlpParamsT, x: torch.Tensor, sde: L.SdeT, ts: torch.Tensor, delta: float, lmbd: float,
                rng: PhiloxState, rng_step0: int = 0):
    """All len(ts)-1 Euler-Maruyama steps in ONE launch (final state only); ts = device fp32 time grid."""
    B, d = x.shape
    if d != P.d:
        raise MsgmError(f"state has d={d}, the MLP was built for d={P.d}")
    if ts.device != x.device or ts.dtype != torch.float32 or ts.numel() < 2:
        raise MsgmError("em loop: ts must be a device fp32 grid of N+1 values")
    check(lib().msgm_mlp_em_loop(P, ptr(f32(x)), B, sde, ptr(ts), ts.numel() - 1, float(delta), float(lmbd), _rng_ptr(rng),
                                 int(rng_step0), stream()), "msgm_mlp_em_loop")
    return x


def mlp_ssm_workspace(d: int, premodule: bool, device) -> torch.Tensor:
    nbytes = int(lib().msgm_mlp_ssm_workspace(d, int(bool(premodule))))
    return torch.empty(nbytes // 4, dtype=torch.float32, device=device)


def ssm_terms(y: torch.Tensor, v: torch.Tensor, t: torch.Tensor, sde: L.SdeT):
    """(u (B,n), cst (B)): loss_b = adot.u + cst + |a|^2/2 for any SDE family."""
    B, n = y.shape
    if tuple(v.shape) != (B, n) or t.numel() != B:
        raise MsgmError("ssm_terms: y, v (B,n) and t (B)")
    u, cst = torch.empty_like(y), torch.empty(B, dtype=torch.float32, device=y.device)
    check(lib().msgm_ssm_terms(ptr(f32(y)), ptr(f32(v)), ptr(f32(t)), ptr(u), ptr(cst), B, n, sde, stream()), "msgm_ssm_terms")
    return u, cst


def ssm_loss(out: torch.Tensor, u: torch.Tensor, cst: torch.Tensor, inv_batch: float):
    B, n = u.shape
    if out.numel() != 2 * B * n or cst.numel() != B:
        raise MsgmError("ssm_loss: out must be [2B][n]")
    per = torch.empty(B, dtype=torch.float32, device=u.device)
    g = torch.empty(2 * B * n, dtype=torch.float32, device=u.device)
    check(lib().msgm_ssm_loss(ptr(f32(out)), ptr(f32(u)), ptr(f32(cst)), ptr(per), ptr(g), B, n, float(inv_batch), stream()),
          "msgm_ssm_loss")
    return per, g


def mlp_ssm_grad(P: L.MlpParamsT, y: torch.Tensor, t: torch.Tensor, v: torch.Tensor, sde: L.SdeT, inv_batch: float,
                 grads: torch.Tensor, workspace: torch.Tensor, loss_per: Optional[torch.Tensor] = None,
                 loss_sum: Optional[torch.Tensor] = None, u: Optional[torch.Tensor] = None,
                 cst: Optional[torch.Tensor] = None):
    B, d = y.shape
    if d != P.d:
        raise MsgmError(f"input has d={d}, the MLP was built for d={P.d}")
    t = t.reshape(-1)
    _same_shape(t, (B,), "t")
    _same_shape(v, (B, d), "v")
    _same_shape(loss_per, (B,), "loss_per")
    if grads.numel() != mlp_num_params(d, bool(P.premodule)):
        raise MsgmError("grads bucket has the wrong size")
    _same_shape(u, (B, d), "u")
    _same_shape(cst, (B,), "cst")
    check(lib().msgm_mlp_ssm_grad(P, ptr(f32(y)), ptr(f32(t)), ptr(f32(v)), ptr(u), ptr(cst), B, sde, float(inv_batch), ptr(f32(grads)),
                                  ptr(loss_per), ptr(loss_sum), ptr(workspace), workspace.numel() * 4, stream()),
          "msgm_mlp_ssm_grad")


# ---------------------------------------------------------------- convolutions (implicit GEMM)
def conv_geom(N, Hi, Wi, Ho, Wo, KH, KW, stride, pad, mode=0, ups=0) -> L.ConvGeomT:
    """stride / pad apply to W and — when the kernel has height (KH > 1) — to H; a 1-D conv (H = 1) never pads H."""
    sH, pH = (int(stride), int(pad)) if KH > 1 else (int(stride) if Hi > 1 else 1, 0)
    return L.ConvGeomT(int(N), int(Hi), int(Wi), int(Ho), int(Wo), int(KH), int(KW), sH, pH, int(stride), int(pad), int(mode),
                       int(ups))


def pad16(c: int) -> int:
    return ((int(c) + 15) // 16) * 16


def conv_forward(geom: L.ConvGeomT, src0: torch.Tensor, C0: int, Wp: torch.Tensor, Cout: int, out: torch.Tensor,
                 src1: Optional[torch.Tensor] = None, C1: int = 0, bias: Optional[torch.Tensor] = None,
                 samp_bias: Optional[torch.Tensor] = None, n_bias: int = 0, accumulate: bool = False,
                 CoutP: Optional[int] = None, n_samp: Optional[int] = None, residual: Optional[torch.Tensor] = None,
                 in_scale: Optional[torch.Tensor] = None, in_shift: Optional[torch.Tensor] = None,
                 in_act: int = 0, tapmask_in=None, tapmask_out=None, wino: bool = False,
                 chanstats: Optional[torch.Tensor] = None, b6: bool = False) -> torch.Tensor:
    """out[N][Ho][Wo][Cout] (+)= implicit-GEMM convolution of channels-last inputs (K6/K11).
    chanstats [N][S][2][Cout] (S = conv_chanstats_slots(...) > 0): per-channel partial sums of the final output, for the
    GroupNorm that reads it next (groupnorm_affine_cs).
    wino: Wp is the Winograd image [16][CoutP][Ktot] and the F(2x2,3x3) forward kernel runs (sampler path).
    b6 (opt-in experiment): Wp is the bf16-SPLIT image [3][9][CoutP][Ktot] (torch.bfloat16, b6_split_weights) and the 3x3
    kernel does its matrix work as six bf16 MFMA products per fp32 product (fp32 accumulate).
    residual: added in the epilogue.  in_scale / in_shift [N][C0+C1] (+ in_act=1: SiLU): the conv reads
    act(a x + b) — GroupNorm(+SiLU) folded into the input staging (see conv_input_transform_supported)."""
    CoutP = pad16(Cout) if CoutP is None else CoutP
    Ktot = pad16(C0) + (pad16(C1) if src1 is not None else 0)
    taps = 16 if wino else geom.KH * geom.KW
    if src0.numel() != geom.N * geom.Hi * geom.Wi * C0:
        raise MsgmError(f"src0 has {src0.numel()} elements, geometry says {geom.N * geom.Hi * geom.Wi * C0}")
    if src1 is not None and src1.numel() != geom.N * geom.Hi * geom.Wi * C1:
        raise MsgmError("src1 does not match the geometry")
    if out.numel() != geom.N * geom.Ho * geom.Wo * Cout:
        raise MsgmError(f"out has {out.numel()} elements, geometry says {geom.N * geom.Ho * geom.Wo * Cout}")
    if Wp.numel() < (3 if b6 else 1) * taps * CoutP * Ktot or (b6 and (wino or Wp.dtype != torch.bfloat16)):
        raise MsgmError("packed weight too small (or not the image this kernel reads)")
    if bias is not None and bias.numel() != Cout:
        raise MsgmError("bias size")
    n_samp = n_bias if n_samp is None else n_samp
    if samp_bias is not None and samp_bias.numel() != n_samp * Cout:
        raise MsgmError("samp_bias must be [n_samp][Cout]")
    fuse = None
    if residual is not None or in_scale is not None or tapmask_in or tapmask_out or chanstats is not None:
        ctot = C0 + (C1 if src1 is not None else 0)
        if residual is not None and residual.numel() != out.numel():
            raise MsgmError("residual must have the output's size")
        if (in_scale is None) != (in_shift is None) or (in_scale is not None and
                                                        (in_scale.numel() != geom.N * ctot or in_shift.numel() != geom.N * ctot)):
            raise MsgmError("in_scale / in_shift must both be [N][C0+C1]")
        fuse = L.ConvFuseT(ptr(residual), ptr(in_scale), ptr(in_shift), int(in_act), 0)
        if chanstats is not None:
            S = ((geom.Ho // 16) * (geom.Wo // 16) * 4 if (wino or b6) else
                 conv_chanstats_slots(geom, C0, C1 if src1 is not None else 0, Cout, CoutP))
            if S <= 0 or chanstats.numel() < geom.N * S * 2 * Cout or chanstats.dtype != torch.float32:
                raise MsgmError("chanstats: this convolution has no statistics by-product, or the buffer is too small")
            fuse.chanstats = ptr(chanstats)
        for i, m in enumerate((tapmask_in or [])[:16]):      # structurally-zero weight blocks (tile kernel skips them)
            fuse.tapmask_in[i] = int(m)
        for i, m in enumerate((tapmask_out or [])[:8]):
            fuse.tapmask_out[i] = int(m)
    if b6:
        if tapmask_in or tapmask_out:
            raise MsgmError("the bf16-split kernel has no tap masks")
        check(lib().msgm_conv_forward_b6(geom, ptr(f32(src0)), C0, ptr(src1), C1, ptr(Wp), Cout, CoutP, Ktot, ptr(bias),
                                         ptr(samp_bias), int(n_bias), int(n_samp), ptr(f32(out)), int(bool(accumulate)),
                                         fuse, stream()), "msgm_conv_forward_b6")
        return out
    if wino:
        if tapmask_in or tapmask_out:
            raise MsgmError("the Winograd kernel has no tap masks")
        check(lib().msgm_conv_forward_wino(geom, ptr(f32(src0)), C0, ptr(src1), C1, ptr(f32(Wp)), Cout, CoutP, Ktot, ptr(bias),
                                           ptr(samp_bias), int(n_bias), int(n_samp), ptr(f32(out)), int(bool(accumulate)),
                                           fuse, stream()), "msgm_conv_forward_wino")
        return out
    check(lib().msgm_conv_forward_fused(geom, ptr(f32(src0)), C0, ptr(src1), C1, ptr(f32(Wp)), Cout, CoutP, Ktot, ptr(bias),
                                        ptr(samp_bias), int(n_bias), int(n_samp), ptr(f32(out)), int(bool(accumulate)),
                                        fuse, stream()), "msgm_conv_forward")
    return out


def conv_chanstats_slots(geom: L.ConvGeomT, C0: int, C1: int, Cout: int, CoutP: int) -> int:
    """Slots per sample of the per-channel statistics by-product of this forward convolution (0 = it has none)."""
    return int(lib().msgm_conv_chanstats_slots(geom, int(C0), int(C1), int(Cout), int(CoutP)))


def groupnorm_affine_cs(cs0, S0, C0, gamma, beta, Bp, P, G, cs1=None, S1=0, C1=0, eps=1e-5):
    """groupnorm_affine() from the producers' channel statistics (conv_forward(chanstats=)) instead of the tensors."""
    C = C0 + (C1 if cs1 is not None else 0)
    if cs0.numel() < Bp * S0 * 2 * C0 or (cs1 is not None and cs1.numel() < Bp * S1 * 2 * C1) or gamma.numel() != C or beta.numel() != C:
        raise MsgmError("groupnorm_affine_cs: sizes")
    scale = torch.empty(Bp * C, dtype=torch.float32, device=cs0.device)
    shift = torch.empty(Bp * C, dtype=torch.float32, device=cs0.device)
    check(lib().msgm_groupnorm_affine_chanstats(ptr(f32(cs0)), int(S0), int(C0), ptr(cs1), int(S1 if cs1 is not None else 0),
                                                int(C1 if cs1 is not None else 0), ptr(f32(gamma)), ptr(f32(beta)), ptr(scale),
                                                ptr(shift), int(Bp), int(P), int(G), float(eps), stream()),
          "msgm_groupnorm_affine_chanstats")
    return scale, shift


def conv_input_transform_supported(geom: L.ConvGeomT, C0: int, C1: int, CoutP: int, Cout: Optional[int] = None) -> bool:
    if Cout is not None and lib().msgm_conv_small_cout_supported(geom, int(C0), int(C1), int(Cout)):
        return True
    return bool(lib().msgm_conv_input_transform_supported(geom, int(C0), int(C1), int(CoutP)))


def groupnorm_affine(x0, C0, gamma, beta, Bp, P, G, x1=None, C1=0, eps=1e-5):
    """GroupNorm statistics of x0 (| x1 concatenated along channels) as a per-(sample, channel) affine map
    (scale, shift), each [Bp][C0+C1], for conv_forward(in_scale=, in_shift=).  No tangent."""
    C = C0 + (C1 if x1 is not None else 0)
    if x0.numel() != Bp * P * C0 or (x1 is not None and x1.numel() != Bp * P * C1) or gamma.numel() != C or beta.numel() != C:
        raise MsgmError("groupnorm_affine: size mismatch")
    scale = torch.empty(Bp * C, dtype=torch.float32, device=x0.device)
    shift = torch.empty_like(scale)
    ws = _gn_ws(Bp, G, x0.device)
    check(lib().msgm_groupnorm_affine(ptr(f32(x0)), C0, ptr(x1), C1, ptr(f32(gamma)), ptr(f32(beta)), ptr(scale), ptr(shift),
                                      Bp, P, G, float(eps), ptr(ws), ws.numel() * 8, stream()), "msgm_groupnorm_affine")
    return scale, shift


def conv_b6_supported(geom: L.ConvGeomT, C0: int, C1: int, CoutP: int) -> bool:
    return bool(lib().msgm_conv_b6_supported(geom, int(C0), int(C1), int(CoutP)))


def b6_split_weights(Wp: torch.Tensor, Wb: torch.Tensor):
    """Wb [3][n] (bfloat16) <- the three bf16 pieces (h, m, l) of every element of the packed fp32 image Wp [n]."""
    n = Wp.numel()
    if Wb.dtype != torch.bfloat16 or Wb.numel() < 3 * n:
        raise MsgmError("b6_split_weights: Wb must be bfloat16 [3][n]")
    check(lib().msgm_b6_split_weights(ptr(f32(Wp)), ptr(Wb), n, stream()), "msgm_b6_split_weights")


def conv_wino_supported(geom: L.ConvGeomT, C0: int, C1: int, CoutP: int) -> bool:
    return bool(lib().msgm_conv_wino_supported(geom, int(C0), int(C1), int(CoutP)))


def conv_wgrad(geom: L.ConvGeomT, gy: torch.Tensor, src: torch.Tensor, C: int, koff: int, dWp: torch.Tensor, Cout: int,
               CoutP: int, Ktot: int, dbias: Optional[torch.Tensor] = None, n_bias: int = 0, tapmask_c32=None,
               tapmask_co32=None):
    """dbias (Cout floats, ACCUMULATED into): bias gradient over the primal rows n < n_bias, as a by-product."""
    taps = geom.KH * geom.KW
    if dbias is not None and (dbias.numel() != Cout or n_bias <= 0 or n_bias > geom.N):
        raise MsgmError("wgrad: bad bias-gradient arguments")
    if gy.numel() != geom.N * geom.Ho * geom.Wo * Cout or src.numel() != geom.N * geom.Hi * geom.Wi * C:
        raise MsgmError("wgrad operands do not match the geometry")
    if dWp.numel() < taps * CoutP * Ktot:
        raise MsgmError("packed gradient too small")
    import ctypes as C_
    mc = (C_.c_uint16 * len(tapmask_c32))(*tapmask_c32) if tapmask_c32 else None
    mo = (C_.c_uint16 * len(tapmask_co32))(*tapmask_co32) if tapmask_co32 else None
    if (mc is not None and len(mc) < (C + 31) // 32) or (mo is not None and len(mo) < (Cout + 31) // 32):
        raise MsgmError("wgrad: tap masks need one entry per 32-channel block")
    if os.environ.get("MSGM_ATOMIC_WGRAD"):            # diagnostic A/B: float atomics across the position chunks
        check(lib().msgm_conv_wgrad(geom, ptr(f32(gy)), ptr(f32(src)), C, koff, ptr(f32(dWp)), Cout, CoutP, Ktot,
                                    ptr(dbias), int(n_bias), mc, mo, stream()), "msgm_conv_wgrad")
        return
    # default: deterministic — per-workgroup slabs added in slot order (no float atomics; same bits every run)
    need = int(lib().msgm_conv_wgrad_workspace(geom, C, Cout, CoutP, int(n_bias) if dbias is not None else 0))
    d = DeferredReduces.active
    if d is not None and d.device == gy.device:
        # inside a backward pass that batches its slot reductions: slabs go to the pass's arena, the reduction is described
        # to the pass and runs with all the others in one launch (DeferredReduces.flush)
        ws, nbytes = d.take(need)
        jobs, nj = (L.ReduceJobT * 2)(), C_.c_int32(0)
        check(lib().msgm_conv_wgrad_slabs(geom, ptr(f32(gy)), ptr(f32(src)), C, koff, ptr(f32(dWp)), Cout, CoutP, Ktot,
                                          ptr(dbias), int(n_bias), mc, mo, ws, nbytes, jobs, C_.byref(nj), stream()),
              "msgm_conv_wgrad_slabs")
        d.add(jobs, nj.value, (dWp, dbias))
        return
    ws = scratch(gy.device, need, "wgrad")
    check(lib().msgm_conv_wgrad_det(geom, ptr(f32(gy)), ptr(f32(src)), C, koff, ptr(f32(dWp)), Cout, CoutP, Ktot,
                                    ptr(dbias), int(n_bias), mc, mo, ptr(ws), ws.numel() * 4, stream()), "msgm_conv_wgrad_det")


class DeferredReduces:
    """One backward pass's slot reductions (deterministic weight / bias gradients) batched into ONE launch.
    ``with DeferredReduces.on(device): ... backward ...`` makes every conv_wgrad inside write its per-workgroup slabs into a
    bump arena that lives until the end of the pass and register its reduction; leaving the block uploads the job table
    (only when it differs from the cached one: the arena hands out the same addresses every step, so under a captured
    hipGraph nothing is uploaded) and runs msgm_slot_reduce_batched.  Same arithmetic and order as the per-call form."""
    active = None
    _cache = {}                      # device -> DeferredReduces (arena chunks and the device job table persist)
    CHUNK = 16 << 20                 # arena granule: chunks are sized from what the pass asks for, rounded up to this

    def __init__(self, device):
        self.device = torch.device(device)
        self.chunks, self.jobs, self.keep = [], [], []
        self.tables = {}             # job table bytes -> device copy
        self.reset()

    @classmethod
    def on(cls, device):
        device = torch.device(device)
        if device.index is None and device.type == "cuda":
            device = torch.device("cuda", torch.cuda.current_device())
        d = cls._cache.get(device)
        if d is None:
            d = cls._cache[device] = cls(device)
        return d

    def reset(self):
        self.ci, self.off, self.jobs, self.keep = 0, 0, [], []

    def take(self, nbytes: int):
        nbytes = (int(nbytes) + 255) & ~255
        while True:
            if self.ci < len(self.chunks) and self.off + nbytes <= self.chunks[self.ci].numel() * 4:
                p = self.chunks[self.ci].data_ptr() + self.off
                self.off += nbytes
                return p, nbytes
            if self.ci < len(self.chunks):
                self.ci, self.off = self.ci + 1, 0
                continue
            if torch.cuda.is_current_stream_capturing():
                raise MsgmError("the slab arena must reach its size in an eager step before graph capture")
            # grow geometrically from the measured need (a tiny UNet1D / smoke run keeps 16 MB, the C4 backward ends
            # near its real footprint) instead of pinning 256 MB per device for every pass
            have = sum(c.numel() * 4 for c in self.chunks)
            want = max(nbytes, have // 2, self.CHUNK)
            want = (want + self.CHUNK - 1) // self.CHUNK * self.CHUNK
            self.chunks.append(torch.empty(want // 4, dtype=torch.float32, device=self.device))

    @classmethod
    def release(cls, device=None):
        """Drop the arena(s) and job tables (bench legs / tests that want the memory back).  Any hipGraph captured with
        the old arena must be dropped first: it holds the addresses."""
        if cls.active is not None:
            raise MsgmError("DeferredReduces.release inside a pass")
        if device is None:
            cls._cache.clear()
        else:
            device = torch.device(device)
            if device.index is None and device.type == "cuda":
                device = torch.device("cuda", torch.cuda.current_device())
            cls._cache.pop(device, None)

    def add(self, jobs, n, keep):
        for i in range(n):
            self.jobs.append(L.ReduceJobT.from_buffer_copy(jobs[i]))
        self.keep.append(keep)

    def __enter__(self):
        if DeferredReduces.active is not None:
            raise MsgmError("DeferredReduces does not nest")
        self.reset()
        DeferredReduces.active = self
        return self

    def __exit__(self, et, ev, tb):
        DeferredReduces.active = None
        if et is None:
            self.flush()
        self.jobs, self.keep = [], []
        return False

    def flush(self):
        n = len(self.jobs)
        if n == 0:
            return
        arr = (L.ReduceJobT * n)()
        blk = 0
        for i, j in enumerate(self.jobs):
            arr[i] = j
            arr[i].block_begin = blk
            blk += (j.n_elem + 31) // 32 + (j.n_elem2 + 31) // 32
        raw = bytes(arr)
        tab = self.tables.get(raw)
        if tab is None:
            # every table ever uploaded stays alive: a captured hipGraph keeps the address of the one it was captured with
            if torch.cuda.is_current_stream_capturing():
                raise MsgmError("the reduction job table changed inside a graph capture (run one eager step first)")
            tab = self.tables[raw] = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        check(lib().msgm_slot_reduce_batched(ptr(tab), n, blk, stream()), "msgm_slot_reduce_batched")


class EmbBank:
    """The time-embedding projections of all ResBlocks of a U-Net (emb_layers[1], model/unet.py:145-151) as ONE launch per
    direction (msgm_emb_bank_forward / _backward) on the PyTorch-layout parameters.  ``items`` = [(weight (co, K), bias (co),
    conv_bias or None)] in execution order; conv_bias is the bias of the conv whose output the projection is added to
    (model/unet.py:179-180): it has the same gradient, so the bank writes both.  ``out[i]`` / ``dout[i]`` are the per-block
    [rows][co] views the ResBlocks read / write."""

    def __init__(self, items, K: int):
        self.items, self.K = list(items), int(K)
        self.cos = [int(w.shape[0]) for w, _, _ in self.items]
        self.blocks, nb = [], 0
        for co in self.cos:
            self.blocks.append(nb)
            nb += (co + 31) // 32
        self.total_blocks = nb
        self.sig, self.table, self.out, self.dout, self._keep = None, None, None, None, []

    def _prepare(self, rows: int, dev):
        grads = tuple((w.grad.data_ptr() if w.grad is not None else 0) for w, _, _ in self.items)
        sig = (rows, str(dev), tuple(w.data_ptr() for w, _, _ in self.items), grads)
        if sig == self.sig:
            return
        if torch.cuda.is_current_stream_capturing():
            raise MsgmError("the embedding bank's job table must be built in an eager step before graph capture")
        tot = sum(self.cos) * rows
        if self.table is not None:
            self._keep.append((self.table, self.eo_all, self.deo_all))     # a captured graph may still hold the old addresses
        self.eo_all = torch.empty(tot, dtype=torch.float32, device=dev)
        self.deo_all = torch.zeros(tot, dtype=torch.float32, device=dev)
        arr = (L.EmbJobT * len(self.items))()
        self.out, self.dout, off = [], [], 0
        for i, ((w, b, cb), co) in enumerate(zip(self.items, self.cos)):
            o, d = self.eo_all[off:off + rows * co], self.deo_all[off:off + rows * co]
            self.out.append(o); self.dout.append(d)
            g = lambda t: (t.grad.data_ptr() if (t is not None and t.grad is not None) else None)
            arr[i] = L.EmbJobT(w.data_ptr(), b.data_ptr() if b is not None else None, o.data_ptr(), d.data_ptr(), g(w), g(b), g(cb),
                               co, self.blocks[i])
            off += rows * co
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.sig = sig

    def forward(self, semb: torch.Tensor, rows: int, n_bias: int):
        if semb.numel() != rows * self.K:
            raise MsgmError("embedding bank: semb must be [rows][K]")
        self._prepare(rows, semb.device)
        check(lib().msgm_emb_bank_forward(ptr(self.table), len(self.items), self.total_blocks, ptr(f32(semb)), rows, self.K,
                                          int(n_bias), stream()), "msgm_emb_bank_forward")
        return self.out

    def backward(self, semb: torch.Tensor, dsemb: torch.Tensor, rows: int, n_bias: int):
        """dout[i] filled by the ResBlocks' backward; writes every weight.grad / bias.grad / conv_bias.grad and dsemb."""
        self._prepare(rows, semb.device)
        if any(w.grad is None for w, _, _ in self.items) or dsemb.numel() != rows * self.K:
            raise MsgmError("embedding bank backward: missing .grad or bad dsemb")
        check(lib().msgm_emb_bank_backward(ptr(self.table), len(self.items), self.total_blocks, ptr(f32(semb)), ptr(dsemb), rows,
                                           self.K, int(n_bias), stream()), "msgm_emb_bank_backward")


def pack_weight(W: torch.Tensor, w_off: int, Wp: torch.Tensor, rows, ncols, col_off, taps, sr, sc, st, rowsP, Ktot, kp_off):
    need = w_off + (rows - 1) * sr + (col_off + ncols - 1) * sc + (taps - 1) * st + 1
    if need > W.numel() or Wp.numel() < taps * rowsP * Ktot:
        raise MsgmError("pack_weight out of range")
    check(lib().msgm_pack_weight(ptr(f32(W)) + 4 * w_off, ptr(f32(Wp)), rows, ncols, col_off, taps, sr, sc, st, rowsP, Ktot,
                                 kp_off, stream()), "msgm_pack_weight")


class PackTable:
    """Device-resident table of (un)pack jobs — every weight image of a network in one launch.
    job = (W, w_off, Wp, rows, ncols, col_off, taps, sr, sc, st, rowsP, Ktot, kp_off), as pack_weight's arguments."""

    def __init__(self, jobs, device):
        import ctypes as C
        arr = (L.PackJobT * len(jobs))()
        self.keep = []
        for i, job in enumerate(jobs):
            W, w_off, Wp, rows, ncols, col_off, taps, sr, sc, st, rowsP, Ktot, kp_off = job[:13]
            acc = int(bool(job[13])) if len(job) > 13 else 0          # unpack only: add (atomically) instead of overwrite
            span = (taps - 1) * st                              # st < 0: the taps are walked backwards from w_off
            need = w_off + (rows - 1) * sr + (col_off + ncols - 1) * sc + max(span, 0) + 1
            if (need > W.numel() or w_off + min(span, 0) < 0 or Wp.numel() < taps * rowsP * Ktot or kp_off + ncols > Ktot
                    or rowsP < rows):
                raise MsgmError("pack table: job out of range")
            arr[i] = L.PackJobT(ptr(f32(W)) + 4 * w_off, ptr(f32(Wp)), sr, sc, st, rows, ncols, col_off, taps, rowsP, Ktot,
                                kp_off, acc)
            self.keep.append((W, Wp))
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.table = raw.to(device)
        self.n = len(jobs)

    def run(self, unpack: bool = False):
        check(lib().msgm_pack_weights_batched(ptr(self.table), self.n, int(bool(unpack)), stream()), "msgm_pack_weights_batched")

    def run_wino(self):
        """The jobs are 3x3 kernels (taps = 9) and Wp the [16][rowsP][Ktot] Winograd images: writes G g G^T."""
        check(lib().msgm_wino_pack_weights_batched(ptr(self.table), self.n, stream()), "msgm_wino_pack_weights_batched")


def unpack_weight(dW: torch.Tensor, w_off: int, dWp: torch.Tensor, rows, ncols, col_off, taps, sr, sc, st, rowsP, Ktot,
                  kp_off, accumulate=False):
    need = w_off + (rows - 1) * sr + (col_off + ncols - 1) * sc + (taps - 1) * st + 1
    if need > dW.numel() or dWp.numel() < taps * rowsP * Ktot:
        raise MsgmError("unpack_weight out of range")
    check(lib().msgm_unpack_weight(ptr(f32(dW)) + 4 * w_off, ptr(f32(dWp)), rows, ncols, col_off, taps, sr, sc, st, rowsP,
                                   Ktot, kp_off, int(bool(accumulate)), stream()), "msgm_unpack_weight")


ACT_GELU, ACT_SILU = 0, 1


def act_dual_forward(act: int, z: torch.Tensor, h: torch.Tensor, dual: bool) -> torch.Tensor:
    half = z.numel() // 2 if dual else z.numel()
    if h.numel() != z.numel() or half % 4:
        raise MsgmError("act_dual_forward: sizes must match and be multiples of 4 per half")
    check(lib().msgm_act_dual_forward(act, ptr(f32(z)), ptr(f32(h)), half, int(bool(dual)), stream()), "msgm_act_dual_forward")
    return h


def act_dual_backward(act: int, z: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    half = z.numel() // 2
    if g.numel() != z.numel() or half % 4 or z.numel() % 2:
        raise MsgmError("act_dual_backward: sizes must match and be multiples of 4 per half")
    check(lib().msgm_act_dual_backward(act, ptr(f32(z)), ptr(f32(g)), half, stream()), "msgm_act_dual_backward")
    return g


def colsum(x: torch.Tensor, N: int, P: int, C: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if x.numel() < N * P * C:
        raise MsgmError("colsum: tensor too small")
    out = torch.empty(N, C, dtype=torch.float32, device=x.device) if out is None else out
    if os.environ.get("MSGM_ATOMIC_WGRAD"):
        check(lib().msgm_colsum(ptr(f32(x)), ptr(out), N, P, C, stream()), "msgm_colsum")
        return out
    need = int(lib().msgm_colsum_workspace(N, P, C))
    ws = scratch(x.device, need, "colsum") if need else None
    check(lib().msgm_colsum_det(ptr(f32(x)), ptr(out), N, P, C, ptr(ws), need, stream()), "msgm_colsum_det")
    return out


def gather_row(x: torch.Tensor, N: int, P: int, C: int, pos: int) -> torch.Tensor:
    if x.numel() < N * P * C:
        raise MsgmError("gather_row: tensor too small")
    out = torch.empty(N, C, dtype=torch.float32, device=x.device)
    check(lib().msgm_gather_row(ptr(f32(x)), ptr(out), N, P, C, pos, stream()), "msgm_gather_row")
    return out


def add_row(x: torch.Tensor, E: torch.Tensor, N: int, P: int, C: int, pos: int, sgn: float):
    if x.numel() < N * P * C or E.numel() != N * C:
        raise MsgmError("add_row: size mismatch")
    check(lib().msgm_add_row(ptr(f32(x)), ptr(f32(E)), N, P, C, pos, float(sgn), stream()), "msgm_add_row")


# ---------------------------------------------------------------- 2-D U-Net ops
_GN_WS = {}


def _gn_ws(Bp: int, G: int, device) -> torch.Tensor:
    """Per-device scratch for the GroupNorm moment accumulators (consumed in stream order)."""
    n = int(lib().msgm_groupnorm_workspace(Bp, G)) // 8
    key = (torch.device(device), n)
    if key not in _GN_WS:
        _GN_WS[key] = torch.empty(n, dtype=torch.float64, device=device)     # per-chunk moment slots: no zero contract
    return _GN_WS[key]


def groupnorm_dual_forward(x, gamma, beta, Bp, P, C, G, dual, silu, stats=None, out=None, eps=1e-5):
    N = 2 * Bp if dual else Bp
    if x.numel() != N * P * C or gamma.numel() != C or beta.numel() != C:
        raise MsgmError("groupnorm: size mismatch")
    if stats is not None and stats.numel() != Bp * G * 4:
        raise MsgmError("groupnorm: stats must be [Bp][G][4]")
    out = torch.empty_like(x) if out is None else out
    ws = _gn_ws(Bp, G, x.device)
    check(lib().msgm_groupnorm_dual_forward(ptr(f32(x)), ptr(f32(gamma)), ptr(f32(beta)), ptr(out), ptr(stats), Bp, P, C, G,
                                            int(bool(dual)), int(bool(silu)), float(eps), ptr(ws), ws.numel() * 8, stream()),
          "msgm_groupnorm_dual_forward")
    return out


def _gn_backward_slots(x0, C0, x1, C1, gamma, beta, stats, gout, gx0, gx1, dgamma, dbeta, Bp, P, G, silu, eps, residual, d,
                       residual2=None):
    """GroupNorm backward inside a DeferredReduces pass: the dgamma / dbeta partials go to the pass's arena and their
    slot-ordered sums join the pass's ONE batched reduction launch."""
    import ctypes as C_
    C = C0 + C1
    need = int(lib().msgm_groupnorm_param_slots_bytes(Bp, P, C))
    ps, nbytes = d.take(need)
    ws = _gn_ws(Bp, G, x0.device)
    jobs, nj = (L.ReduceJobT * 2)(), C_.c_int32(0)
    check(lib().msgm_groupnorm_dual_backward_slots(ptr(f32(x0)), C0, ptr(x1), C1, ptr(f32(gamma)), ptr(f32(beta)), ptr(f32(stats)),
                                                   ptr(f32(gout)), ptr(gx0), ptr(gx1), ptr(dgamma), ptr(dbeta), Bp, P, G,
                                                   int(bool(silu)), float(eps), ptr(residual), ptr(residual2), ptr(ws), ws.numel() * 8, ps, nbytes,
                                                   jobs, C_.byref(nj), stream()), "msgm_groupnorm_dual_backward_slots")
    d.add(jobs, nj.value, (dgamma, dbeta))


def groupnorm_dual_backward(x, gamma, beta, stats, gout, dgamma, dbeta, Bp, P, C, G, silu, gx=None, eps=1e-5, residual=None,
                            residual2=None):
    """``residual`` (same shape as x) is added to the returned cotangent in the apply pass (skip branch, no extra axpy);
    ``residual2``: a second addend (the skip-stack cotangent) — in the same pass inside a DeferredReduces backward, by one
    lincomb otherwise."""
    if x.numel() != 2 * Bp * P * C or gout.numel() != x.numel() or stats.numel() != Bp * G * 4:
        raise MsgmError("groupnorm backward: size mismatch")
    if dgamma.numel() != C or dbeta.numel() != C:
        raise MsgmError("groupnorm backward: dgamma/dbeta size")
    gx = gout if gx is None else gx
    if (residual is not None and residual.numel() != x.numel()) or (residual2 is not None and residual2.numel() != x.numel()):
        raise MsgmError("groupnorm backward: residual size")
    d = DeferredReduces.active
    if d is not None and d.device == x.device:
        _gn_backward_slots(x, C, None, 0, gamma, beta, stats, gout, gx, None, dgamma, dbeta, Bp, P, G, silu, eps, residual, d,
                           residual2=residual2)
        return gx
    ws = _gn_ws(Bp, G, x.device)
    check(lib().msgm_groupnorm_dual_backward(ptr(f32(x)), ptr(f32(gamma)), ptr(f32(beta)), ptr(f32(stats)), ptr(f32(gout)),
                                             ptr(gx), ptr(dgamma), ptr(dbeta), Bp, P, C, G, int(bool(silu)), float(eps),
                                             ptr(residual), ptr(ws), ws.numel() * 8, stream()), "msgm_groupnorm_dual_backward")
    if residual2 is not None:
        lincomb(gx, gx, 1.0, residual2, 1.0)
    return gx


def groupnorm_dual_forward2(x0, C0, x1, C1, gamma, beta, Bp, P, G, dual, silu, stats=None, eps=1e-5):
    """GroupNorm(+SiLU) of the channel concatenation [x0 | x1] without materialising it: ONE normalised output tensor."""
    n = (2 if dual else 1) * Bp * P
    if x0.numel() != n * C0 or x1.numel() != n * C1:
        raise MsgmError("groupnorm2: size mismatch")
    out = torch.empty(n * (C0 + C1), dtype=torch.float32, device=x0.device)
    ws = _gn_ws(Bp, G, x0.device)
    check(lib().msgm_groupnorm_dual_forward2(ptr(f32(x0)), C0, ptr(f32(x1)), C1, ptr(f32(gamma)), ptr(f32(beta)), ptr(out), ptr(stats),
                                             Bp, P, G, int(bool(dual)), int(bool(silu)), float(eps), ptr(ws), ws.numel() * 8, stream()),
          "msgm_groupnorm_dual_forward2")
    return out


def groupnorm_dual_backward2(x0, C0, x1, C1, gamma, beta, stats, gout, dgamma, dbeta, Bp, P, G, silu, eps=1e-5):
    """Backward of ``groupnorm_dual_forward2``: the input cotangent as two tensors shaped like x0 / x1."""
    if gout.numel() != 2 * Bp * P * (C0 + C1) or x0.numel() != 2 * Bp * P * C0 or x1.numel() != 2 * Bp * P * C1:
        raise MsgmError("groupnorm2 backward: size mismatch")
    gx0, gx1 = torch.empty_like(x0), torch.empty_like(x1)
    d = DeferredReduces.active
    if d is not None and d.device == x0.device:
        _gn_backward_slots(x0, C0, x1, C1, gamma, beta, stats, gout, gx0, gx1, dgamma, dbeta, Bp, P, G, silu, eps, None, d)
        return gx0, gx1
    ws = _gn_ws(Bp, G, x0.device)
    check(lib().msgm_groupnorm_dual_backward2(ptr(f32(x0)), C0, ptr(f32(x1)), C1, ptr(f32(gamma)), ptr(f32(beta)), ptr(f32(stats)),
                                              ptr(f32(gout)), ptr(gx0), ptr(gx1), ptr(dgamma), ptr(dbeta), Bp, P, G, int(bool(silu)),
                                              float(eps), ptr(ws), ws.numel() * 8, stream()), "msgm_groupnorm_dual_backward2")
    return gx0, gx1


def bmm(A, a_off, B, b_off, Cm, c_off, M, N, K, batch, sA, sB, sC, alpha=1.0, accumulate=False, pair2=None, third=None):
    """C[b](i,j) (+)= alpha (sum_k A[b](i,k) B[b](k,j) [+ A2.B2]); sA = (batch, i, k), sB = (batch, k, j),
    sC = (batch, i, j) element strides; *_off are element offsets into the given tensors (channel slices of a fused
    qkv tensor).  pair2 = (A2, a2_off, B2, b2_off): a second product with the same strides, summed in registers."""
    def span(off, s, dims):
        return off + sum((d - 1) * st for d, st in zip(dims, s)) + 1
    if span(a_off, sA, (batch, M, K)) > A.numel() or span(b_off, sB, (batch, K, N)) > B.numel() or \
            span(c_off, sC, (batch, M, N)) > Cm.numel():
        raise MsgmError("bmm: strides run past the end of a tensor")
    pa2 = pb2 = None
    if pair2 is not None:
        A2, a2_off, B2, b2_off = pair2
        if span(a2_off, sA, (batch, M, K)) > A2.numel() or span(b2_off, sB, (batch, K, N)) > B2.numel():
            raise MsgmError("bmm: second pair runs past the end of a tensor")
        pa2, pb2 = ptr(f32(A2)) + 4 * a2_off, ptr(f32(B2)) + 4 * b2_off
    pb3 = pc3 = None
    if third is not None:        # third = (B3, b3_off, C3, c3_off): second output C3 = alpha A.B3, A streamed once
        B3, b3_off, C3, c3_off = third
        if span(b3_off, sB, (batch, K, N)) > B3.numel() or span(c3_off, sC, (batch, M, N)) > C3.numel():
            raise MsgmError("bmm: third operand / second output runs past the end of a tensor")
        pb3, pc3 = ptr(f32(B3)) + 4 * b3_off, ptr(f32(C3)) + 4 * c3_off
    check(lib().msgm_bmm_dual(ptr(f32(A)) + 4 * a_off, ptr(f32(B)) + 4 * b_off, pa2, pb2, pb3, ptr(f32(Cm)) + 4 * c_off, pc3,
                              M, N, K, batch, sA[0], sA[1], sA[2], sB[0], sB[1], sB[2], sC[0], sC[1], sC[2], float(alpha),
                              int(bool(accumulate)), stream()), "msgm_bmm")


def softmax_dual_forward(S, T, Wd=None, Pd=None):
    rows = S.numel() // T
    dual = Wd is not None
    if dual and (Wd.numel() != S.numel() or Pd.numel() != S.numel()):
        raise MsgmError("softmax: size mismatch")
    check(lib().msgm_softmax_dual_forward(ptr(f32(S)), ptr(Wd), ptr(Pd), rows, T, int(dual), stream()), "msgm_softmax_dual_forward")


def rbf_kernel(x: torch.Tensor, y: torch.Tensor, want_matrix: bool = False, want_sum: bool = True):
    """exp(-mean_d((x_i - y_j)^2) / d) over all pairs: (K or None, sum as a 1-element float64 tensor or None)."""
    if x.dim() != 2 or y.dim() != 2 or x.shape[1] != y.shape[1]:
        raise MsgmError("rbf_kernel: x (Nx,d) and y (Ny,d) expected")
    x, y = f32(x), f32(y)
    Kmat = torch.empty(x.shape[0], y.shape[0], dtype=torch.float32, device=x.device) if want_matrix else None
    ssum = torch.empty(1, dtype=torch.float64, device=x.device) if want_sum else None
    check(lib().msgm_rbf_kernel(ptr(x), ptr(y), x.shape[0], y.shape[0], x.shape[1], ptr(Kmat), ptr(ssum), stream()),
          "msgm_rbf_kernel")
    return Kmat, ssum


def attention_supported(T, C) -> bool:
    return bool(lib().msgm_attention_supported(int(T), int(C)))


def attention_forward(qkv, out, N, T, C, scale):
    """Fused softmax(scale q k^T) v on channels-last qkv [N][T][3C] -> out [N][T][C] (no tangent)."""
    if qkv.numel() < N * T * 3 * C or out.numel() < N * T * C:
        raise MsgmError("attention: buffer too small")
    check(lib().msgm_attention_forward(ptr(f32(qkv)), ptr(f32(out)), N, T, C, float(scale), stream()), "msgm_attention_forward")
    return out


def attention_dual_supported(T, C) -> bool:
    return bool(lib().msgm_attention_dual_supported(int(T), int(C)))


def attention_dual_forward(qkv, Bp, T, C, scale):
    """Training-path attention on dual numbers: qkv [2Bp][T][3C] -> (att [2Bp][T][C] = o | odot, stats [2][Bp*T])."""
    if qkv.numel() < 2 * Bp * T * 3 * C:
        raise MsgmError("attention_dual: qkv too small")
    att = torch.empty(2 * Bp * T * C, dtype=torch.float32, device=qkv.device)
    stats = torch.empty(2 * Bp * T, dtype=torch.float32, device=qkv.device)
    check(lib().msgm_attention_dual_forward(ptr(f32(qkv)), ptr(att), ptr(stats), Bp, T, C, float(scale), stream()),
          "msgm_attention_dual_forward")
    return att, stats


_HIP_NODE_KINDS = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event",
                   7: "event_record", 8: "ext_semas_signal", 9: "ext_semas_wait", 10: "mem_alloc", 11: "mem_free",
                   12: "memcpy_from_symbol", 13: "memcpy_to_symbol"}


def new_graph():
    """A torch hipGraph wrapper that KEEPS the captured hipGraph_t next to the executable one, so that graph_node_kinds()
    can list what was captured."""
    return torch.cuda.CUDAGraph(keep_graph=True)


def graph_node_kinds(graph) -> dict:
    """{node kind: count} of a captured graph made by new_graph().  The captured steps of this package are meant to be
    kernel nodes only: a hipMemsetAsync / D2D hipMemcpyAsync node can lose its ordering against the kernel nodes around it
    when a replay starts on an idle GPU (ROCm 7.2; see msgm_zero_async in csrc/common.h), so tests assert on this."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    g = ctypes.c_void_p(graph.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    if hip.hipGraphGetNodes(g, None, ctypes.byref(n)) != 0:
        raise MsgmError("hipGraphGetNodes failed")
    nodes = (ctypes.c_void_p * max(n.value, 1))()
    if hip.hipGraphGetNodes(g, nodes, ctypes.byref(n)) != 0:
        raise MsgmError("hipGraphGetNodes failed")
    kinds = {}
    for i in range(n.value):
        t = ctypes.c_int(-1)
        if hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t)) != 0:
            raise MsgmError("hipGraphNodeGetType failed")
        k = _HIP_NODE_KINDS.get(t.value, f"kind{t.value}")
        kinds[k] = kinds.get(k, 0) + 1
    return kinds


_SCRATCH = {}
_SCRATCH_KEEP = []          # outgrown workspaces stay alive: a captured hipGraph may still hold their address


def scratch(device, nbytes: int, tag: str) -> torch.Tensor:
    """Per-(device, purpose) workspace that grows to the largest request.  A step enqueues its kernels one after the
    other on one stream, so consecutive ops of one purpose share it.  It must reach its final size in an eager step
    BEFORE a hipGraph capture (the trainers / samplers run one eager step first): a captured graph keeps the address."""
    key = (torch.device(device), tag)
    ws = _SCRATCH.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise MsgmError(f"{tag} workspace must exist before graph capture (run one eager step first)")
        if ws is not None:
            _SCRATCH_KEEP.append(ws)
        ws = _SCRATCH[key] = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=device)
    return ws


def attention_dual_backward(qkv, att, datt, stats, Bp, T, C, scale):
    """dqkv [2Bp][T][3C] from datt = [obar ; odbar]; the slab workspace is cached per device and grows to the largest use
    (one step runs its attention blocks one after the other on one stream, so they share it)."""
    if min(qkv.numel() // 3, att.numel(), datt.numel()) < 2 * Bp * T * C or stats.numel() < 2 * Bp * T:
        raise MsgmError("attention_dual backward: buffer too small")
    need = int(lib().msgm_attention_dual_workspace(Bp, T, C))
    ws = scratch(qkv.device, need, "attention")
    dqkv = torch.empty(2 * Bp * T * 3 * C, dtype=torch.float32, device=qkv.device)
    check(lib().msgm_attention_dual_backward(ptr(f32(qkv)), ptr(f32(att)), ptr(f32(datt)), ptr(f32(stats)), ptr(dqkv), Bp, T, C,
                                             float(scale), ptr(ws), ws.numel() * 4, stream()), "msgm_attention_dual_backward")
    return dqkv


def softmax_dual_backward(Pm, Wd, Pb, Pdb, T):
    rows = Pm.numel() // T
    if not (Wd.numel() == Pb.numel() == Pdb.numel() == Pm.numel()):
        raise MsgmError("softmax backward: size mismatch")
    check(lib().msgm_softmax_dual_backward(ptr(f32(Pm)), ptr(f32(Wd)), ptr(f32(Pb)), ptr(f32(Pdb)), rows, T, stream()),
          "msgm_softmax_dual_backward")


def timestep_embedding(t, dim, max_period=10000.0):
    t = t.reshape(-1).contiguous()
    emb = torch.empty(t.numel(), dim, dtype=torch.float32, device=t.device)
    check(lib().msgm_timestep_embedding(ptr(f32(t)), ptr(emb), t.numel(), dim, float(max_period), stream()), "msgm_timestep_embedding")
    return emb


def timestep_embedding_dual(t, Bp, dim, max_period=10000.0):
    """t = [t ; tdot] (2*Bp) -> [emb ; embdot] (2*Bp, dim)."""
    t = t.reshape(-1).contiguous()
    if t.numel() != 2 * Bp:
        raise MsgmError("timestep_embedding_dual: t must hold 2*Bp values")
    emb = torch.empty(2 * Bp, dim, dtype=torch.float32, device=t.device)
    check(lib().msgm_timestep_embedding_dual(ptr(f32(t)), ptr(emb), Bp, dim, float(max_period), stream()), "msgm_timestep_embedding_dual")
    return emb


def normalize_dual(x, Bp, n, dual, scale):
    """NormalizeLogRadius (+ rescale) on a stacked (N, n) input; returns (out (N,n), logr (N,))."""
    N = 2 * Bp if dual else Bp
    if x.numel() != N * n:
        raise MsgmError("normalize_dual: size mismatch")
    out = torch.empty(N, n, dtype=torch.float32, device=x.device)
    logr = torch.empty(N, dtype=torch.float32, device=x.device)
    check(lib().msgm_normalize_dual(ptr(f32(x)), ptr(out), ptr(logr), Bp, n, int(bool(dual)), float(scale), 1e-6, stream()),
          "msgm_normalize_dual")
    return out, logr


def flat_to_image(flat, B, C, H, W, forder, scale):
    if flat.numel() != B * C * H * W:
        raise MsgmError("flat_to_image: size mismatch")
    img = torch.empty(B * H * W * C, dtype=torch.float32, device=flat.device)
    check(lib().msgm_flat_to_image(ptr(f32(flat)), ptr(img), B, C, H, W, int(bool(forder)), float(scale), stream()), "msgm_flat_to_image")
    return img


def image_to_flat(img, B, C, H, W, forder, scale):
    if img.numel() != B * C * H * W:
        raise MsgmError("image_to_flat: size mismatch")
    flat = torch.empty(B, C * H * W, dtype=torch.float32, device=img.device)
    check(lib().msgm_image_to_flat(ptr(f32(img)), ptr(flat), B, C, H, W, int(bool(forder)), float(scale), stream()), "msgm_image_to_flat")
    return flat


def sum2x2(x, N, H, W, C):
    if x.numel() != N * 4 * H * W * C:
        raise MsgmError("sum2x2: size mismatch")
    out = torch.empty(N * H * W * C, dtype=torch.float32, device=x.device)
    check(lib().msgm_sum2x2(ptr(f32(x)), ptr(out), N, H, W, C, stream()), "msgm_sum2x2")
    return out
