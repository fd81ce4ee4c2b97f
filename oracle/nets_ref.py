"""Oracle: functional score networks driven by a reference-keyed state_dict.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The reference builds
``nn.Module`` trees (NN.py, NNUnet.py, NNUnet1D.py, model/unet.py); this
file restates their forward arithmetic as pure functions of
``(params: dict[str, Tensor], x, t)`` where ``params`` uses the reference's
``state_dict`` key names.  Being pure functions they can be pushed through
``torch.func.jvp`` / ``torch.autograd`` for the SSM oracle.

Citations are ``file:line`` relative to /root/reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


def swish(x):
    """sigmoid(x)*x.  NN.py:52-53, model/nn_utils.py:44-46."""
    return torch.sigmoid(x) * x


def normalize_log_radius(x, eps=1e-6):
    """x -> (x/(||x||+eps), log(||x||+eps)) over the last axis.  NN.py:56-70."""
    r = torch.norm(x, dim=-1, keepdim=True) + eps
    return x / r, torch.log(r)


def sinusoidal_embedding(t, dim, max_period=10000):
    """[cos(t f_j), sin(t f_j)], f_j = exp(-ln(max_period) j/half).
    model/nn_utils.py:130-148 (t is used as given — NNUnet.py:97 passes
    continuous t in [0,1], and log||x|| at :104)."""
    half = dim // 2
    dt = torch.float64 if t.dtype == torch.float64 else torch.float32   # reference: always fp32
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half).to(t.device)
    args = t[:, None].to(dt) * freqs[None].to(dt)
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _lin(p: Params, key: str, x):
    return F.linear(x, p[key + ".weight"], p.get(key + ".bias"))


# --------------------------------------------------------------------------
# MLP                                                           NN.py:73-120
# --------------------------------------------------------------------------

def mlp_forward(p: Params, x, t, premodule: Optional[str] = None):
    """[x | x^, log r] (+) t -> 3x(Linear+Swish) -> Linear.  NN.py:108-120."""
    shape = x.shape
    d = p["main.6.weight"].shape[0]
    h = x.reshape(-1, d)
    t = t.reshape(-1, 1).to(h.dtype)
    if premodule == "NormalizeLogRadius":
        xn, lr = normalize_log_radius(h)
        h = torch.cat([xn, lr], dim=-1)
    h = torch.cat([h, t], dim=1)
    for i in (0, 2, 4):
        h = swish(_lin(p, f"main.{i}", h))
    return _lin(p, "main.6", h).reshape(shape)


# --------------------------------------------------------------------------
# 1-D U-Net                                                NNUnet1D.py:13-179
# --------------------------------------------------------------------------

def _conv_block_1d(p: Params, key: str, h):
    """conv k3 p1 -> GELU(erf) -> conv k3 p1 -> GELU.  NNUnet1D.py:13-24."""
    h = F.gelu(F.conv1d(h, p[key + ".net.0.weight"], p[key + ".net.0.bias"], padding=1))
    return F.gelu(F.conv1d(h, p[key + ".net.2.weight"], p[key + ".net.2.bias"], padding=1))


def unet1d_forward(p: Params, x, t, premodule: Optional[str] = None, levels: int = 3):
    """NNUnet1D.py:110-179.  The time embedding (and optional log-radius
    embedding) is broadcast along L and concatenated as extra channels in
    front of every conv block."""
    h = x.unsqueeze(1) if x.ndim == 2 else x
    t = t.reshape(-1, 1).to(h.dtype)
    emb = _lin(p, "time_mlp.2", F.gelu(_lin(p, "time_mlp.0", t)))
    if premodule == "NormalizeLogRadius":
        h, lr = normalize_log_radius(h)
        h = h * math.sqrt(h.shape[-1])                                   # NNUnet1D.py:134
        lr = lr.reshape(lr.shape[0], -1)
        emb = emb + _lin(p, "scale_embed.2", F.gelu(_lin(p, "scale_embed.0", lr)))
    emb = emb.unsqueeze(-1)
    rep = lambda z: emb.expand(-1, -1, z.shape[-1])
    skips = []
    for i in range(levels):
        h = _conv_block_1d(p, f"enc_blocks.{i}", torch.cat([h, rep(h)], dim=1))
        skips.append(h)
        h = F.conv1d(h, p[f"downs.{i}.weight"], p[f"downs.{i}.bias"], stride=2, padding=1)
    h = _conv_block_1d(p, "middle", torch.cat([h, rep(h)], dim=1))
    for i in range(levels):
        h = F.conv_transpose1d(h, p[f"up_convs.{i}.weight"], p[f"up_convs.{i}.bias"], stride=2, padding=1)
        s = skips.pop()
        if h.shape[-1] != s.shape[-1]:
            h = F.pad(h, (0, s.shape[-1] - h.shape[-1]))                 # NNUnet1D.py:171-172
        h = _conv_block_1d(p, f"dec_blocks.{i}", torch.cat([h, s, rep(h)], dim=1))
    return F.conv1d(h, p["final.weight"], p["final.bias"]).squeeze(1)


# --------------------------------------------------------------------------
# 2-D U-Net                          model/unet.py:276-517, NNUnet.py:80-245
# --------------------------------------------------------------------------

@dataclass
class UNet2DConfig:
    """Topology knobs of UNetModel.__init__ (model/unet.py:300-446) at the
    values VorticityUNet passes (NNUnet.py:175-192)."""
    in_channels: int = 1
    out_channels: int = 1
    model_channels: int = 32
    channel_mult: Sequence[int] = (1, 2, 4)
    num_res_blocks: int = 2
    attention_resolutions: Sequence[int] = (2, 4)
    in_space: int = 16
    use_log_norm: bool = False


def _gn_silu(p: Params, key: str, h):
    """GroupNorm(min(C,32) groups, eps 1e-5, affine) then SiLU.
    model/nn_utils.py:107-114,39-46."""
    C = h.shape[1]
    return swish(F.group_norm(h, min(C, 32), p[key + ".weight"], p[key + ".bias"], eps=1e-5))


def resblock(p: Params, key: str, x, emb):
    """model/unet.py:182-195 (use_scale_shift_norm=False, dropout 0)."""
    h = F.conv2d(_gn_silu(p, key + ".in_layers.0", x),
                 p[key + ".in_layers.2.weight"], p[key + ".in_layers.2.bias"], padding=1)
    e = _lin(p, key + ".emb_layers.1", swish(emb))
    h = h + e[:, :, None, None]
    h = F.conv2d(_gn_silu(p, key + ".out_layers.0", h),
                 p[key + ".out_layers.3.weight"], p[key + ".out_layers.3.bias"], padding=1)
    if key + ".skip_connection.weight" in p:
        x = F.conv2d(x, p[key + ".skip_connection.weight"], p.get(key + ".skip_connection.bias"))
    return x + h


def attention_block(p: Params, key: str, x):
    """Single-head self-attention over H*W tokens; q and k are each scaled
    by ch^-1/4; residual.  model/unet.py:220-250."""
    b, c = x.shape[:2]
    xf = x.reshape(b, c, -1)
    C = xf.shape[1]
    hn = F.group_norm(xf, min(C, 32), p[key + ".norm.weight"], p[key + ".norm.bias"], eps=1e-5)
    qkv = F.conv1d(hn, p[key + ".qkv.weight"], p[key + ".qkv.bias"])
    q, k, v = torch.split(qkv, c, dim=1)
    s = 1.0 / math.sqrt(math.sqrt(c))
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v)
    a = F.conv1d(a, p[key + ".proj_out.weight"], p[key + ".proj_out.bias"])
    return (xf + a).reshape(x.shape)


def unet2d_plan(cfg: UNet2DConfig):
    """Replays the constructor loops of model/unet.py:349-446 and returns,
    for input / middle / output blocks, the list of layer kinds per block —
    the state_dict sub-index of a layer is its position in that list."""
    mc = cfg.model_channels
    inp = [["conv"]]
    ds = 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            blk = ["res"]
            if ds in cfg.attention_resolutions:
                blk.append("attn")
            inp.append(blk)
        if level != len(cfg.channel_mult) - 1:
            inp.append(["down"])
            ds *= 2
    out = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            blk = ["res"]
            if ds in cfg.attention_resolutions:
                blk.append("attn")
            if level and i == cfg.num_res_blocks:
                blk.append("up")
                ds //= 2
            out.append(blk)
    return inp, ["res", "attn", "res"], out


def _run_block(p: Params, key: str, kinds, h, emb):
    for j, kind in enumerate(kinds):
        k = f"{key}.{j}"
        if kind == "conv":
            h = F.conv2d(h, p[k + ".weight"], p[k + ".bias"], padding=1)
        elif kind == "res":
            h = resblock(p, k, h, emb)
        elif kind == "attn":
            h = attention_block(p, k, h)
        elif kind == "down":                                             # model/unet.py:106-108
            h = F.conv2d(h, p[k + ".op.weight"], p[k + ".op.bias"], stride=2, padding=1)
        elif kind == "up":                                               # model/unet.py:60-73
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            h = F.conv2d(h, p[k + ".conv.weight"], p[k + ".conv.bias"], padding=1)
    return h


def unet2d_core_forward(p: Params, x_img, t, cfg: UNet2DConfig, log_norm=None, prefix=""):
    """UNetModelWithLogNorm.forward.  NNUnet.py:96-142, model/unet.py:469-517."""
    q = {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)} if prefix else p
    mc = cfg.model_channels
    emb = _lin(q, "time_embed.2", swish(_lin(q, "time_embed.0", sinusoidal_embedding(t, mc))))
    if cfg.use_log_norm:
        le = sinusoidal_embedding(log_norm.reshape(-1), mc)
        emb = emb + _lin(q, "scale_embed.2", swish(_lin(q, "scale_embed.0", le)))
    inp, mid, out = unet2d_plan(cfg)
    hs = []
    h = x_img
    for i, kinds in enumerate(inp):
        h = _run_block(q, f"input_blocks.{i}", kinds, h, emb)
        hs.append(h)
    h = _run_block(q, "middle_block", mid, h, emb)
    for i, kinds in enumerate(out):
        h = _run_block(q, f"output_blocks.{i}", kinds, torch.cat([h, hs.pop()], dim=1), emb)
    h = _gn_silu(q, "out.0", h)
    return F.conv2d(h, q["out.2.weight"], q["out.2.bias"], padding=1)


IMAGE_SCALE = 5.0   # NNUnet.py:19


def flat_to_image(x, H, W, order="C", channels=1):
    """(B, C*H*W)/5 -> (B,C,H,W); 'F' order = view (.,W,H) then transpose.
    NNUnet.py:26-36 (the reference is C=1 only; C>1 is channel-major)."""
    B = x.shape[0]
    x = x / IMAGE_SCALE
    if order == "C":
        return x.reshape(B, channels, H, W)
    return x.reshape(B, channels, W, H).transpose(2, 3).contiguous()


def image_to_flat(y, order="C"):
    """Inverse of flat_to_image, times 5.  NNUnet.py:53-77."""
    B = y.shape[0]
    y = IMAGE_SCALE * y
    if order == "C":
        return y.reshape(B, -1)
    return y.transpose(2, 3).contiguous().reshape(B, -1)


def vorticity_unet_forward(p: Params, x, t, cfg: UNet2DConfig, premodule=None, order="C"):
    """VorticityUNet.forward on flat input.  NNUnet.py:195-245."""
    t = t.reshape(-1)
    log_norm = None
    if premodule == "NormalizeLogRadius":
        x, log_norm = normalize_log_radius(x)
        x = x * math.sqrt(x.shape[-1])                                   # NNUnet.py:205
    H = W = cfg.in_space
    img = flat_to_image(x, H, W, order, cfg.in_channels)
    out = unet2d_core_forward(p, img, t, cfg, log_norm=log_norm, prefix="core.")
    return image_to_flat(out, order)
