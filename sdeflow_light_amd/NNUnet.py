"""2-D U-Net score network on HIP kernels — host mirror of the reference's
``NNUnet.py`` + ``model/unet.py`` (``VorticityUNet`` / ``UNetModelWithLogNorm``):
same constructor signature and ``state_dict`` keys (``core.time_embed.*``,
``core.input_blocks.N.0.{in_layers,emb_layers,out_layers,skip_connection}.*``,
``…N.1.{norm,qkv,proj_out}.*``, ``…N.0.op.*``, ``core.middle_block.*``,
``core.output_blocks.N.K.*`` (``.conv`` for Upsample), ``core.out.{0,2}.*``;
model/unet.py:338-446).

The ``nn`` children only hold parameters (PyTorch layouts); the computation is a
hand-scheduled pipeline of the implicit-GEMM conv kernels, the dual GroupNorm+SiLU
kernels and the attention pieces (batched MFMA GEMM + dual softmax), channels-last,
with the forward-mode tangent as the second half of the batch and a hand-written
backward (no autograd tape, no double backward).

Extension over the reference wrapper: ``channels`` (default
1) lets the flat state be ``(B, channels*H*W)`` — needed for the 64x64x3 config;
the core UNet upstream already supports any ``in_channels``.
"""
from __future__ import annotations

import math
import os
from typing import Literal, Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import MsgmError
from .NN import FlatParamMixin
from .convnet import ConvOp, ConvOpSet

SILU = ops.ACT_SILU
scale_image = 5          # NNUnet.py:19


def flat_to_img(x: torch.Tensor, H: int, W: int, order: Literal["C", "F"] = "C") -> torch.Tensor:
    """x (B, H*W) / scale_image -> (B, 1, H, W), C or F (column-major) ordering (NNUnet.py:26-51) — one kernel."""
    B, d = x.shape
    assert d == H * W, f"Expected d={H * W}, got {d}"
    return ops.flat_to_image(x.contiguous().float(), B, 1, H, W, order == "F", 1.0 / scale_image).view(B, 1, H, W)


def img_to_flat(y: torch.Tensor, order: Literal["C", "F"] = "C") -> torch.Tensor:
    """y (B, 1, H, W) * scale_image -> (B, H*W) (NNUnet.py:53-77)."""
    B, C, H, W = y.shape
    assert C == 1, f"Expected 1 channel, got {C}"
    return ops.image_to_flat(y.contiguous().float().view(-1), B, 1, H, W, order == "F", float(scale_image))


def zero_module(m):
    for p in m.parameters():
        p.detach().zero_()
    return m


def _norm(c):            # normalization(): GroupNorm32(min(c,32), c)      model/nn_utils.py:107-114
    return nn.GroupNorm(min(c, 32), c)


class ResBlock(nn.Module):
    """Parameter holder with the reference's child names (model/unet.py:139-168)."""

    def __init__(self, channels, emb_channels, out_channels):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels
        self.in_layers = nn.Sequential(_norm(channels), nn.Identity(), nn.Conv2d(channels, out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.Identity(), nn.Linear(emb_channels, out_channels))
        self.out_layers = nn.Sequential(_norm(out_channels), nn.Identity(), nn.Identity(),
                                        zero_module(nn.Conv2d(out_channels, out_channels, 3, padding=1)))
        self.skip_connection = nn.Identity() if out_channels == channels else nn.Conv2d(channels, out_channels, 1)


class AttentionBlock(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.channels = channels
        self.norm = _norm(channels)
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.proj_out = zero_module(nn.Conv1d(channels, channels, 1))


class Downsample(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.op = nn.Conv2d(channels, channels, 3, stride=2, padding=1)


class Upsample(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)


class UNetModelWithLogNorm(nn.Module):
    """Topology of UNetModel.__init__ (model/unet.py:300-446) at the options the
    driver uses (dims=2, conv_resample, 1 head, no scale-shift norm, no classes)."""

    def __init__(self, in_channels, model_channels, out_channels, in_space, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False,
                 num_heads=1, num_heads_upsample=-1, use_scale_shift_norm=False, learn_potential=False, use_log_norm=False):
        super().__init__()
        if dims != 2 or not conv_resample or num_classes is not None or num_heads != 1 or use_scale_shift_norm or \
                learn_potential or dropout != 0:
            raise MsgmError("HIP U-Net is built for the driver's options (dims=2, conv_resample, 1 head, dropout 0)")
        self.use_log_norm = use_log_norm
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.channel_mult, self.num_res_blocks = tuple(channel_mult), num_res_blocks
        self.attention_resolutions = tuple(attention_resolutions)
        ted = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.Identity(), nn.Linear(ted, ted))
        if use_log_norm:               # mirrors the time MLP for log||x|| (NNUnet.py:88-94)
            self.scale_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.Identity(), nn.Linear(ted, ted))
        ch = model_channels * channel_mult[0]
        self.input_blocks = nn.ModuleList([nn.Sequential(nn.Conv2d(in_channels, ch, 3, padding=1))])
        chans, ds = [ch], 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [ResBlock(ch, ted, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(AttentionBlock(ch))
                self.input_blocks.append(nn.Sequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(nn.Sequential(Downsample(ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = nn.Sequential(ResBlock(ch, ted, ch), AttentionBlock(ch), ResBlock(ch, ted, ch))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [ResBlock(ch + chans.pop(), ted, model_channels * mult)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(AttentionBlock(ch))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch))
                    ds //= 2
                self.output_blocks.append(nn.Sequential(*layers))
        self.out = nn.Sequential(_norm(ch), nn.Identity(), zero_module(nn.Conv2d(model_channels * channel_mult[0], out_channels, 3, padding=1)))


# ----------------------------------------------------------------------------- executable ops
class _Res:
    def __init__(self, m: ResBlock):
        self.m = m
        ci, co = m.channels, m.out_channels
        self.ci, self.co = ci, co
        self.conv1 = ConvOp(m.in_layers[2].weight, m.in_layers[2].bias, "conv", (3, 3), 1, 1, [ci])
        # the embedding projection emb_layers[1] runs in the network's ops.EmbBank (one launch for all ResBlocks);
        # bank_i = this block's index in it
        self.bank_i = -1
        self.conv2 = ConvOp(m.out_layers[3].weight, m.out_layers[3].bias, "conv", (3, 3), 1, 1, [co])
        self.skip = None if isinstance(m.skip_connection, nn.Identity) else \
            ConvOp(m.skip_connection.weight, m.skip_connection.bias, "conv", (1, 1), 1, 0, [ci])
        self.ops = [self.conv1, self.conv2] + ([self.skip] if self.skip else [])
        self.conv1_2 = self.skip_2 = None      # two-source twins for decoder blocks (sampler path, see make_two_source)
        self.skip_2t = None                    # the skip twin of the TRAINING path (its own gradient image)
        self.split = None

    def make_two_source(self, C0: int, C1: int):
        """Decoder ResBlocks read cat([h, skip]) (model/unet.py:514).  On the sampler path the concatenation is never
        materialised: GroupNorm statistics, conv1 and the 1x1 skip conv read the two tensors directly.  The twins
        share the weight Parameters and only differ in the packed layout (each source padded to 16 channels)."""
        m = self.m
        if self.skip is None or C0 + C1 != self.ci:
            raise MsgmError("two-source twins are for decoder ResBlocks with a skip convolution")
        self.split = (C0, C1)
        self.conv1_2 = ConvOp(m.in_layers[2].weight, m.in_layers[2].bias, "conv", (3, 3), 1, 1, [C0, C1])
        self.skip_2 = ConvOp(m.skip_connection.weight, m.skip_connection.bias, "conv", (1, 1), 1, 0, [C0, C1])
        # training: GroupNorm reads the two tensors and writes ONE normalised tensor (conv1 stays single-source); only the
        # 1x1 skip conv needs a two-source twin, with its own gradient image (it REPLACES self.skip on that path)
        self.skip_2t = ConvOp(m.skip_connection.weight, m.skip_connection.bias, "conv", (1, 1), 1, 0, [C0, C1])
        return [self.conv1_2, self.skip_2]


class _Attn:
    def __init__(self, m: AttentionBlock):
        self.m, self.c = m, m.channels
        self.qkv = ConvOp(m.qkv.weight, m.qkv.bias, "conv", (1,), 1, 0, [m.channels])
        self.proj = ConvOp(m.proj_out.weight, m.proj_out.bias, "conv", (1,), 1, 0, [m.channels])
        self.ops = [self.qkv, self.proj]


class VorticityUNet(nn.Module, FlatParamMixin):
    def __init__(self, base_channels: int = 32, channel_mults=(1, 2, 4), num_res_blocks: int = 2, emb_dim_ignored: int = 128,
                 dropout: float = 0.0, premodule: Optional[str] = None, in_space: int = 16, attention_resolutions=(2, 4),
                 conv_resample: bool = True, num_heads: int = 1, use_checkpoint: bool = False, learn_potential: bool = False,
                 flatten_order: Literal["C", "F"] = "C", channels: int = 1):
        super().__init__()
        assert premodule in (None, "NormalizeLogRadius")
        self.pre = premodule == "NormalizeLogRadius" or None
        self.in_space = int(in_space)
        assert flatten_order in ("C", "F")
        self.flatten_order = flatten_order
        self.channels = channels
        self.core = UNetModelWithLogNorm(in_channels=channels, model_channels=base_channels, out_channels=channels,
                                         in_space=in_space, num_res_blocks=num_res_blocks,
                                         attention_resolutions=attention_resolutions, dropout=dropout,
                                         channel_mult=tuple(channel_mults), conv_resample=conv_resample, dims=2,
                                         num_classes=None, use_checkpoint=use_checkpoint, num_heads=num_heads,
                                         use_scale_shift_norm=False, learn_potential=learn_potential,
                                         use_log_norm=(premodule == "NormalizeLogRadius"))
        self._x = None
        self._flat = None

    # ------------------------------------------------------------------ build
    def _build(self):
        self.flat_parameters()
        core = self.core
        first = core.time_embed[0].weight
        if self._x is not None and self._x["t0"].weight is first and self._x["t0"].Wp.device == first.device:
            return self._x
        mc = core.model_channels
        x = {"t0": ConvOp(core.time_embed[0].weight, core.time_embed[0].bias, "linear", (1,), 1, 0, [mc]),
             "t2": ConvOp(core.time_embed[2].weight, core.time_embed[2].bias, "linear", (1,), 1, 0, [4 * mc])}
        if core.use_log_norm:
            x["s0"] = ConvOp(core.scale_embed[0].weight, core.scale_embed[0].bias, "linear", (1,), 1, 0, [mc])
            x["s2"] = ConvOp(core.scale_embed[2].weight, core.scale_embed[2].bias, "linear", (1,), 1, 0, [4 * mc])

        def wrap(seq):
            out = []
            for layer in seq:
                if isinstance(layer, ResBlock):
                    out.append(("res", _Res(layer)))
                elif isinstance(layer, AttentionBlock):
                    out.append(("attn", _Attn(layer)))
                elif isinstance(layer, Downsample):
                    c = layer.op.in_channels
                    out.append(("down", ConvOp(layer.op.weight, layer.op.bias, "conv", (3, 3), 2, 1, [c])))
                elif isinstance(layer, Upsample):
                    c = layer.conv.in_channels
                    out.append(("up", ConvOp(layer.conv.weight, layer.conv.bias, "conv", (3, 3), 1, 1, [c], ups=True)))
                elif isinstance(layer, nn.Conv2d):
                    out.append(("conv", ConvOp(layer.weight, layer.bias, "conv", (3, 3), 1, 1, [layer.in_channels])))
                else:
                    raise MsgmError(f"unexpected layer {type(layer)}")
            return out
        x["in"] = [wrap(b) for b in core.input_blocks]
        x["mid"] = wrap(core.middle_block)
        x["outb"] = [wrap(b) for b in core.output_blocks]
        x["fin"] = ConvOp(core.out[2].weight, core.out[2].bias, "conv", (3, 3), 1, 1, [core.out[2].in_channels])
        allops = [x["t0"], x["t2"], x["fin"]] + ([x["s0"], x["s2"]] if core.use_log_norm else [])
        for blk in x["in"] + [x["mid"]] + x["outb"]:
            for kind, o in blk:
                allops += o.ops if kind in ("res", "attn") else [o]
        x["all"] = allops
        x["set"] = ConvOpSet(allops)
        res = [o for blk in x["in"] + [x["mid"]] + x["outb"] for kind, o in blk if kind == "res"]
        for i, r in enumerate(res):
            r.bank_i = i
        x["bank"] = ops.EmbBank([(r.m.emb_layers[1].weight, r.m.emb_layers[1].bias, r.m.in_layers[2].bias) for r in res], 4 * mc)
        # channel count after every input block = the skip tensors the decoder pops (model/unet.py:506-514)
        chans = []
        for blk in x["in"]:
            kind, o = blk[-1] if blk[-1][0] != "attn" else blk[-2]
            chans.append(o.co if kind == "res" else o.Cout)
        twins = []
        for blk in x["outb"]:
            Cs = chans.pop()
            kind, res = blk[0]
            if kind == "res" and res.skip is not None and res.ci > Cs and (res.ci - Cs) % 16 == 0 and Cs % 16 == 0:
                twins += res.make_two_source(res.ci - Cs, Cs)
        x["set2"] = ConvOpSet(twins) if twins else None
        tw_t = [o.skip_2t for blk in x["outb"] for kind, o in blk if kind == "res" and o.skip_2t is not None]
        x["set2t"] = ConvOpSet(tw_t) if tw_t else None
        self._x = x
        return x

    # ------------------------------------------------------------------ forward pieces
    def _gn(self, gnm: nn.GroupNorm, h, Bp, P, C, dual, silu, tape):
        G = gnm.num_groups
        stats = torch.empty(Bp * G * 4, device=h.device) if tape is not None else None
        out = ops.groupnorm_dual_forward(h, gnm.weight.detach(), gnm.bias.detach(), Bp, P, C, G, dual, silu, stats=stats)
        return out, stats

    def _gn_fold(self, gnm: nn.GroupNorm, x, Bp, P, C, x1=None, C1=0):
        """GroupNorm as a per-(sample, channel) affine map for a conv that applies it while staging its input.  When the
        convolution(s) that produced x (and x1, the concatenated second source) left their per-channel sums on the tensor
        (ConvOp.forward(stats=True)), the statistics come from those few KB instead of a pass over the tensor."""
        cs = getattr(x, "_msgm_cs", None)
        cs1 = getattr(x1, "_msgm_cs", None) if x1 is not None else None
        if cs is not None and (x1 is None or cs1 is not None) and self._cs_on:
            return ops.groupnorm_affine_cs(cs[0], cs[1], C, gnm.weight.detach(), gnm.bias.detach(), Bp, P, gnm.num_groups,
                                           cs1=cs1[0] if cs1 is not None else None, S1=cs1[1] if cs1 is not None else 0, C1=C1)
        return ops.groupnorm_affine(x, C, gnm.weight.detach(), gnm.bias.detach(), Bp, P, gnm.num_groups, x1=x1, C1=C1)

    def _res_fwd(self, r: _Res, x, N, Bp, H, W, semb, dual, tape, er):
        P = H * W
        eo = self._eo[r.bank_i]                                  # emb_layers projection, er rows (N with log-radius conditioning)
        if isinstance(x, tuple) and tape is not None:            # training, decoder block: cat([h, skip]) never materialised
            h0, s0 = x
            C0, C1 = r.split
            gnm = r.m.in_layers[0]
            st1 = torch.empty(Bp * gnm.num_groups * 4, device=h0.device)
            h1 = ops.groupnorm_dual_forward2(h0, C0, s0, C1, gnm.weight.detach(), gnm.bias.detach(), Bp, P, gnm.num_groups, dual, True,
                                             stats=st1)
            h2, _, _ = r.conv1.forward([h1], N, H, W, Bp, samp_bias=eo, emb_rows=er)
            h3, st2 = self._gn(r.m.out_layers[0], h2, Bp, P, r.co, dual, True, tape)
            out, _, _ = r.skip_2t.forward([h0, s0], N, H, W, Bp)
            r.conv2.forward([h3], N, H, W, Bp, out=out, accumulate=True)
            tape.append(("res2", r, h0, s0, H, W, h1, st1, h2, st2, h3))
            return out
        if isinstance(x, tuple):                                 # (h, skip): decoder block without the concatenation
            h0, s0 = x
            C0, C1 = r.split
            gnm = r.m.in_layers[0]
            aff = self._gn_fold(gnm, h0, Bp, P, C0, x1=s0, C1=C1)
            wn = getattr(self, "_wino", False)
            h2, _, _ = r.conv1_2.forward([h0, s0], N, H, W, Bp, samp_bias=eo, emb_rows=er, in_affine=aff, in_act=1, wino=wn,
                                         stats=True)
            out, _, _ = r.skip_2.forward([h0, s0], N, H, W, Bp)
            if r.conv2.can_transform_input(N, H, W):
                r.conv2.forward([h2], N, H, W, Bp, out=out, accumulate=True,
                                in_affine=self._gn_fold(r.m.out_layers[0], h2, Bp, P, r.co), in_act=1, wino=wn, stats=True)
            else:
                h3, _ = self._gn(r.m.out_layers[0], h2, Bp, P, r.co, False, True, None)
                r.conv2.forward([h3], N, H, W, Bp, out=out, accumulate=True, wino=wn, stats=True)
            return out
        fold = (not dual and tape is None and not os.environ.get("MSGM_NO_GN_FOLD")
                and r.conv1.can_transform_input(N, H, W) and r.conv2.can_transform_input(N, H, W))
        if fold:
            # sampler path: GroupNorm+SiLU are applied by the consuming conv while it stages its input tile, the residual
            # is added in conv2's epilogue — the two normalised tensors and the separate add pass never exist
            wn = getattr(self, "_wino", False)
            h2, _, _ = r.conv1.forward([x], N, H, W, Bp, samp_bias=eo, emb_rows=er,
                                       in_affine=self._gn_fold(r.m.in_layers[0], x, Bp, P, r.ci), in_act=1, wino=wn, stats=True)
            aff2 = self._gn_fold(r.m.out_layers[0], h2, Bp, P, r.co)
            if r.skip is not None:
                out, _, _ = r.skip.forward([x], N, H, W, Bp)
                r.conv2.forward([h2], N, H, W, Bp, out=out, accumulate=True, in_affine=aff2, in_act=1, wino=wn, stats=True)
            else:
                out, _, _ = r.conv2.forward([h2], N, H, W, Bp, residual=x, in_affine=aff2, in_act=1, wino=wn, stats=True)
            return out
        h1, st1 = self._gn(r.m.in_layers[0], x, Bp, P, r.ci, dual, True, tape)
        h2, _, _ = r.conv1.forward([h1], N, H, W, Bp, samp_bias=eo, emb_rows=er)
        h3, st2 = self._gn(r.m.out_layers[0], h2, Bp, P, r.co, dual, True, tape)
        if r.skip is not None:
            out, _, _ = r.skip.forward([x], N, H, W, Bp)
            r.conv2.forward([h3], N, H, W, Bp, out=out, accumulate=True)
        else:
            out, _, _ = r.conv2.forward([h3], N, H, W, Bp, residual=x)          # h + x in the epilogue (unet.py:187)
        if tape is not None:
            tape.append(("res", r, x, H, W, h1, st1, h2, st2, h3))
        return out

    def _attn_fwd(self, a: _Attn, x, N, Bp, H, W, dual, tape):
        T, C = H * W, a.c
        dev = x.device
        s2 = 1.0 / math.sqrt(C)                                          # (ch^-1/4)^2           model/unet.py:245-248
        if not dual and tape is None and ops.attention_supported(T, C):  # sampler: nothing to keep, no tangent
            if a.qkv.can_transform_input(N, 1, T) and not os.environ.get("MSGM_NO_GN_FOLD"):
                qkv, _, _ = a.qkv.forward([x], N, 1, T, Bp, in_affine=self._gn_fold(a.m.norm, x, Bp, T, C))   # GN folded in
            else:
                hn, _ = self._gn(a.m.norm, x, Bp, T, C, False, False, None)
                qkv, _, _ = a.qkv.forward([hn], N, 1, T, Bp)
            att = ops.attention_forward(qkv, torch.empty(N * T * C, device=dev), N, T, C, s2)
            out, _, _ = a.proj.forward([att], N, 1, T, Bp, residual=x, stats=True)    # x + proj(.) in the epilogue (unet.py:232)
            return out
        hn, st = self._gn(a.m.norm, x, Bp, T, C, dual, False, tape)
        qkv, _, _ = a.qkv.forward([hn], N, 1, T, Bp)                     # [N][T][3C]: q | k | v channel slices
        if dual and ops.attention_dual_supported(T, C) and not os.environ.get("MSGM_NO_ATTN_DUAL"):
            # training: ONE kernel for the six products of the dual forward, nothing of size (T,T) written; the
            # backward recomputes the logits from q, k and the per-query (log-sum-exp, rbar) kept here
            att, stats = ops.attention_dual_forward(qkv, Bp, T, C, s2)
            out, _, _ = a.proj.forward([att], N, 1, T, Bp, residual=x)
            if tape is not None:
                tape.append(("attn", a, x, H, W, hn, st, qkv, None, None, stats, att))
            return out
        ld = 3 * C
        half = Bp * T * ld                                               # offset of the tangent rows
        S = torch.empty(Bp * T * T, device=dev)
        sq, sk, sS = (T * ld, ld, 1), (T * ld, 1, ld), (T * T, T, 1)
        ops.bmm(qkv, 0, qkv, C, S, 0, T, T, C, Bp, sq, sk, sS, alpha=s2)
        Wd = Pd = None
        if dual:
            Wd, Pd = torch.empty_like(S), torch.empty_like(S)
            ops.bmm(qkv, half, qkv, C, Wd, 0, T, T, C, Bp, sq, sk, sS, alpha=s2,                     # qdot k^T + q kdot^T
                    pair2=(qkv, 0, qkv, half + C))
        ops.softmax_dual_forward(S, T, Wd, Pd)                           # S <- P
        att = torch.empty(N * T * C, device=dev)
        sP, sv, sa = (T * T, T, 1), (T * ld, ld, 1), (T * C, C, 1)
        if dual:
            # adot = P vdot + Pdot v and a = P v in ONE pass over P (msgm_bmm_dual)
            offa = Bp * T * C
            ops.bmm(S, 0, qkv, half + 2 * C, att, offa, T, C, T, Bp, sP, sv, sa, pair2=(Pd, 0, qkv, 2 * C),
                    third=(qkv, 2 * C, att, 0))
        else:
            ops.bmm(S, 0, qkv, 2 * C, att, 0, T, C, T, Bp, sP, sv, sa)                               # a = P v
        out, _, _ = a.proj.forward([att], N, 1, T, Bp, residual=x)
        if tape is not None:
            tape.append(("attn", a, x, H, W, hn, st, qkv, S, Wd, Pd, att))
        return out

    def _run(self, img, t, N, Bp, dual, tape, logr=None):
        """img: channels-last [N][H][W][Cin].  Returns channels-last [N][H][W][Cout].
        logr: [log r ; rdot/r] (N,) with NormalizeLogRadius conditioning."""
        x = self._build()
        x["set"].pack()
        # sampler path (no tangent, nothing kept): the 3x3 stride-1 convolutions on 16-multiple images take the Winograd
        # F(2x2,3x3) forward kernel — 2.25x fewer MFMAs, all fp32, 1.13-1.26x the direct kernel with the folded GroupNorm +
        # SiLU staging (tools/bench_wino.py), same fused options and statistics by-product; it differs from the direct form
        # by the rounding of its transforms (2-6e-7 per conv) and every sampler parity test runs through it.
        # MSGM_NO_WINO=1 keeps the direct kernels (A/B).  (Training: see _twino below.)
        self._wino = not dual and tape is None and not os.environ.get("MSGM_NO_WINO")
        self._cs_on = not os.environ.get("MSGM_NO_CHANSTATS")            # diagnostic A/B: GroupNorm statistics by a pass over the tensor
        # r3: the TRAINING pass's 3x3 stride-1 convolutions — forward and dgrad, primal and tangent rows alike — take the
        # Winograd kernel too (the dgrad as a Winograd forward of the cotangent with the flipped, transposed kernels): C4 step
        # 122.5 -> 116.9 ms at B = 256, 19.9 -> 18.6 ms at the 32-row shard.  fp32 throughout; its transforms round about twice
        # as much as the direct kernel (per conv 4-8e-7 against 3-4e-7 vs float64; on the ill-conditioned det_params benchmark
        # the per-sample loss is 2.2x the fp32 oracle's own distance from float64 instead of 1.07x — tests/test_round2_gpu.py;
        # the well-conditioned reference fixture g17 holds its absolute tolerances).  MSGM_TRAIN_WINO=0 keeps the direct kernels.
        self._twino = tape is not None and os.environ.get("MSGM_TRAIN_WINO", "1") != "0" and not os.environ.get("MSGM_NO_WINO")
        if self._twino:
            x["set"].pack_wino(train=True)
        elif self._wino:
            x["set"].pack_wino()
        else:
            x["set"].clear_train_wino()
        # opt-in experiment (never the default): the sampler's 3x3 convolutions with 32-multiple channel counts
        # in bf16-split arithmetic (six bf16 MFMA products per fp32 product, fp32 accumulate) instead of Winograd
        self._b6 = self._wino and bool(os.environ.get("MSGM_SAMPLER_BF16X3"))
        x["set"].pack_b6(self._b6)
        core = self.core
        mc = core.model_channels
        H = W = self.in_space
        e0 = ops.timestep_embedding(t, mc)
        z1, _, _ = x["t0"].forward([e0.view(-1)], Bp, 1, 1, Bp)
        a1 = ops.act_dual_forward(SILU, z1, torch.empty_like(z1), False)
        emb, _, _ = x["t2"].forward([a1], Bp, 1, 1, Bp)
        er, pre = Bp, None
        if core.use_log_norm:
            # emb += scale_embed(timestep_embedding(log r)) — it has a tangent (rows Bp..N-1)        NNUnet.py:101-105
            le = ops.timestep_embedding_dual(logr, Bp, mc) if dual else ops.timestep_embedding(logr, mc)
            zs, _, _ = x["s0"].forward([le.view(-1)], N, 1, 1, Bp)
            as_ = ops.act_dual_forward(SILU, zs, torch.empty_like(zs), dual)
            es, _, _ = x["s2"].forward([as_], N, 1, 1, Bp)
            pv = es[: emb.numel()]
            ops.lincomb(pv, pv, 1.0, emb, 1.0)
            pre = (le, zs, as_)
            emb, er = es, N
        semb = ops.act_dual_forward(SILU, emb, torch.empty_like(emb), dual and er == N and N != Bp)   # emb_layers[0] = SiLU
        self._eo = x["bank"].forward(semb, er, Bp)               # every ResBlock's emb_layers[1] in ONE launch
        if tape is not None:
            tape.append(("emb", e0, z1, a1, emb, semb, pre, er))

        def run_block(blk, h, C, H, W):
            for kind, o in blk:
                if kind == "conv":
                    if tape is not None:
                        tape.append(("conv", o, h, H, W))
                    h, H, W = o.forward([h], N, H, W, Bp, stats=not dual and tape is None)
                    C = o.Cout
                elif kind == "res":
                    h = self._res_fwd(o, h, N, Bp, H, W, semb, dual, tape, er)
                    C = o.co
                elif kind == "attn":
                    h = self._attn_fwd(o, h, N, Bp, H, W, dual, tape)
                elif kind in ("down", "up"):
                    if tape is not None:
                        tape.append((kind, o, h, H, W))
                    h, H, W = o.forward([h], N, H, W, Bp, wino=getattr(self, "_wino", False), stats=not dual and tape is None)
            return h, C, H, W

        h, C = img, core.in_channels
        hs = []
        for blk in x["in"]:
            h, C, H, W = run_block(blk, h, C, H, W)
            hs.append((h, C))
            if tape is not None:
                tape.append(("save_skip",))
        h, C, H, W = run_block(x["mid"], h, C, H, W)
        nocat = (not dual and tape is None and x["set2"] is not None and not os.environ.get("MSGM_NO_GN_FOLD"))
        nocat_t = tape is not None and dual and x["set2t"] is not None and not os.environ.get("MSGM_TRAIN_CAT")
        if nocat:
            x["set2"].pack()
            if self._wino:
                x["set2"].pack_wino()
            x["set2"].pack_b6(self._b6)
        if nocat_t:
            x["set2t"].pack()
            if self._twino:
                x["set2t"].pack_wino(train=True)
            else:
                x["set2t"].clear_train_wino()
        for blk in x["outb"]:
            s, Cs = hs.pop()
            r0 = blk[0][1]
            if nocat_t and blk[0][0] == "res" and r0.split == (C, Cs) and r0.skip_2t is not None:
                # training path: GroupNorm, the skip conv and their backward read h and the skip tensor as two sources
                h = self._res_fwd(r0, (h, s), N, Bp, H, W, semb, dual, tape, er)
                h, C, H, W = run_block(blk[1:], h, r0.co, H, W)
                continue
            if nocat and blk[0][0] == "res" and r0.split == (C, Cs) and r0.conv1_2.can_transform_input(N, H, W):
                # sampler path: the decoder ResBlock reads h and the skip tensor as two sources — no concatenation
                h = self._res_fwd(r0, (h, s), N, Bp, H, W, semb, dual, tape, er)
                h, C, H, W = run_block(blk[1:], h, r0.co, H, W)
                continue
            cat = torch.cat([h.view(N, H * W, C), s.view(N, H * W, Cs)], dim=2).reshape(-1)    # model/unet.py:514
            if tape is not None:
                tape.append(("cat", C, Cs, H, W))
            h, C, H, W = run_block(blk, cat, C + Cs, H, W)
        if not dual and tape is None and x["fin"].can_transform_input(N, H, W) and not os.environ.get("MSGM_NO_GN_FOLD"):
            # sampler: the output GroupNorm + SiLU (model/unet.py:442-444) applied by the output conv while it stages its input
            out, _, _ = x["fin"].forward([h], N, H, W, Bp, in_affine=self._gn_fold(core.out[0], h, Bp, H * W, C), in_act=1)
            return out
        hf, stf = self._gn(core.out[0], h, Bp, H * W, C, dual, True, tape)
        if tape is not None:
            tape.append(("fin", h, stf, hf, H, W, C))
        out, _, _ = x["fin"].forward([hf], N, H, W, Bp)
        return out

    @torch.no_grad()
    def forward(self, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        """x: (B, channels*H*W) flat or (B,channels,H,W); t: (B,) or (B,1) (NNUnet.py:195-245)."""
        t = t.reshape(-1).float().contiguous()
        S_, Cc = self.in_space, self.channels
        need_flat = x.dim() == 2
        if need_flat:
            B, d = x.shape
            assert d == Cc * S_ * S_, f"Flat dim {d} != {Cc}*{S_}*{S_}"
            flat, forder, sc_in, sc_out = x.contiguous().float(), self.flatten_order == "F", 1.0 / scale_image, float(scale_image)
        elif x.dim() == 4:
            B = x.shape[0]
            assert x.size(1) == Cc
            flat, forder, sc_in, sc_out = x.contiguous().float().reshape(B, -1), False, 1.0, 1.0
        else:
            raise ValueError(f"Unexpected input shape {tuple(x.shape)}")
        if t.numel() == 1 and B != 1:
            t = t.expand(B).contiguous()
        logr = None
        if self.pre:
            d = flat.shape[1]
            flat, logr = ops.normalize_dual(flat, B, d, False, float(d) ** 0.5)      # NNUnet.py:203-205
        img = ops.flat_to_image(flat, B, Cc, S_, S_, forder, sc_in)
        out = self._run(img, t, B, B, False, None, logr=logr)
        y = ops.image_to_flat(out, B, Cc, S_, S_, forder, sc_out)
        return y if need_flat else y.view(B, Cc, S_, S_)

    # ------------------------------------------------------------------ training
    @torch.no_grad()
    def ssm_grad(self, y, t, v, u, cst, inv_batch):
        """Per-sample SSM loss (B,) in the general form loss_b = adot.u + cst + |a|^2/2 (u, cst from
        ``msgm_ssm_terms``: any SDE family); gradients of sum_b loss_b*inv_batch into .grad."""
        B, d = y.shape
        N = 2 * B
        S_, Cc = self.in_space, self.channels
        self.flat_parameters()
        for p in self.parameters():
            if p.grad is None:
                self._flatten_parameters()
                break
        x = self._build()
        x["set"].zero_grad_images(bias_grads_zeroed=True)
        if x["set2t"] is not None:
            x["set2t"].zero_grad_images(bias_grads_zeroed=True)
        flat, gflat = self.flat_parameters()
        gflat.zero_()                    # GroupNorm parameter gradients are accumulated with atomics; one memset for all
        forder = self.flatten_order == "F"
        stacked = torch.cat([y.contiguous().float(), v.contiguous().float()], 0)
        logr = None
        if self.pre:
            stacked, logr = ops.normalize_dual(stacked, B, d, True, float(d) ** 0.5)
        img = ops.flat_to_image(stacked, N, Cc, S_, S_, forder, 1.0 / scale_image)
        tape = []
        tt = t.reshape(-1).contiguous().float()
        out = self._run(img, tt, N, B, True, tape, logr=logr)
        a_flat = ops.image_to_flat(out, N, Cc, S_, S_, forder, float(scale_image))      # [2B][d]: a | adot
        per, g = ops.ssm_loss(a_flat.view(-1), u, cst, inv_batch)
        gimg = ops.flat_to_image(g.view(N, d), N, Cc, S_, S_, forder, float(scale_image))   # adjoint of (x5, unflatten)
        with ops.DeferredReduces.on(y.device):           # the ~140 slot reductions of the weight / bias gradients: one launch
            self._backward(tape, gimg, N, B)
        x["set"].unpack_grads()
        if x["set2t"] is not None and any(rec[0] == "res2" for rec in tape):
            x["set2t"].unpack_grads()            # after "set": the twins' images replace the (unused, zero) single-source ones
        return per

    def _groupnorms(self):
        return [m for m in self.modules() if isinstance(m, nn.GroupNorm)]

    def _gn_bwd(self, gnm, xin, stats, g, Bp, P, C, silu, residual=None, residual2=None):
        return ops.groupnorm_dual_backward(xin, gnm.weight.detach(), gnm.bias.detach(), stats, g, gnm.weight.grad, gnm.bias.grad,
                                           Bp, P, C, gnm.num_groups, silu, residual=residual, residual2=residual2)

    def _backward(self, tape, g, N, Bp):
        x = self._x
        dev = g.device
        emb_rec = tape[0]
        _, e0, z1, a1, emb, semb, pre, er = emb_rec
        dsemb = torch.empty_like(semb)       # written (not accumulated) by the embedding bank's backward after the loop
        deo_all = x["bank"].dout
        dh = g
        pend = []                       # gradients w.r.t. the skip (hs) tensors, in pop order
        i = len(tape) - 1
        while i >= 1:
            r = tape[i]
            kind = r[0]
            if kind == "fin":
                _, h, stf, hf, H, W, C = r
                (dhf,) = x["fin"].backward(dh, [hf], N, H, W, Bp)
                dh = self._gn_bwd(self.core.out[0], h, stf, dhf, Bp, H * W, C, True)
            elif kind == "res":
                _, rb, xin, H, W, h1, st1, h2, st2, h3 = r
                P = H * W
                (dh3,) = rb.conv2.backward(dh, [h3], N, H, W, Bp)
                dh2 = self._gn_bwd(rb.m.out_layers[0], h2, st2, dh3, Bp, P, rb.co, True)
                (dh1,) = rb.conv1.backward(dh2, [h1], N, H, W, Bp, dsamp_bias=deo_all[rb.bank_i], emb_rows=er,
                                           bias_grad_elsewhere=True)
                # identity skip: the `h + x` cotangent is added in the GroupNorm apply pass (no separate axpy); so is the cotangent
                # that reaches this block's INPUT through the skip stack, when the tensor was saved for the decoder (the
                # "save_skip" record in front of this block: it was a separate `dh += skip` pass, nine per step)
                sk = None
                if i - 1 >= 1 and tape[i - 1][0] == "save_skip" and pend and not os.environ.get("MSGM_NO_SKIP_FOLD"):
                    sk = pend.pop()
                    i -= 1                                          # that record is consumed here
                dx = self._gn_bwd(rb.m.in_layers[0], xin, st1, dh1, Bp, P, rb.ci, True, residual=None if rb.skip is not None else dh,
                                  residual2=sk)
                if rb.skip is not None:
                    rb.skip.backward(dh, [xin], N, H, W, Bp, dsrc=[dx], dacc=[True])
                dh = dx
            elif kind == "res2":                                   # decoder ResBlock on (h, skip) without the concatenation
                _, rb, h0, s0, H, W, h1, st1, h2, st2, h3 = r
                P = H * W
                C0, C1 = rb.split
                (dh3,) = rb.conv2.backward(dh, [h3], N, H, W, Bp)
                dh2 = self._gn_bwd(rb.m.out_layers[0], h2, st2, dh3, Bp, P, rb.co, True)
                (dh1,) = rb.conv1.backward(dh2, [h1], N, H, W, Bp, dsamp_bias=deo_all[rb.bank_i], emb_rows=er,
                                           bias_grad_elsewhere=True)
                gnm = rb.m.in_layers[0]
                dx0, dx1 = ops.groupnorm_dual_backward2(h0, C0, s0, C1, gnm.weight.detach(), gnm.bias.detach(), st1, dh1,
                                                        gnm.weight.grad, gnm.bias.grad, Bp, P, gnm.num_groups, True)
                rb.skip_2t.backward(dh, [h0, s0], N, H, W, Bp, dsrc=[dx0, dx1], dacc=[True, True])
                pend.append(dx1)                                    # cotangent of the skip tensor, consumed by the encoder
                dh = dx0
            elif kind == "attn":
                dh = self._attn_bwd(r, dh, N, Bp)
            elif kind == "conv":
                _, o, hin, H, W = r
                o.backward(dh, [hin], N, H, W, Bp, need=[False])
                dh = None
            elif kind == "down":
                _, o, hin, H, W = r
                if i - 1 >= 1 and tape[i - 1][0] == "save_skip" and pend and not os.environ.get("MSGM_NO_SKIP_FOLD"):
                    sk = pend.pop()                                 # the dgrad accumulates onto the skip-stack cotangent
                    i -= 1
                    (dh,) = o.backward(dh, [hin], N, H, W, Bp, dsrc=[sk], dacc=[True])
                else:
                    (dh,) = o.backward(dh, [hin], N, H, W, Bp)
            elif kind == "up":
                _, o, hin, H, W = r
                dh = o.backward_ups(dh, hin, N, H, W, Bp)
            elif kind == "cat":
                _, C, Cs, H, W = r
                gg = dh.view(N, H * W, C + Cs)
                pend.append(gg[:, :, C:].contiguous().view(-1))
                dh = gg[:, :, :C].contiguous().view(-1)
            elif kind == "save_skip":
                # this tensor also fed an output block through the skip stack (model/unet.py:514)
                sk = pend.pop()
                if dh is None:
                    dh = sk
                else:
                    ops.lincomb(dh, dh, 1.0, sk, 1.0)
            i -= 1
        # emb_layers[1] of every ResBlock: weight / bias gradients (+ the conv1 biases, same gradient) and dsemb, 2 launches
        x["bank"].backward(semb, dsemb, er, Bp)
        # embedding MLPs: Linear -> SiLU -> Linear -> SiLU (shared emb_layers[0])
        if pre is not None:
            le, zs, as_ = pre
            ops.act_dual_backward(SILU, emb, dsemb)                       # (primal | tangent) rows
            ge = dsemb
            (das,) = x["s2"].backward(ge, [as_], N, 1, 1, Bp)
            ops.act_dual_backward(SILU, zs, das)
            x["s0"].backward(das, [le.view(-1)], N, 1, 1, Bp, need=[False])
        else:
            z = torch.zeros_like(emb)
            ge = torch.cat([dsemb, z])
            ops.act_dual_backward(SILU, torch.cat([emb, z]), ge)
        (da1,) = x["t2"].backward(ge[: z1.numel()].contiguous(), [a1], Bp, 1, 1, Bp)
        z1z = torch.zeros_like(z1)
        g1 = torch.cat([da1, z1z])
        ops.act_dual_backward(SILU, torch.cat([z1, z1z]), g1)
        x["t0"].backward(g1[: z1.numel()].contiguous(), [e0.view(-1)], Bp, 1, 1, Bp, need=[False])
        return dh

    def _attn_bwd(self, r, dout, N, Bp):
        _, a, xin, H, W, hn, st, qkv, Pm, Wd, Pd, att = r
        T, C = H * W, a.c
        dev = dout.device
        s2 = 1.0 / math.sqrt(C)
        ld = 3 * C
        half, offa = Bp * T * ld, Bp * T * C
        (datt,) = a.proj.backward(dout, [att], N, 1, T, Bp)             # [N][T][C]: abar | adotbar
        if Pm is None:                                                  # fused dual attention (Pd slot = its row stats)
            dqkv = ops.attention_dual_backward(qkv, att, datt, Pd, Bp, T, C, s2)
            (dhn,) = a.qkv.backward(dqkv, [hn], N, 1, T, Bp)
            return self._gn_bwd(a.m.norm, xin, st, dhn, Bp, T, C, False, residual=dout)     # x + proj(.): skip cotangent fused
        dqkv = torch.empty(N * T * ld, device=dev)                     # every slice is written exactly once below
        sP, sPt = (T * T, T, 1), (T * T, 1, T)                          # P(t,s) / P^T(s,t)
        sa, sq = (T * C, C, 1), (T * ld, ld, 1)
        # vbar = P^T abar + Pdot^T adotbar ; vdotbar = P^T adotbar
        ops.bmm(Pm, 0, datt, 0, dqkv, 2 * C, T, C, T, Bp, sPt, sa, sq, pair2=(Pd, 0, datt, offa),
                third=(datt, offa, dqkv, half + 2 * C))                  # both from one pass over P^T
        # Pbar = abar v^T + adotbar vdot^T ; Pdotbar = adotbar v^T
        Pb, Pdb = torch.empty_like(Pm), torch.empty_like(Pm)
        svT = (T * ld, 1, ld)                                            # B(k=c, j=s) = v[s][c]
        ops.bmm(datt, 0, qkv, 2 * C, Pb, 0, T, T, C, Bp, sa, svT, sP, pair2=(datt, offa, qkv, half + 2 * C))
        ops.bmm(datt, offa, qkv, 2 * C, Pdb, 0, T, T, C, Bp, sa, svT, sP)
        ops.softmax_dual_backward(Pm, Wd, Pb, Pdb, T)                   # Pb <- Wbar, Pdb <- Wdotbar
        # qbar = s2 (Wbar k + Wdotbar kdot) ; qdotbar = s2 Wdotbar k
        ops.bmm(Pdb, 0, qkv, half + C, dqkv, 0, T, C, T, Bp, sP, sq, sq, alpha=s2, pair2=(Pb, 0, qkv, C),
                third=(qkv, C, dqkv, half))                              # Wdotbar streamed once for both
        # kbar = s2 (Wbar^T q + Wdotbar^T qdot) ; kdotbar = s2 Wdotbar^T q
        ops.bmm(Pdb, 0, qkv, half, dqkv, C, T, C, T, Bp, sPt, sq, sq, alpha=s2, pair2=(Pb, 0, qkv, 0),
                third=(qkv, 0, dqkv, half + C))
        (dhn,) = a.qkv.backward(dqkv, [hn], N, 1, T, Bp)
        return self._gn_bwd(a.m.norm, xin, st, dhn, Bp, T, C, False, residual=dout)
