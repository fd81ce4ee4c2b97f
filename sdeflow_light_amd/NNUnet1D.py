"""1-D U-Net score network on the implicit-GEMM HIP kernels — host mirror of
the reference's ``NNUnet1D.py`` (same class names, constructor signature and
``state_dict`` keys: ``time_mlp.{0,2}``, ``enc_blocks.N.net.{0,2}``, ``downs.N``,
``middle.net.{0,2}``, ``up_convs.N``, ``dec_blocks.N.net.{0,2}``, ``final``;
NNUnet1D.py:28-107).

The ``nn.Conv1d`` / ``nn.Linear`` children only *hold* the parameters (PyTorch
layouts, so reference checkpoints load); their ``forward`` is never called.
``forward(x, t)`` runs the hand-scheduled HIP pipeline (channels-last, fused
concat / bias / embedding folding); ``ssm_grad`` runs the dual-number forward
(tangent rows = second half of the batch) and the hand-written backward, and
leaves d(mean SSM loss)/d(param) in ``.grad`` — no autograd tape, no double
backward.
"""
from __future__ import annotations

import os

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import MsgmError
from .NN import FlatParamMixin, NormalizeLogRadius
from .convnet import ConvOp, ConvOpSet, Stride2PairOp

GELU = ops.ACT_GELU


class ConvBlock1D(nn.Module):
    """conv k3 -> GELU -> conv k3 -> GELU (NNUnet1D.py:13-24); parameter holder."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.net = nn.Sequential(nn.Conv1d(in_ch, out_ch, kernel_size=3, padding=1), nn.GELU(),
                                 nn.Conv1d(out_ch, out_ch, kernel_size=3, padding=1), nn.GELU())


class UNet1D(nn.Module, FlatParamMixin):
    def __init__(self, input_dim, base_channels=32, channel_mults=(1, 2, 4), num_res_blocks=2,
                 premodule: Optional[str] = None, emb_dim=128):
        super().__init__()
        self.input_dim = input_dim
        assert premodule in (None, "NormalizeLogRadius")
        self.premodule = NormalizeLogRadius() if premodule == "NormalizeLogRadius" else None
        self.emb_dim = emb_dim
        self.time_mlp = nn.Sequential(nn.Linear(1, emb_dim), nn.GELU(), nn.Linear(emb_dim, emb_dim))
        # log-radius conditioning: scalar log||x|| -> emb_dim, added to the time embedding (NNUnet1D.py:59-69,137-145)
        self.scale_embed = nn.Sequential(nn.Linear(1, emb_dim), nn.GELU(), nn.Linear(emb_dim, emb_dim)) \
            if self.premodule is not None else None
        chs = [base_channels * m for m in channel_mults]
        self.chs = chs
        self.enc_blocks, self.downs = nn.ModuleList(), nn.ModuleList()
        in_ch = 1
        for out_ch in chs:
            self.enc_blocks.append(ConvBlock1D(in_ch + emb_dim, out_ch))
            self.downs.append(nn.Conv1d(out_ch, out_ch, kernel_size=4, stride=2, padding=1))
            in_ch = out_ch
        self.middle = ConvBlock1D(in_ch + emb_dim, in_ch)
        self.up_convs, self.dec_blocks = nn.ModuleList(), nn.ModuleList()
        for out_ch in reversed(chs):
            self.up_convs.append(nn.ConvTranspose1d(in_ch, out_ch, kernel_size=4, stride=2, padding=1))
            self.dec_blocks.append(ConvBlock1D(out_ch * 2 + emb_dim, out_ch))
            in_ch = out_ch
        self.final = nn.Conv1d(in_ch, 1, kernel_size=1)
        self._ops = None
        self._flat = None

    # ------------------------------------------------------------------ ops
    def _build(self):
        """(Re)create the ConvOps over the current parameter storage."""
        self.flat_parameters()
        first = self.time_mlp[0].weight
        if self._ops is not None and self._ops["t0"].weight is first and self._ops["t0"].Wp.device == first.device:
            return self._ops
        E, chs = self.emb_dim, self.chs
        # Downsample / Upsample (k=4, s=2, p=1) as 3-tap stride-1 convs over position PAIRS (convnet.Stride2PairOp):
        # needs an even length at every level and channel counts that are multiples of 16
        paired = (self.input_dim % (1 << len(chs)) == 0 and all(c % 16 == 0 for c in chs)
                  and not os.environ.get("MSGM_NO_PAIRED_STRIDE"))
        o = {"t0": ConvOp(self.time_mlp[0].weight, self.time_mlp[0].bias, "linear", (1,), 1, 0, [1]),
             "t2": ConvOp(self.time_mlp[2].weight, self.time_mlp[2].bias, "linear", (1,), 1, 0, [E])}
        if self.scale_embed is not None:
            o["s0"] = ConvOp(self.scale_embed[0].weight, self.scale_embed[0].bias, "linear", (1,), 1, 0, [1])
            o["s2"] = ConvOp(self.scale_embed[2].weight, self.scale_embed[2].bias, "linear", (1,), 1, 0, [E])
        cin = 1
        for i, c in enumerate(chs):
            b = self.enc_blocks[i].net
            o[f"e{i}a"] = ConvOp(b[0].weight, b[0].bias, "conv", (3,), 1, 1, [cin], emb_channels=E)
            o[f"e{i}b"] = ConvOp(b[2].weight, b[2].bias, "conv", (3,), 1, 1, [c])
            o[f"d{i}"] = (Stride2PairOp(self.downs[i].weight, self.downs[i].bias, "conv") if paired
                          else ConvOp(self.downs[i].weight, self.downs[i].bias, "conv", (4,), 2, 1, [c]))
            cin = c
        m = self.middle.net
        o["ma"] = ConvOp(m[0].weight, m[0].bias, "conv", (3,), 1, 1, [cin], emb_channels=E)
        o["mb"] = ConvOp(m[2].weight, m[2].bias, "conv", (3,), 1, 1, [cin])
        for i, c in enumerate(reversed(chs)):
            o[f"u{i}"] = (Stride2PairOp(self.up_convs[i].weight, self.up_convs[i].bias, "convT") if paired
                          else ConvOp(self.up_convs[i].weight, self.up_convs[i].bias, "convT", (4,), 2, 1, [cin]))
            b = self.dec_blocks[i].net
            o[f"x{i}a"] = ConvOp(b[0].weight, b[0].bias, "conv", (3,), 1, 1, [c, c], emb_channels=E)
            o[f"x{i}b"] = ConvOp(b[2].weight, b[2].bias, "conv", (3,), 1, 1, [c])
            cin = c
        o["fin"] = ConvOp(self.final.weight, self.final.bias, "conv", (1,), 1, 0, [cin])
        self._ops = o
        self._opset = ConvOpSet(list(o.values()))
        return o

    # ------------------------------------------------------------------ pipeline
    def _run(self, h0: torch.Tensor, t: torch.Tensor, N: int, Bp: int, L: int, dual: bool, tape: Optional[list]):
        """h0: [N][L][1] (primal rows then tangent rows).  Returns [N][L] output."""
        o = self._build()
        self._opset.pack()
        dev = h0.device
        nl = len(self.chs)
        act = lambda z: ops.act_dual_forward(GELU, z, torch.empty_like(z), dual)
        rec = tape.append if tape is not None else (lambda r: None)

        tt = t.reshape(Bp, 1).contiguous()
        ze, _, _ = o["t0"].forward([tt], Bp, 1, 1, Bp)
        he = ops.act_dual_forward(GELU, ze, torch.empty_like(ze), False)
        emb, _, _ = o["t2"].forward([he], Bp, 1, 1, Bp)
        er, pre = Bp, None
        if self.premodule is not None:
            # x <- sqrt(L) x/(|x|+eps); log-radius embedding carries a tangent (rows Bp..N-1) when dual
            h0, logr = ops.normalize_dual(h0, Bp, L, dual, float(L) ** 0.5)
            h0 = h0.view(-1)
            zs, _, _ = o["s0"].forward([logr], N, 1, 1, Bp)
            hs_ = ops.act_dual_forward(GELU, zs, torch.empty_like(zs), dual)
            sv, _, _ = o["s2"].forward([hs_], N, 1, 1, Bp)
            pv = sv[: Bp * self.emb_dim]
            ops.lincomb(pv, pv, 1.0, emb, 1.0)                      # t_emb + scale_vec        NNUnet1D.py:145
            pre = (logr, zs, hs_)
            emb, er = sv, N
        rec(("emb", tt, ze, he, emb, pre, er))

        def block(ka, kb, srcs, Lc):
            z1, _, _ = o[ka].forward(srcs, N, 1, Lc, Bp, emb=emb, emb_rows=er)
            h1 = act(z1)
            z2, _, _ = o[kb].forward([h1], N, 1, Lc, Bp)
            h2 = act(z2)
            rec(("block", ka, kb, srcs, Lc, z1, h1, z2))
            return h2

        h, Lc = h0, L
        skips = []
        for i in range(nl):
            h = block(f"e{i}a", f"e{i}b", [h], Lc)
            skips.append((h, Lc))
            hd, _, Ld = o[f"d{i}"].forward([h], N, 1, Lc, Bp)
            rec(("down", f"d{i}", h, Lc))
            h, Lc = hd, Ld
        h = block("ma", "mb", [h], Lc)
        for i in range(nl):
            up, _, Lu = o[f"u{i}"].forward([h], N, 1, Lc, Bp)
            skip, Ls = skips.pop()
            C = o[f"u{i}"].Cout
            pad_from = None
            if Lu != Ls:                       # F.pad(h, (0, Ls - Lu)) (NNUnet1D.py:171-172)
                upp = torch.zeros(N, Ls, C, device=dev)
                upp[:, :Lu].copy_(up.view(N, Lu, C))
                pad_from, up = Lu, upp.view(-1)
            rec(("up", f"u{i}", h, Lc, pad_from, Ls))
            h = block(f"x{i}a", f"x{i}b", [up, skip], Ls)
            Lc = Ls
        out, _, _ = o["fin"].forward([h], N, 1, Lc, Bp)
        rec(("fin", h, Lc))
        return out

    @torch.no_grad()
    def forward(self, x, t):
        """a(x, t): x (B, L) or (B,1,L), t (B,) or (B,1) -> (B, L) (NNUnet1D.py:110-179)."""
        if x.ndim == 3:
            x = x.squeeze(1)
        B, L = x.shape
        t = t.reshape(-1).float()
        if t.numel() == 1 and B != 1:
            t = t.expand(B)
        out = self._run(x.contiguous().float().reshape(-1), t.contiguous(), B, B, L, False, None)
        return out.view(B, L)

    # ------------------------------------------------------------------ training
    @torch.no_grad()
    def ssm_grad(self, y: torch.Tensor, t: torch.Tensor, v: torch.Tensor, u: torch.Tensor, cst: torch.Tensor,
                 inv_batch: float):
        """Per-sample SSM loss (B,) in the general form loss_b = adot.u + cst + |a|^2/2 (u, cst from
        ``msgm_ssm_terms``: any SDE family); d(sum_b loss_b * inv_batch)/d(params) is written into ``.grad``."""
        B, L = y.shape
        N = 2 * B
        flat, gflat = self.flat_parameters()
        for p in self.parameters():
            if p.grad is None:
                self._flatten_parameters()
                break
        o = self._build()
        self._opset.zero_grad_images(bias_grads_zeroed=True)
        self.flat_parameters()[1].zero_()          # bias gradients are accumulated as a by-product of wgrad
        tape = []
        h0 = torch.cat([y.contiguous().float(), v.contiguous().float()], 0).reshape(-1)
        out = self._run(h0, t.reshape(-1).contiguous().float(), N, B, L, True, tape)
        per, g = ops.ssm_loss(out, u, cst, inv_batch)
        with ops.DeferredReduces.on(y.device):           # all slot reductions of the weight / bias gradients in one launch
            self._backward(tape, g, N, B)
        self._opset.unpack_grads()
        return per

    def _backward(self, tape, g, N, Bp):
        o = self._ops
        dev = g.device
        E = self.emb_dim
        emb_rec = tape[0]
        emb, pre, er = emb_rec[4], emb_rec[5], emb_rec[6]
        demb = torch.zeros(er * E, device=dev)
        pending_skip = []                      # gradients w.r.t. skip tensors, consumed by the encoder
        dh = g
        for r in reversed(tape[1:]):
            kind = r[0]
            if kind == "fin":
                _, h, Lc = r
                (dh,) = o["fin"].backward(dh, [h], N, 1, Lc, Bp)
            elif kind == "block":
                _, ka, kb, srcs, Lc, z1, h1, z2 = r
                g2 = ops.act_dual_backward(GELU, z2, dh)
                (dh1,) = o[kb].backward(g2, [h1], N, 1, Lc, Bp)
                g1 = ops.act_dual_backward(GELU, z1, dh1)
                first = ka == "e0a"
                ds = o[ka].backward(g1, srcs, N, 1, Lc, Bp, emb=emb, demb=demb, need=[not first] + [True] * (len(srcs) - 1),
                                    emb_rows=er)
                if len(srcs) == 2:
                    pending_skip.append(ds[1])
                dh = ds[0]
            elif kind == "up":
                _, ku, h, Lc, pad_from, Ls = r
                if pad_from is not None:
                    C = o[ku].Cout
                    dh = dh.view(N, Ls, C)[:, :pad_from].contiguous().view(-1)
                (dh,) = o[ku].backward(dh, [h], N, 1, Lc, Bp)
            elif kind == "down":
                _, kd, h, Lc = r
                dskip = pending_skip.pop()
                (dh,) = o[kd].backward(dh, [h], N, 1, Lc, Bp, dsrc=[dskip], dacc=[True])
        if pre is not None:                    # log-radius embedding (primal | tangent rows): Linear -> GELU -> Linear
            logr, zs, hs_ = pre
            (dhs,) = o["s2"].backward(demb, [hs_], N, 1, 1, Bp)
            ops.act_dual_backward(GELU, zs, dhs)
            o["s0"].backward(dhs, [logr], N, 1, 1, Bp, need=[False])
        # time MLP (primal rows only): Linear -> GELU -> Linear
        tt, ze, he = emb_rec[1], emb_rec[2], emb_rec[3]
        (dhe,) = o["t2"].backward(demb[: Bp * E].contiguous(), [he], Bp, 1, 1, Bp)
        zz = torch.cat([ze, torch.zeros_like(ze)])
        gg = torch.cat([dhe, torch.zeros_like(dhe)])
        ops.act_dual_backward(GELU, zz, gg)
        o["t0"].backward(gg[: ze.numel()].contiguous(), [tt], Bp, 1, 1, Bp, need=[False])
