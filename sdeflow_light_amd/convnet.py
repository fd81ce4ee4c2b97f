"""Building blocks shared by the U-Net score nets (NNUnet1D / NNUnet): one
``ConvOp`` = one Conv1d / Conv2d / ConvTranspose1d / Linear of the reference
run on the implicit-GEMM HIP kernels (csrc/conv_kernels.hip), with its own
packed weights and hand-written backward (dgrad / wgrad / bias / embedding).

Conventions
-----------
* activations are channels-last ``[N][H][W][C]`` fp32 (1-D: H = 1);
* the forward-mode tangent (J.v of the SSM loss) rides as the second half of
  the batch: rows ``n >= n_bias`` are tangent rows and receive no bias — conv /
  linear layers are linear, so they need no other dual-number logic;
* parameters stay in the reference's PyTorch layouts (state_dict compatible);
  every step they are re-packed into ``[tap][CoutP][K]`` (forward) and
  ``[tap][CinP][CoutK]`` (dgrad) images by ``msgm_pack_weight`` (a few MB);
* the 128 broadcast time-embedding channels the 1-D U-Net concatenates in front
  of every block (NNUnet1D.py:156,162,175) are NOT convolved: a channel that is
  constant along L contributes a per-sample bias ``sum_t E_t`` with
  ``E_t = emb . W[:, emb_ch, t]^T`` except at the two zero-padded borders, where
  the missing tap is subtracted again (exact up to fp32 re-association; removes
  ~36 % of the 1-D U-Net's FLOPs).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import ops
from ._lib import MsgmError

pad16 = ops.pad16


class ConvOpSet:
    """All ConvOps of one network: their gradient images live in ONE flat buffer (one memset per step) and their
    weight (un)packing runs as ONE launch each (ops.PackTable) instead of ~4 tiny launches per op."""

    def __init__(self, ops_list: List["ConvOp"]):
        self.ops = list(ops_list)
        dev = self.ops[0].weight.device
        names = ("dWp", "dWpE", "dbias_i")                       # every gradient image an op accumulates into
        total = sum(getattr(o, a).numel() for o in self.ops for a in names if getattr(o, a, None) is not None)
        self.gimg = torch.zeros(total, device=dev)
        off = 0
        for o in self.ops:
            for a in names:
                t = getattr(o, a, None)
                if t is not None:
                    n = t.numel()
                    setattr(o, a, self.gimg[off:off + n]); off += n
        self._pack = self._unpack = self._pack_w = None
        self._pack_sig = self._unpack_sig = self._pack_w_sig = None

    def pack(self):
        sig = tuple(o.weight.data_ptr() for o in self.ops)          # the parameter bucket may have been rebuilt
        if self._pack is None or sig != self._pack_sig:
            self._pack = ops.PackTable([j for o in self.ops for j in o.pack_jobs()], self.gimg.device)
            self._pack_sig = sig
        self._pack.run(False)

    def pack_wino(self, train: bool = False):
        """Winograd images G g G^T of every 3x3 stride-1 convolution of the set, ONE launch.  train = False (sampler path):
        the forward images; the caller asks for the kernel per call (forward(wino=True)).  train = True (experiment switch
        MSGM_TRAIN_WINO): also the images of the FLIPPED, transposed kernels (the dgrad as a Winograd forward), and every
        capable op takes the Winograd kernel for its forward and its dgrad by itself."""
        wops = [o for o in self.ops if getattr(o, "wino_capable", lambda: False)()]
        for o in self.ops:
            if hasattr(o, "train_wino"):
                o.train_wino = False
        if not wops:
            return
        sig = (bool(train),) + tuple(o.weight.data_ptr() for o in wops)
        if self._pack_w is None or sig != self._pack_w_sig:
            jobs = [j for o in wops for j in o.wino_jobs()]
            if train:
                jobs += [j for o in wops for j in o.wino_dgrad_jobs()]
            self._pack_w = ops.PackTable(jobs, self.gimg.device)
            self._pack_w_sig = sig
        self._pack_w.run_wino()
        if train:
            for o in wops:
                o.train_wino = True

    def clear_train_wino(self):
        for o in self.ops:
            if hasattr(o, "train_wino"):
                o.train_wino = False

    def pack_b6(self, use: bool = True):
        """bf16-split images of every 3x3 stride-1 convolution of the set from its packed fp32 image (opt-in experiment,
        sampler path; run after pack()).  use = False only clears the ops' switch (a net that sampled with it before)."""
        for o in self.ops:
            cap = use and getattr(o, "b6_capable", lambda: False)()
            if cap:
                o.pack_b6()
            if hasattr(o, "use_b6"):
                o.use_b6 = bool(cap)

    def zero_grad_images(self, bias_grads_zeroed: bool = False):
        """bias_grads_zeroed: the caller has zeroed the flat gradient bucket for this step, so each op's next
        backward() may accumulate its bias gradient without a memset of its own."""
        self.gimg.zero_()
        for o in self.ops:
            o._bias_zeroed = bool(bias_grads_zeroed)

    def unpack_grads(self):
        sig = tuple(o.weight.grad.data_ptr() for o in self.ops)
        if self._unpack is None or sig != self._unpack_sig:
            self._unpack = ops.PackTable([j for o in self.ops for j in o.unpack_jobs()], self.gimg.device)
            self._unpack_sig = sig
        self._unpack.run(True)


class ConvOp:
    def __init__(self, weight: torch.nn.Parameter, bias: Optional[torch.nn.Parameter], kind: str, ksize: Sequence[int],
                 stride: int, pad: int, src_channels: Sequence[int], emb_channels: int = 0, ups: bool = False):
        self.weight, self.bias, self.kind = weight, bias, kind
        self.KH, self.KW = (1, ksize[0]) if len(ksize) == 1 else (ksize[0], ksize[1])
        self.taps = self.KH * self.KW
        self.stride, self.pad, self.ups = stride, pad, ups
        self.srcC = list(src_channels)
        self.embC = emb_channels
        self.cin_tot = sum(self.srcC) + emb_channels
        if len(self.srcC) > 2:
            raise MsgmError("at most two concatenated sources")
        if kind in ("conv", "linear"):
            self.Cout = weight.shape[0]
            assert weight.shape[1] == self.cin_tot, (weight.shape, self.cin_tot)
            self.s_row, self.s_col = self.cin_tot * self.taps, self.taps      # element (co, ci, t) strides
            self.mode = 0
        elif kind == "convT":
            self.Cout = weight.shape[1]
            assert weight.shape[0] == self.cin_tot and emb_channels == 0
            self.s_row, self.s_col = self.taps, self.Cout * self.taps          # (co, ci, t) in a (Cin, Cout, k) tensor
            self.mode = 1
        else:
            raise MsgmError(f"unknown conv kind {kind}")
        dev = weight.device
        self.CoutP = pad16(self.Cout)
        self.koff = [0] + [pad16(self.srcC[0])] if len(self.srcC) == 2 else [0]
        self.Ktot = sum(pad16(c) for c in self.srcC)
        self.Wp = torch.zeros(self.taps * self.CoutP * self.Ktot, device=dev)
        self.dWp = torch.zeros_like(self.Wp)
        self.Wd = [torch.zeros(self.taps * pad16(c) * pad16(self.Cout), device=dev) for c in self.srcC]
        if emb_channels:
            if self.Cout % 16 or self.taps != 3 or self.pad != 1 or stride != 1:
                raise MsgmError("embedding-channel folding is built for k=3, pad=1, stride=1, Cout % 16 == 0")
            self.WpE = torch.zeros(3 * self.Cout * pad16(emb_channels), device=dev)
            self.dWpE = torch.zeros_like(self.WpE)
            self.WdE = torch.zeros(3 * pad16(emb_channels) * pad16(self.Cout), device=dev)
        self._E = None
        self._bias_zeroed = False      # set per step by ConvOpSet.zero_grad_images, consumed by the next backward
        self.WpW = None                # Winograd F(2x2,3x3) image [16][CoutP][Ktot] (sampler path), built on demand
        self.Wb = None                 # bf16-split image [3][9][CoutP][Ktot] (opt-in experiment), built on demand
        self.use_b6 = False            # forward(wino=True) takes the bf16-split kernel instead (ConvOpSet.pack_b6)
        self.WdW = [None] * len(self.srcC)   # Winograd images of the flipped / transposed kernels (dgrad), per source
        self.train_wino = False        # forward and dgrad take the Winograd kernel by themselves (ConvOpSet.pack_wino(train=True))

    def wino_capable(self) -> bool:
        return (self.kind == "conv" and self.KH == 3 and self.KW == 3 and self.stride == 1 and self.pad == 1 and not self.embC
                and self.CoutP % 32 == 0 and all(c % 16 == 0 for c in self.srcC))

    def wino_dgrad_jobs(self):
        """The dgrad of a 3x3 stride-1 pad-1 convolution is the same convolution of gy with the kernels flipped and
        transposed: g'[ci][co][t] = W[co][ci][8 - t] — a pack job that starts at tap 8 and walks the taps backwards."""
        W = self.weight.detach()
        jobs, off = [], 0
        for s, C in enumerate(self.srcC):
            if C % 32 == 0 and self.Cout % 16 == 0:
                if self.WdW[s] is None:
                    self.WdW[s] = torch.zeros(16 * pad16(C) * pad16(self.Cout), device=W.device)
                jobs.append((W, off * self.s_col + 8, self.WdW[s], C, self.Cout, 0, 9, self.s_col, self.s_row, -1, pad16(C),
                             pad16(self.Cout), 0))
            off += C
        return jobs

    def b6_capable(self) -> bool:
        return (self.kind == "conv" and self.KH == 3 and self.KW == 3 and self.stride == 1 and self.pad == 1 and not self.embC
                and self.CoutP % 32 == 0 and all(c % 32 == 0 for c in self.srcC))

    def pack_b6(self):
        if self.Wb is None:
            self.Wb = torch.zeros(3 * self.Wp.numel() + 32, dtype=torch.bfloat16, device=self.weight.device)
        ops.b6_split_weights(self.Wp, self.Wb)

    def wino_jobs(self):
        if self.WpW is None:
            self.WpW = torch.zeros(16 * self.CoutP * self.Ktot, device=self.weight.device)
        W = self.weight.detach()
        jobs, off = [], 0
        for s, C in enumerate(self.srcC):
            jobs.append((W, 0, self.WpW, self.Cout, C, off, 9, self.s_row, self.s_col, 1, self.CoutP, self.Ktot, self.koff[s]))
            off += C
        return jobs

    # -------------------------------------------------------------- packing
    def pack(self):
        W = self.weight.detach()
        off = 0
        for s, C in enumerate(self.srcC):
            ops.pack_weight(W, 0, self.Wp, self.Cout, C, off, self.taps, self.s_row, self.s_col, 1, self.CoutP, self.Ktot,
                            self.koff[s])
            # dgrad image: rows = input channels of this source, K = output channels
            ops.pack_weight(W, off * self.s_col, self.Wd[s], C, self.Cout, 0, self.taps, self.s_col, self.s_row, 1, pad16(C),
                            pad16(self.Cout), 0)
            off += C
        if self.embC:
            E = self.embC
            ops.pack_weight(W, 0, self.WpE, self.Cout, E, off, 3, self.s_row, self.s_col, 1, self.Cout, pad16(E), 0)
            ops.pack_weight(W, off * self.s_col, self.WdE, E, self.Cout, 0, 3, self.s_col, self.s_row, 1, pad16(E),
                            pad16(self.Cout), 0)

    def pack_jobs(self):
        """The jobs of pack() as tuples for ops.PackTable."""
        W = self.weight.detach()
        jobs, off = [], 0
        for s, C in enumerate(self.srcC):
            jobs.append((W, 0, self.Wp, self.Cout, C, off, self.taps, self.s_row, self.s_col, 1, self.CoutP, self.Ktot, self.koff[s]))
            jobs.append((W, off * self.s_col, self.Wd[s], C, self.Cout, 0, self.taps, self.s_col, self.s_row, 1, pad16(C),
                         pad16(self.Cout), 0))
            off += C
        if self.embC:
            E = self.embC
            jobs.append((W, 0, self.WpE, self.Cout, E, off, 3, self.s_row, self.s_col, 1, self.Cout, pad16(E), 0))
            jobs.append((W, off * self.s_col, self.WdE, E, self.Cout, 0, 3, self.s_col, self.s_row, 1, pad16(E), pad16(self.Cout), 0))
        return jobs

    def unpack_jobs(self):
        """The jobs of unpack_grads() (weight.grad must exist)."""
        gW = self.weight.grad
        jobs, off = [], 0
        for s, C in enumerate(self.srcC):
            jobs.append((gW, 0, self.dWp, self.Cout, C, off, self.taps, self.s_row, self.s_col, 1, self.CoutP, self.Ktot, self.koff[s]))
            off += C
        if self.embC:
            jobs.append((gW, 0, self.dWpE, self.Cout, self.embC, off, 3, self.s_row, self.s_col, 1, self.Cout, pad16(self.embC), 0))
        return jobs

    def zero_grad_images(self):
        self.dWp.zero_()
        if self.embC:
            self.dWpE.zero_()

    def unpack_grads(self):
        """packed gradient images -> .grad of weight (PyTorch layout); bias grad is written by backward()."""
        gW = self.weight.grad
        off = 0
        for s, C in enumerate(self.srcC):
            ops.unpack_weight(gW, 0, self.dWp, self.Cout, C, off, self.taps, self.s_row, self.s_col, 1, self.CoutP, self.Ktot,
                              self.koff[s])
            off += C
        if self.embC:
            ops.unpack_weight(gW, 0, self.dWpE, self.Cout, self.embC, off, 3, self.s_row, self.s_col, 1, self.Cout,
                              pad16(self.embC), 0)

    # -------------------------------------------------------------- geometry
    def out_hw(self, Hi, Wi):
        Hu, Wu = (2 * Hi, 2 * Wi) if self.ups else (Hi, Wi)
        if self.mode == 0:
            f = lambda n, k: (n + 2 * self.pad - k) // self.stride + 1
        else:
            f = lambda n, k: (n - 1) * self.stride - 2 * self.pad + k
        return (f(Hu, self.KH) if self.KH > 1 or Hu > 1 else 1), f(Wu, self.KW)

    def _geom(self, N, Hi, Wi):
        Ho, Wo = self.out_hw(Hi, Wi)
        return ops.conv_geom(N, Hi, Wi, Ho, Wo, self.KH, self.KW, self.stride, self.pad, self.mode, int(self.ups)), Ho, Wo

    def can_transform_input(self, N: int, Hi: int, Wi: int) -> bool:
        """True when forward() can apply a per-(sample, channel) affine (+SiLU) to its input on the fly."""
        geom, _, _ = self._geom(N, Hi, Wi)
        return ops.conv_input_transform_supported(geom, self.srcC[0], self.srcC[1] if len(self.srcC) > 1 else 0, self.CoutP,
                                                  Cout=self.Cout)

    # -------------------------------------------------------------- forward
    def forward(self, srcs: List[torch.Tensor], N: int, Hi: int, Wi: int, n_bias: int, emb: Optional[torch.Tensor] = None,
                samp_bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, accumulate=False,
                emb_rows: Optional[int] = None, residual: Optional[torch.Tensor] = None,
                in_affine: Optional[tuple] = None, in_act: int = 0, wino: bool = False, stats: bool = False, b6: bool = False):
        """emb_rows: rows that carry an embedding / per-sample bias (n_bias by default; N when the embedding itself
        has a tangent — NormalizeLogRadius conditioning).  wino: take the Winograd F(2x2,3x3) forward kernel when this
        op has its image (ConvOpSet.pack_wino) and the geometry allows it (sampler path).
        stats: if the kernel serving this convolution can, leave the per-channel sums of the output on the returned
        tensor as ``out._msgm_cs = (chanstats, slots)`` for the GroupNorm that reads it next."""
        geom, Ho, Wo = self._geom(N, Hi, Wi)
        b6 = bool((b6 or (wino and self.use_b6)) and self.Wb is not None and
                  ops.conv_b6_supported(geom, self.srcC[0], self.srcC[1] if len(srcs) > 1 else 0, self.CoutP))
        wino = bool(not b6 and (wino or self.train_wino) and self.WpW is not None and
                    ops.conv_wino_supported(geom, self.srcC[0], self.srcC[1] if len(srcs) > 1 else 0, self.CoutP))
        dev = srcs[0].device
        if out is None:
            out = torch.empty(N * Ho * Wo * self.Cout, device=dev)
        sb = samp_bias
        er = n_bias if emb_rows is None else emb_rows
        if self.embC:
            Bp = er
            g1 = ops.conv_geom(Bp, 1, 1, 1, 1, 1, 1, 1, 0)
            E = [torch.empty(Bp * self.Cout, device=dev) for _ in range(3)]
            per = self.Cout * pad16(self.embC)
            for t in range(3):
                ops.conv_forward(g1, emb, self.embC, self.WpE[t * per:(t + 1) * per], self.Cout, E[t], CoutP=self.Cout)
            sb = torch.empty(Bp * self.Cout, device=dev)
            ops.lincomb(sb, E[0], 1.0, E[1], 1.0, E[2], 1.0)
            self._E = E
        cs, S = None, 0
        out._msgm_cs = None                      # whatever was there described the values about to be overwritten
        if stats and not self.embC:
            S = ((Ho // 16) * (Wo // 16) * 4 if (wino or b6) and self.Cout % 4 == 0 else 0 if (wino or b6) else
                 ops.conv_chanstats_slots(geom, self.srcC[0], self.srcC[1] if len(srcs) > 1 else 0, self.Cout, self.CoutP))
            if S > 0:
                cs = torch.empty(N * S * 2 * self.Cout, device=dev)
        ops.conv_forward(geom, srcs[0], self.srcC[0], self.Wb if b6 else (self.WpW if wino else self.Wp), self.Cout, out,
                         src1=srcs[1] if len(srcs) > 1 else None, C1=self.srcC[1] if len(srcs) > 1 else 0,
                         bias=self.bias.detach() if self.bias is not None else None, samp_bias=sb, n_bias=n_bias,
                         accumulate=accumulate, CoutP=self.CoutP, n_samp=er, residual=residual,
                         in_scale=in_affine[0] if in_affine is not None else None,
                         in_shift=in_affine[1] if in_affine is not None else None, in_act=in_act, wino=wino, chanstats=cs, b6=b6)
        if cs is not None:
            out._msgm_cs = (cs, S)
        if self.embC:
            # taps 0 / 2 fall on the zero padding at l = 0 / L-1 (rows that carry an embedding)
            ops.add_row(out, self._E[0], er, Ho * Wo, self.Cout, 0, -1.0)
            ops.add_row(out, self._E[2], er, Ho * Wo, self.Cout, Ho * Wo - 1, -1.0)
        return out, Ho, Wo

    # -------------------------------------------------------------- backward
    def backward(self, gy: torch.Tensor, srcs: List[torch.Tensor], N: int, Hi: int, Wi: int, n_bias: int,
                 emb: Optional[torch.Tensor] = None, demb: Optional[torch.Tensor] = None,
                 need: Optional[Sequence[bool]] = None, dsrc: Optional[List[Optional[torch.Tensor]]] = None,
                 dacc: Optional[Sequence[bool]] = None, dsamp_bias: Optional[torch.Tensor] = None,
                 emb_rows: Optional[int] = None, bias_grad_zeroed: bool = False, bias_grad_elsewhere: bool = False):
        """gy: cotangent of the output [N][Ho][Wo][Cout].  Accumulates the packed
        weight gradient, writes bias.grad, adds to ``demb`` / writes ``dsamp_bias``
        (per-sample bias cotangent, primal rows) and returns d(src_s).
        bias_grad_elsewhere: with ``dsamp_bias``, the caller derives bias.grad from the per-sample sums itself (ops.EmbBank)."""
        geom, Ho, Wo = self._geom(N, Hi, Wi)
        dev = gy.device
        P = Ho * Wo
        # bias only (no per-sample bias / embedding): the gradient is a by-product of the first source's wgrad
        fuse_bias = self.bias is not None and not self.embC and dsamp_bias is None
        bias_grad_zeroed = bias_grad_zeroed or self._bias_zeroed
        self._bias_zeroed = False
        if fuse_bias and not bias_grad_zeroed:
            self.bias.grad.zero_()
        for s, C in enumerate(self.srcC):
            ops.conv_wgrad(geom, gy, srcs[s], C, self.koff[s], self.dWp, self.Cout, self.CoutP, self.Ktot,
                           dbias=self.bias.grad.view(-1) if (fuse_bias and s == 0) else None, n_bias=n_bias)
        S = None
        er = n_bias if emb_rows is None else emb_rows
        if fuse_bias:
            pass
        elif self.bias is not None or self.embC or dsamp_bias is not None:
            S = dsamp_bias if dsamp_bias is not None else torch.empty(er * self.Cout, device=dev)
            ops.colsum(gy, er, P, self.Cout, out=S)                           # rows that carry a bias / embedding
            if self.bias is not None and not (bias_grad_elsewhere and dsamp_bias is not None):   # the bias itself: primal rows only
                ops.colsum(S, 1, n_bias, self.Cout, out=self.bias.grad.view(1, -1))
        if self.embC:
            Bp, E = er, self.embC
            g0 = ops.gather_row(gy, Bp, P, self.Cout, 0)
            gL = ops.gather_row(gy, Bp, P, self.Cout, P - 1)
            G = [torch.empty(Bp * self.Cout, device=dev), S, torch.empty(Bp * self.Cout, device=dev)]
            ops.lincomb(G[0], S, 1.0, g0, -1.0)
            ops.lincomb(G[2], S, 1.0, gL, -1.0)
            gE = ops.conv_geom(Bp, 1, 1, 1, 1, 1, 1, 1, 0)
            perE, perD = self.Cout * pad16(E), pad16(E) * pad16(self.Cout)
            for t in range(3):
                # demb[b][ci] += sum_co G_t[b][co] W[co][emb ci][t]
                ops.conv_forward(gE, G[t], self.Cout, self.WdE[t * perD:(t + 1) * perD], E, demb, accumulate=True, CoutP=pad16(E))
                # dW[co][emb ci][t] = sum_b G_t[b][co] emb[b][ci]
                ops.conv_wgrad(gE, G[t], emb, E, 0, self.dWpE[t * perE:(t + 1) * perE], self.Cout, self.Cout, pad16(E))
        outs = []
        need = [True] * len(self.srcC) if need is None else need
        gd = ops.conv_geom(N, Ho, Wo, Hi, Wi, self.KH, self.KW, self.stride, self.pad, 1 - self.mode, 0)
        for s, C in enumerate(self.srcC):
            if not need[s]:
                outs.append(None)
                continue
            if self.ups:
                raise MsgmError("dgrad through a folded upsample is handled by the caller (sum of 2x2 blocks)")
            d = dsrc[s] if (dsrc is not None and dsrc[s] is not None) else torch.empty(N * Hi * Wi * C, device=dev)
            g0 = ops.conv_geom(N, Ho, Wo, Hi, Wi, self.KH, self.KW, 1, 1, 0, 0) if (self.train_wino and self.WdW[s] is not None) else None
            if g0 is not None and ops.conv_wino_supported(g0, self.Cout, 0, pad16(C)):
                ops.conv_forward(g0, gy, self.Cout, self.WdW[s], C, d, accumulate=bool(dacc[s]) if dacc is not None else False,
                                 CoutP=pad16(C), wino=True)
            else:
                ops.conv_forward(gd, gy, self.Cout, self.Wd[s], C, d, accumulate=bool(dacc[s]) if dacc is not None else False,
                                 CoutP=pad16(C))
            outs.append(d)
        return outs

    def backward_ups(self, gy: torch.Tensor, src: torch.Tensor, N: int, Hi: int, Wi: int, n_bias: int,
                     bias_grad_zeroed: bool = False) -> torch.Tensor:
        """Backward of a conv whose input is nearest-upsampled 2x on the fly (Upsample, model/unet.py:60-73):
        wgrad through the folded gather, dgrad on the 2x grid, then the 2x2 block sum (adjoint of the upsample)."""
        if not self.ups or len(self.srcC) != 1:
            raise MsgmError("backward_ups is for a single-source folded-upsample conv")
        geom, Ho, Wo = self._geom(N, Hi, Wi)
        dev, C = gy.device, self.srcC[0]
        bias_grad_zeroed = bias_grad_zeroed or self._bias_zeroed
        self._bias_zeroed = False
        if self.bias is not None and not bias_grad_zeroed:
            self.bias.grad.zero_()
        ops.conv_wgrad(geom, gy, src, C, 0, self.dWp, self.Cout, self.CoutP, self.Ktot,
                       dbias=self.bias.grad.view(-1) if self.bias is not None else None, n_bias=n_bias)
        gd = ops.conv_geom(N, Ho, Wo, 2 * Hi, 2 * Wi, self.KH, self.KW, self.stride, self.pad, 1 - self.mode, 0)
        gup = torch.empty(N * 4 * Hi * Wi * C, device=dev)
        g0 = ops.conv_geom(N, Ho, Wo, 2 * Hi, 2 * Wi, self.KH, self.KW, 1, 1, 0, 0) if (self.train_wino and self.WdW[0] is not None) else None
        if g0 is not None and ops.conv_wino_supported(g0, self.Cout, 0, pad16(C)):
            ops.conv_forward(g0, gy, self.Cout, self.WdW[0], C, gup, CoutP=pad16(C), wino=True)     # the dgrad on the 2x grid: a plain same-size conv
        else:
            ops.conv_forward(gd, gy, self.Cout, self.Wd[0], C, gup, CoutP=pad16(C))
        return ops.sum2x2(gup, N, Hi, Wi, C)


class Stride2PairOp:
    """Conv1d(k=4, s=2, p=1) / ConvTranspose1d(k=4, s=2, p=1) — the 1-D U-Net's Downsample / Upsample
    (NNUnet1D.py:84,98) — run on the stride-1 halo-tile kernels.

    In channels-last memory two neighbouring positions (2j, 2j+1) of C channels ARE one position of 2C channels, so
    a reinterpretation (no copy) turns the stride-2 conv into a 3-tap stride-1 "same" conv over pairs:
        out[o] = W0 x[2o-1] + W1 x[2o] + W2 x[2o+1] + W3 x[2o+2]
               = [0 | W0] xp[o-1] + [W1 | W2] xp[o] + [W3 | 0] xp[o+1],     xp[j] = [x[2j] | x[2j+1]],
    and the transposed conv into a 3-tap conv whose OUTPUT is pairs:
        [out[2j] | out[2j+1]] = [W3 | 0] x[j-1] + [W1 | W2] x[j] + [0 | W0] x[j+1].
    Two of the six (tap, half) weight blocks are zero (1.5x the FLOPs), but forward, dgrad and wgrad all become
    shapes the LDS-tiled kernels serve (measured: the strided implicit GEMM ran these layers at ~40 TFLOP/s, the tile
    kernels at ~90).  Same interface as ConvOp (forward / backward / pack_jobs / unpack_jobs); L must be even.
    Blocks (tap t, half) -> original tap k:  conv: (0,1)->0 (1,0)->1 (1,1)->2 (2,0)->3;  convT: (0,0)->3 (1,0)->1
    (1,1)->2 (2,1)->0."""

    BLOCKS = {"conv": ((0, 1, 0), (1, 0, 1), (1, 1, 2), (2, 0, 3)), "convT": ((0, 0, 3), (1, 0, 1), (1, 1, 2), (2, 1, 0))}

    def __init__(self, weight: torch.nn.Parameter, bias: Optional[torch.nn.Parameter], kind: str):
        if kind not in ("conv", "convT") or weight.shape[2] != 4:
            raise MsgmError("Stride2PairOp is Conv1d / ConvTranspose1d with k=4, s=2, p=1")
        self.weight, self.bias, self.kind = weight, bias, kind
        if kind == "conv":
            self.Cout, self.C = weight.shape[0], weight.shape[1]
            self.cin_i, self.cout_i = 2 * self.C, self.Cout                 # inner conv: pairs in
        else:
            self.C, self.Cout = weight.shape[0], weight.shape[1]
            self.cin_i, self.cout_i = self.C, 2 * self.Cout                 # inner conv: pairs out
        if self.C % 16 or self.Cout % 16:
            raise MsgmError("Stride2PairOp needs channel counts that are multiples of 16")
        dev = weight.device
        self.embC, self.srcC = 0, [self.C]
        self.CoutP, self.Ktot = pad16(self.cout_i), pad16(self.cin_i)
        self.Wp = torch.zeros(3 * self.CoutP * self.Ktot, device=dev)              # forward image of the inner conv
        self.dWp = torch.zeros_like(self.Wp)
        self.Wd = torch.zeros(3 * pad16(self.cin_i) * pad16(self.cout_i), device=dev)    # dgrad image
        self.bias_i = torch.zeros(self.cout_i, device=dev) if bias is not None else None   # convT: [b | b]
        self.dbias_i = torch.zeros(self.cout_i, device=dev) if (bias is not None and kind == "convT") else None
        self._bias_zeroed = False
        # taps present per half of the paired channel axis (bit t = tap t): the tile kernels skip the zero blocks
        half_taps = [0, 0]
        for t, half, _ in self.BLOCKS[kind]:
            half_taps[half] |= 1 << t
        nch = 2 * (self.C if kind == "conv" else self.Cout)                 # the paired axis: 2C in / 2Cout out
        per_half = nch // 2

        def masks(block):                                                   # one mask per `block` channels of the axis
            if per_half % block:
                return None                                                 # a block straddles the halves: keep all taps
            return [half_taps[(i * block) // per_half] for i in range(nch // block)]
        self.mask32 = masks(32)                                             # per 32-channel input chunk
        self.mask_ob = masks(64 if pad16(nch) % 64 == 0 else 32)            # per output-channel block of the tile kernel

    # ---- geometry
    def out_hw(self, Hi, Wi):
        return 1, (Wi // 2 if self.kind == "conv" else 2 * Wi)

    def _inner(self, N, Wi):
        if self.kind == "conv" and Wi % 2:
            raise MsgmError("Stride2PairOp: odd length (use the generic ConvOp)")
        Lp = Wi // 2 if self.kind == "conv" else Wi                                     # positions of the inner conv
        return ops.conv_geom(N, 1, Lp, 1, Lp, 1, 3, 1, 1, 0, 0), Lp

    # ---- weight images
    def _jobs(self, Wt, img_f, img_d, bias_t, bias_img):
        """(un)pack jobs between a tensor in the weight's PyTorch layout and the inner conv's images."""
        jobs = []
        C, Co = self.C, self.Cout
        if self.kind == "conv":                       # W (Cout, Cin, 4): (co, ci, k) at co*Cin*4 + ci*4 + k
            for t, half, k in self.BLOCKS["conv"]:
                if img_f is not None:
                    jobs.append((Wt, k, img_f[t * self.CoutP * self.Ktot:], Co, C, 0, 1, C * 4, 4, 1, self.CoutP, self.Ktot, half * C))
                if img_d is not None:                 # rows = inner input channel (half*C + ci), K = co
                    rp, kt = pad16(self.cin_i), pad16(self.cout_i)
                    jobs.append((Wt, k, img_d[(t * rp + half * C) * kt:], C, Co, 0, 1, 4, C * 4, 1, C, kt, 0))
        else:                                         # W (Cin, Cout, 4): (ci, co, k) at ci*Cout*4 + co*4 + k
            for t, half, k in self.BLOCKS["convT"]:
                if img_f is not None:                 # rows = inner output channel (half*Cout + co), K = ci
                    jobs.append((Wt, k, img_f[(t * self.CoutP + half * Co) * self.Ktot:], Co, C, 0, 1, 4, Co * 4, 1, Co, self.Ktot, 0))
                if img_d is not None:                 # rows = ci, K = inner output channel
                    rp, kt = pad16(self.cin_i), pad16(self.cout_i)
                    jobs.append((Wt, k, img_d[t * rp * kt:], C, Co, 0, 1, Co * 4, 4, 1, rp, kt, half * Co))
        if bias_t is not None and bias_img is not None and self.kind == "convT":
            for half in (0, 1):
                jobs.append((bias_t, 0, bias_img[half * Co:], 1, Co, 0, 1, 0, 1, 1, 1, Co, 0))
        return jobs

    def pack_jobs(self):
        return self._jobs(self.weight.detach(), self.Wp, self.Wd, self.bias.detach() if self.bias is not None else None,
                          self.bias_i)

    def unpack_jobs(self):
        """Gradient images -> .grad (weight; convT bias: the two halves are ADDED into the zeroed bias.grad)."""
        jobs = self._jobs(self.weight.grad, self.dWp, None, None, None)
        if self.dbias_i is not None:
            for half in (0, 1):
                jobs.append((self.bias.grad, 0, self.dbias_i[half * self.Cout:], 1, self.Cout, 0, 1, 0, 1, 1, 1, self.Cout, 0, True))
        return jobs

    def pack(self):
        ops.PackTable(self.pack_jobs(), self.weight.device).run(False)

    def zero_grad_images(self):
        self.dWp.zero_()
        if self.dbias_i is not None:
            self.dbias_i.zero_()

    def unpack_grads(self):
        ops.PackTable(self.unpack_jobs(), self.weight.device).run(True)

    # ---- forward / backward (ConvOp's interface)
    def forward(self, srcs, N, Hi, Wi, n_bias, out=None, accumulate=False, **_):
        geom, Lp = self._inner(N, Wi)
        _, Lo = self.out_hw(Hi, Wi)
        if out is None:
            out = torch.empty(N * Lo * self.Cout, device=srcs[0].device)
        b = None if self.bias is None else (self.bias.detach() if self.kind == "conv" else self.bias_i)
        ops.conv_forward(geom, srcs[0], self.cin_i, self.Wp, self.cout_i, out, bias=b, n_bias=n_bias, accumulate=accumulate,
                         CoutP=self.CoutP, tapmask_in=self.mask32 if self.kind == "conv" else None,
                         tapmask_out=self.mask_ob if self.kind == "convT" else None)
        return out, 1, Lo

    def backward(self, gy, srcs, N, Hi, Wi, n_bias, need=None, dsrc=None, dacc=None, bias_grad_zeroed=False, **_):
        geom, Lp = self._inner(N, Wi)
        zeroed = bias_grad_zeroed or self._bias_zeroed
        self._bias_zeroed = False
        db = None
        if self.bias is not None:
            if self.kind == "conv":
                if not zeroed:
                    self.bias.grad.zero_()
                db = self.bias.grad.view(-1)
            else:
                if not zeroed:
                    self.bias.grad.zero_()
                    self.dbias_i.zero_()
                db = self.dbias_i
        ops.conv_wgrad(geom, gy, srcs[0], self.cin_i, 0, self.dWp, self.cout_i, self.CoutP, self.Ktot, dbias=db, n_bias=n_bias,
                       tapmask_c32=self.mask32 if self.kind == "conv" else None,
                       tapmask_co32=self.mask32 if self.kind == "convT" else None)
        if need is not None and not need[0]:
            return [None]
        d = dsrc[0] if (dsrc is not None and dsrc[0] is not None) else torch.empty(N * Wi * self.C, device=gy.device)
        gd = ops.conv_geom(N, 1, Lp, 1, Lp, 1, 3, 1, 1, 1, 0)                    # transposed gather of the inner conv
        ops.conv_forward(gd, gy, self.cout_i, self.Wd, self.cin_i, d, accumulate=bool(dacc[0]) if dacc is not None else False,
                         CoutP=pad16(self.cin_i), tapmask_in=self.mask32 if self.kind == "convT" else None,
                         tapmask_out=self.mask_ob if self.kind == "conv" else None)
        return [d]
