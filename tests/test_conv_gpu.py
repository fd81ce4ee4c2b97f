"""GPU parity of the implicit-GEMM convolution kernels (K6/K11) and the
dual-number activations against plain PyTorch fp32 CPU ops (the torch fp32
reference of a floating-point kernel).  Tolerance 1e-5 rel-L2 (fp32 MFMA is an
exact-fp32 fma chain; only the summation order differs)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def cl1(x):   # (N,C,L) -> [N][L][C]
    return x.permute(0, 2, 1).contiguous()


def cl2(x):   # (N,C,H,W) -> [N][H][W][C]
    return x.permute(0, 2, 3, 1).contiguous()


def mk(weight, bias, kind, k, stride, pad, srcC, embC=0, ups=False):
    from sdeflow_light_amd.convnet import ConvOp
    w = torch.nn.Parameter(weight.to(DEV))
    b = torch.nn.Parameter(bias.to(DEV)) if bias is not None else None
    w.grad = torch.zeros_like(w)
    if b is not None:
        b.grad = torch.zeros_like(b)
    op = ConvOp(w, b, kind, k, stride, pad, srcC, embC, ups)
    op.pack()
    op.zero_grad_images()
    return op


@pytest.mark.parametrize("N,Cin,Cout,L,k,s,p", [(3, 32, 32, 100, 3, 1, 1), (2, 1, 32, 64, 3, 1, 1), (2, 64, 128, 37, 3, 1, 1),
                                                 (2, 32, 32, 101, 4, 2, 1), (2, 128, 128, 64, 4, 2, 1), (2, 32, 1, 50, 1, 1, 0),
                                                 (5, 96, 48, 33, 3, 1, 1), (2, 3, 32, 40, 3, 1, 1),
                                                 (2, 32, 32, 300, 3, 1, 1), (3, 64, 64, 513, 3, 1, 1)])   # 256-pixel tiles, ragged
def test_conv1d_fwd_bwd(N, Cin, Cout, L, k, s, p):
    torch.manual_seed(N * 100 + Cin)
    x = torch.randn(N, Cin, L, requires_grad=True)
    W = torch.randn(Cout, Cin, k) * 0.2
    b = torch.randn(Cout) * 0.1
    Wt, bt = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.conv1d(x, Wt, bt, stride=s, padding=p)
    gy = torch.randn_like(y)
    y.backward(gy)
    op = mk(W, b, "conv", (k,), s, p, [Cin])
    xs = cl1(x.detach()).to(DEV)
    out, Ho, Lo = op.forward([xs], N, 1, L, n_bias=N)
    assert (Ho, Lo) == (1, y.shape[-1])
    assert rel_l2(out.view(N, Lo, Cout).cpu(), cl1(y.detach())) <= 1e-5
    (dx,) = op.backward(cl1(gy).to(DEV), [xs], N, 1, L, n_bias=N)
    assert rel_l2(dx.view(N, L, Cin).cpu(), cl1(x.grad)) <= 1e-5
    op.unpack_grads()
    assert rel_l2(op.weight.grad.cpu(), Wt.grad) <= 1e-5
    assert rel_l2(op.bias.grad.cpu(), bt.grad) <= 1e-5
    # tangent rows (second half of the batch) get no bias
    out2, _, _ = op.forward([torch.cat([xs, xs])], 2 * N, 1, L, n_bias=N)
    o2 = out2.view(2 * N, Lo, Cout).cpu()
    assert rel_l2(o2[:N], cl1(y.detach())) <= 1e-5
    assert rel_l2(o2[N:], cl1(F.conv1d(x.detach(), W, None, stride=s, padding=p))) <= 1e-5


@pytest.mark.parametrize("N,Cin,Cout,L", [(2, 128, 128, 16), (3, 64, 32, 25), (2, 128, 64, 64)])
def test_conv_transpose1d_fwd_bwd(N, Cin, Cout, L):
    torch.manual_seed(Cin + L)
    x = torch.randn(N, Cin, L, requires_grad=True)
    W = torch.randn(Cin, Cout, 4) * 0.2
    b = torch.randn(Cout) * 0.1
    Wt, bt = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.conv_transpose1d(x, Wt, bt, stride=2, padding=1)
    gy = torch.randn_like(y)
    y.backward(gy)
    op = mk(W, b, "convT", (4,), 2, 1, [Cin])
    xs = cl1(x.detach()).to(DEV)
    out, _, Lo = op.forward([xs], N, 1, L, n_bias=N)
    assert Lo == 2 * L
    assert rel_l2(out.view(N, Lo, Cout).cpu(), cl1(y.detach())) <= 1e-5
    (dx,) = op.backward(cl1(gy).to(DEV), [xs], N, 1, L, n_bias=N)
    assert rel_l2(dx.view(N, L, Cin).cpu(), cl1(x.grad)) <= 1e-5
    op.unpack_grads()
    assert rel_l2(op.weight.grad.cpu(), Wt.grad) <= 1e-5
    assert rel_l2(op.bias.grad.cpu(), bt.grad) <= 1e-5


def test_conv1d_concat_and_embedding_channels():
    """[up, skip, emb] concat (NNUnet1D.py:175): two real sources + 128 broadcast channels folded into a bias."""
    torch.manual_seed(7)
    N, Ca, Cb, E, Cout, L = 3, 32, 32, 128, 32, 50
    a, bsrc = torch.randn(N, Ca, L, requires_grad=True), torch.randn(N, Cb, L, requires_grad=True)
    emb = torch.randn(N, E, requires_grad=True)
    W = torch.randn(Cout, Ca + Cb + E, 3) * 0.1
    bias = torch.randn(Cout) * 0.1
    Wt, bt = W.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    y = F.conv1d(torch.cat([a, bsrc, emb[:, :, None].expand(-1, -1, L)], 1), Wt, bt, padding=1)
    gy = torch.randn_like(y)
    y.backward(gy)
    op = mk(W, bias, "conv", (3,), 1, 1, [Ca, Cb], embC=E)
    sa, sb_, e = cl1(a.detach()).to(DEV), cl1(bsrc.detach()).to(DEV), emb.detach().to(DEV).contiguous()
    out, _, _ = op.forward([sa, sb_], N, 1, L, n_bias=N, emb=e)
    assert rel_l2(out.view(N, L, Cout).cpu(), cl1(y.detach())) <= 1e-5
    demb = torch.zeros(N * E, device=DEV)
    da, db = op.backward(cl1(gy).to(DEV), [sa, sb_], N, 1, L, n_bias=N, emb=e, demb=demb)
    assert rel_l2(da.view(N, L, Ca).cpu(), cl1(a.grad)) <= 1e-5
    assert rel_l2(db.view(N, L, Cb).cpu(), cl1(bsrc.grad)) <= 1e-5
    assert rel_l2(demb.view(N, E).cpu(), emb.grad) <= 1e-5
    op.unpack_grads()
    assert rel_l2(op.weight.grad.cpu(), Wt.grad) <= 1e-5
    assert rel_l2(op.bias.grad.cpu(), bt.grad) <= 1e-5


@pytest.mark.parametrize("N,Cin,Cout,H,W_,k,s,p", [(2, 3, 32, 16, 16, 3, 1, 1), (2, 32, 32, 16, 12, 3, 2, 1), (1, 96, 64, 8, 8, 3, 1, 1),
                                                    (2, 64, 128, 8, 8, 1, 1, 0), (2, 192, 64, 4, 4, 3, 1, 1), (1, 32, 3, 10, 10, 3, 1, 1),
                                                    (3, 32, 32, 20, 23, 3, 1, 1), (2, 96, 64, 17, 32, 3, 1, 1),   # 16x16 tiles, ragged
                                                    (2, 64, 32, 28, 28, 3, 1, 1)])
def test_conv2d_fwd_bwd(N, Cin, Cout, H, W_, k, s, p):
    torch.manual_seed(Cin + H)
    x = torch.randn(N, Cin, H, W_, requires_grad=True)
    W = torch.randn(Cout, Cin, k, k) * 0.1
    b = torch.randn(Cout) * 0.1
    Wt, bt = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.conv2d(x, Wt, bt, stride=s, padding=p)
    gy = torch.randn_like(y)
    y.backward(gy)
    op = mk(W, b, "conv", (k, k), s, p, [Cin])
    xs = cl2(x.detach()).to(DEV)
    out, Ho, Wo = op.forward([xs], N, H, W_, n_bias=N)
    assert (Ho, Wo) == tuple(y.shape[-2:])
    assert rel_l2(out.view(N, Ho, Wo, Cout).cpu(), cl2(y.detach())) <= 1e-5
    (dx,) = op.backward(cl2(gy).to(DEV), [xs], N, H, W_, n_bias=N)
    assert rel_l2(dx.view(N, H, W_, Cin).cpu(), cl2(x.grad)) <= 1e-5
    op.unpack_grads()
    assert rel_l2(op.weight.grad.cpu(), Wt.grad) <= 1e-5
    assert rel_l2(op.bias.grad.cpu(), bt.grad) <= 1e-5


def test_conv2d_upsample_folded_and_two_sources():
    torch.manual_seed(3)
    N, C, H = 2, 64, 8
    x = torch.randn(N, C, H, H)
    W, b = torch.randn(C, C, 3, 3) * 0.1, torch.randn(C) * 0.1
    y = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), W, b, padding=1)      # model/unet.py:67-69
    op = mk(W, b, "conv", (3, 3), 1, 1, [C], ups=True)
    out, Ho, Wo = op.forward([cl2(x).to(DEV)], N, H, H, n_bias=N)
    assert (Ho, Wo) == (16, 16) and rel_l2(out.view(N, 16, 16, C).cpu(), cl2(y)) <= 1e-5
    a, c = torch.randn(N, 128, H, H), torch.randn(N, 64, H, H)
    W2, b2 = torch.randn(128, 192, 3, 3) * 0.05, torch.randn(128) * 0.1
    y2 = F.conv2d(torch.cat([a, c], 1), W2, b2, padding=1)                               # model/unet.py:514
    op2 = mk(W2, b2, "conv", (3, 3), 1, 1, [128, 64])
    out2, _, _ = op2.forward([cl2(a).to(DEV), cl2(c).to(DEV)], N, H, H, n_bias=N)
    assert rel_l2(out2.view(N, H, H, 128).cpu(), cl2(y2)) <= 1e-5


def test_linear_as_1x1():
    torch.manual_seed(4)
    x = torch.randn(37, 1, requires_grad=True)
    W, b = torch.randn(128, 1), torch.randn(128)
    Wt = W.clone().requires_grad_(True)
    y = F.linear(x, Wt, b)
    gy = torch.randn_like(y)
    y.backward(gy)
    op = mk(W, b, "linear", (1,), 1, 0, [1])
    out, _, _ = op.forward([x.detach().to(DEV).contiguous()], 37, 1, 1, n_bias=37)
    assert rel_l2(out.view(37, 128).cpu(), y.detach()) <= 1e-5
    (dx,) = op.backward(gy.to(DEV).contiguous(), [x.detach().to(DEV).contiguous()], 37, 1, 1, n_bias=37)
    op.unpack_grads()
    assert rel_l2(dx.view(37, 1).cpu(), x.grad) <= 1e-5 and rel_l2(op.weight.grad.cpu(), Wt.grad) <= 1e-5


@pytest.mark.parametrize("act,fn", [(0, lambda z: F.gelu(z)), (1, lambda z: torch.sigmoid(z) * z)])
def test_act_dual_forward_backward(act, fn):
    from sdeflow_light_amd import ops
    torch.manual_seed(5)
    n = 4096
    zp, zt = torch.randn(n) * 2, torch.randn(n)
    hp, ht = torch.func.jvp(fn, (zp,), (zt,))
    z = torch.cat([zp, zt]).to(DEV)
    h = ops.act_dual_forward(act, z, torch.empty_like(z), dual=True)
    assert rel_l2(h[:n].cpu(), hp) <= 1e-6 and rel_l2(h[n:].cpu(), ht) <= 1e-5
    h1 = ops.act_dual_forward(act, z[:n].contiguous(), torch.empty(n, device=DEV), dual=False)
    assert rel_l2(h1.cpu(), hp) <= 1e-6
    gp, gt = torch.randn(n), torch.randn(n)
    zp_, zt_ = zp.clone().requires_grad_(True), zt.clone().requires_grad_(True)
    hp2, ht2 = torch.func.jvp(fn, (zp_,), (zt_,))
    (hp2 * gp).sum().add((ht2 * gt).sum()).backward()
    g = torch.cat([gp, gt]).to(DEV)
    ops.act_dual_backward(act, z, g)
    assert rel_l2(g[:n].cpu(), zp_.grad) <= 1e-5 and rel_l2(g[n:].cpu(), zt_.grad) <= 1e-5


def test_colsum_rows():
    from sdeflow_light_amd import ops
    torch.manual_seed(6)
    for (N, P, C) in ((3, 100, 32), (2, 7, 300), (5, 1, 16), (1, 1000, 1)):
        x = torch.randn(N, P, C)
        assert rel_l2(ops.colsum(x.to(DEV), N, P, C).cpu(), x.sum(1)) <= 1e-5
        assert torch.equal(ops.gather_row(x.to(DEV), N, P, C, P - 1).cpu(), x[:, P - 1])
        xd = x.to(DEV).clone()
        E = torch.randn(N, C)
        ops.add_row(xd, E.to(DEV), N, P, C, 0, -1.0)
        ref = x.clone(); ref[:, 0] -= E
        assert rel_l2(xd.cpu(), ref) <= 1e-6


@pytest.mark.parametrize("N,C0,C1,Cout,H,W_,k,act,ups", [(3, 32, 0, 32, 16, 16, 3, 1, False), (2, 64, 32, 64, 8, 16, 3, 1, False),
                                                         (2, 32, 0, 96, 1, 128, 1, 0, False), (2, 64, 0, 64, 8, 8, 3, 1, True)])
def test_conv_fused_groupnorm_silu_input_and_residual(N, C0, C1, Cout, H, W_, k, act, ups):
    """msgm_conv_forward_fused: GroupNorm(+SiLU) folded into the input staging (two sources = channel concat, folded
    2x upsample) and the residual added in the epilogue, vs GroupNorm -> SiLU -> (upsample) -> conv -> + residual in
    plain PyTorch fp32.  1e-5 rel-L2."""
    from sdeflow_light_amd import ops
    torch.manual_seed(C0 + C1 + H)
    C = C0 + C1
    G = min(C, 32)
    x = torch.randn(N, C, H, W_) * 1.5 + 0.3
    gam, bet = torch.randn(C) * 0.5 + 1.0, torch.randn(C) * 0.2
    w, b = torch.randn(Cout, C, k if H > 1 else 1, k) * 0.1, torch.randn(Cout) * 0.1
    Ho, Wo = (2 * H, 2 * W_) if ups else (H, W_)
    res = torch.randn(N, Cout, Ho, Wo)
    h = F.group_norm(x, G, gam, bet, 1e-5)
    if act:
        h = F.silu(h)
    if ups:
        h = F.interpolate(h, scale_factor=2, mode="nearest")
    ref = F.conv2d(h, w, b, 1, ((k - 1) // 2 if H > 1 else 0, (k - 1) // 2)) + res
    kk = (k, k) if H > 1 else (k,)
    op = mk(w.reshape(Cout, C, -1) if H == 1 else w, b, "conv", kk, 1, (k - 1) // 2, [C0, C1] if C1 else [C0], ups=ups)
    assert op.can_transform_input(N, H, W_)
    x0 = cl2(x[:, :C0]).reshape(-1).to(DEV)
    x1 = cl2(x[:, C0:]).reshape(-1).to(DEV) if C1 else None
    sc, sh = ops.groupnorm_affine(x0, C0, gam.to(DEV), bet.to(DEV), N, H * W_, G, x1=x1, C1=C1)
    xh = (x - x.reshape(N, G, -1).mean(2).repeat_interleave(C // G, 1)[:, :, None, None])       # check the affine itself
    inv = 1.0 / torch.sqrt(x.reshape(N, G, -1).var(2, unbiased=False) + 1e-5).repeat_interleave(C // G, 1)
    assert rel_l2(sc.view(N, C).cpu(), inv * gam) <= 1e-5
    out, ho, wo = op.forward([x0] + ([x1] if C1 else []), N, H, W_, N, in_affine=(sc, sh), in_act=act,
                             residual=cl2(res).reshape(-1).to(DEV))
    assert (ho, wo) == (Ho, Wo)
    assert rel_l2(out.view(N, Ho, Wo, Cout).cpu(), cl2(ref)) <= 1e-5


def test_conv_input_transform_refused_where_not_built():
    from sdeflow_light_amd import ops
    from sdeflow_light_amd._lib import MsgmError
    w, b = torch.randn(32, 32, 3, 3) * 0.1, None
    op = mk(w, b, "conv", (3, 3), 2, 1, [32])                       # strided: not a halo-tile shape
    assert not op.can_transform_input(2, 16, 16)
    x = torch.randn(2 * 16 * 16 * 32, device=DEV)
    ab = (torch.ones(2 * 32, device=DEV), torch.zeros(2 * 32, device=DEV))
    with pytest.raises(MsgmError):
        op.forward([x], 2, 16, 16, 2, in_affine=ab)
    out, _, _ = op.forward([x], 2, 16, 16, 2, residual=torch.ones(2 * 8 * 8 * 32, device=DEV))    # residual works everywhere
    ref, _, _ = op.forward([x], 2, 16, 16, 2)
    assert rel_l2(out.cpu(), (ref + 1.0).cpu()) <= 1e-6


@pytest.mark.parametrize("kind,N,Cin,Cout,L", [("conv", 3, 32, 32, 128), ("conv", 2, 64, 64, 256), ("conv", 2, 128, 128, 34),
                                               ("convT", 3, 128, 128, 128), ("convT", 2, 64, 32, 256), ("convT", 2, 32, 64, 17)])
def test_stride2_pair_op_fwd_bwd(kind, N, Cin, Cout, L):
    """Downsample / Upsample of the 1-D U-Net (k=4, s=2, p=1; NNUnet1D.py:84,98) re-expressed as 3-tap stride-1 convs
    over position pairs (convnet.Stride2PairOp) vs plain PyTorch fp32: forward (incl. tangent rows without bias),
    dgrad, wgrad and the bias gradient.  Small L exercises the generic kernels, large L the LDS-tiled ones."""
    from sdeflow_light_amd.convnet import Stride2PairOp
    torch.manual_seed(Cin + Cout + L)
    x = torch.randn(N, Cin, L, requires_grad=True)
    W = torch.randn(*((Cout, Cin, 4) if kind == "conv" else (Cin, Cout, 4))) * 0.2
    b = torch.randn(Cout) * 0.1
    Wt, bt = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    f = (lambda xx, ww, bb: F.conv1d(xx, ww, bb, stride=2, padding=1)) if kind == "conv" else \
        (lambda xx, ww, bb: F.conv_transpose1d(xx, ww, bb, stride=2, padding=1))
    y = f(x, Wt, bt)
    gy = torch.randn_like(y)
    y.backward(gy)
    w, bp = torch.nn.Parameter(W.to(DEV)), torch.nn.Parameter(b.to(DEV))
    w.grad, bp.grad = torch.zeros_like(w), torch.zeros_like(bp)
    op = Stride2PairOp(w, bp, kind)
    op.pack()
    op.zero_grad_images()
    xs = cl1(x.detach()).to(DEV).reshape(-1)
    out, _, Lo = op.forward([xs], N, 1, L, n_bias=N)
    assert Lo == y.shape[-1]
    assert rel_l2(out.view(N, Lo, Cout).cpu(), cl1(y.detach())) <= 1e-5
    (dx,) = op.backward(cl1(gy).to(DEV).reshape(-1), [xs], N, 1, L, n_bias=N)
    assert rel_l2(dx.view(N, L, Cin).cpu(), cl1(x.grad)) <= 1e-5
    op.unpack_grads()
    assert rel_l2(op.weight.grad.cpu(), Wt.grad) <= 1e-5
    assert rel_l2(op.bias.grad.cpu(), bt.grad) <= 1e-5
    out2, _, _ = op.forward([torch.cat([xs, xs])], 2 * N, 1, L, n_bias=N)          # tangent rows: no bias
    o2 = out2.view(2 * N, Lo, Cout).cpu()
    assert rel_l2(o2[:N], cl1(y.detach())) <= 1e-5
    assert rel_l2(o2[N:], cl1(f(x.detach(), W, None))) <= 1e-5


@pytest.mark.parametrize("N,H,Ci,Co,two,ups,aff", [(3, 32, 64, 64, False, False, False), (2, 64, 32, 32, False, False, True),
                                                   (2, 32, 64, 32, True, False, True), (2, 16, 128, 128, False, False, False),
                                                   (2, 16, 64, 64, False, True, False), (1, 16, 32, 96, False, False, True),
                                                   # the LDS-weight form (>= 64 input channels) with ragged 16-channel tails
                                                   (2, 16, 48, 64, True, False, True), (1, 32, 80, 32, False, False, False),
                                                   (2, 16, 144, 96, True, True, True),
                                                   # >= 2048 tiles of a 32 -> 32 layer: the persistent form (weights resident in LDS), with a
                                                   # tile count that does not divide over the workgroups, and with the folded upsample
                                                   (130, 64, 32, 32, False, False, True), (520, 32, 32, 32, False, True, False)])
def test_winograd_forward_equals_direct_conv(N, H, Ci, Co, two, ups, aff):
    """Winograd F(2x2,3x3) forward (sampler path) vs the direct halo-tile kernel on the same inputs and fused options:
    second source, folded 2x upsample, GroupNorm(+SiLU) input transform, bias, per-sample bias, residual.  Both are fp32;
    they differ by the rounding of the transforms only."""
    from sdeflow_light_amd import ops
    from sdeflow_light_amd.convnet import ConvOp
    torch.manual_seed(N * 1000 + H + Ci + Co)
    dev = "cuda"
    C0, C1 = (Ci, 32) if two else (Ci, 0)
    w = torch.nn.Parameter(torch.randn(Co, C0 + C1, 3, 3, device=dev) * (2.0 / (9 * (C0 + C1))) ** 0.5)
    b = torch.nn.Parameter(torch.randn(Co, device=dev) * 0.1)
    op = ConvOp(w, b, "conv", (3, 3), 1, 1, [C0, C1] if two else [C0], ups=ups)
    op.pack()
    assert op.wino_capable()
    tab = ops.PackTable(op.wino_jobs(), dev)
    tab.run_wino()
    Hi = H // 2 if ups else H
    srcs = [torch.randn(N * Hi * Hi * C0, device=dev)] + ([torch.randn(N * Hi * Hi * C1, device=dev)] if two else [])
    sb = torch.randn(N * Co, device=dev) * 0.1
    res = torch.randn(N * H * H * Co, device=dev)
    in_aff = (1 + 0.3 * torch.randn(N * (C0 + C1), device=dev), 0.2 * torch.randn(N * (C0 + C1), device=dev)) if aff else None
    kw = dict(samp_bias=sb, residual=res, in_affine=in_aff, in_act=1 if aff else 0)
    ref, _, _ = op.forward(srcs, N, Hi, Hi, N, **kw)
    got, _, _ = op.forward(srcs, N, Hi, Hi, N, wino=True, stats=True, **kw)
    e = rel_l2(got.cpu(), ref.cpu())
    print(f"Winograd vs direct 3x3 conv N={N} {H}x{H} {C0}+{C1}->{Co} ups={ups} affine={aff}: rel-L2 {e:.2e}")
    assert e <= 2e-6
    assert not torch.equal(got, ref)                       # it really took the other kernel
    cs, S = got._msgm_cs                                   # its statistics by-product: one slot per (16x16 tile, wave)
    assert S == (H // 16) ** 2 * 4
    o = got.view(N, H * H, Co).double()
    tot = cs.view(N, S, 2, Co).double().sum(1)
    assert rel_l2(tot[:, 0].cpu(), o.sum(1).cpu()) <= 2e-6 and rel_l2(tot[:, 1].cpu(), (o * o).sum(1).cpu()) <= 2e-6


# ------------------------------------------------------------------ channel statistics as a by-product of the epilogue
@pytest.mark.parametrize("N,Ci,C1,Co,H,W_,k,st,res,acc", [
    (24, 64, 0, 64, 32, 32, 3, 1, True, False),     # 16x16 tiles, 4 channel tiles per wave; residual in the epilogue
    (3, 32, 0, 32, 32, 32, 3, 1, False, True),      # small launch: 8x16 tiles; accumulate onto a skip-conv result
    (20, 64, 32, 32, 64, 64, 3, 1, False, False),   # two sources (decoder), 32 output channels
    (5, 64, 0, 64, 1, 1024, 1, 1, True, False),     # pixel-stationary 1x1 (attention proj_out + residual), PT = 4
    (18, 128, 0, 128, 1, 256, 1, 1, True, False),   # 1x1 with 8 input groups, PT = 2
    (2, 32, 0, 32, 1, 320, 3, 1, False, False),     # 1-D 3-tap, ragged last tile
    (6, 64, 0, 64, 32, 32, 3, 2, False, False),     # Downsample (stride 2): implicit GEMM from L2, NT = 2
    (4, 3, 0, 32, 64, 64, 3, 1, False, False),      # the input convolution (3 channels): implicit GEMM, NT = 4
])
def test_conv_channel_statistics_byproduct(N, Ci, C1, Co, H, W_, k, st, res, acc):
    """msgm_conv_fuse_t.chanstats: the per-(sample, slot, channel) sums the conv epilogue leaves behind add up to the
    sums of the FINAL output (bias, accumulate and residual included), and msgm_groupnorm_affine_chanstats gives the
    (scale, shift) that msgm_groupnorm_affine computes by reading the tensor (model/unet.py:140-143: every GroupNorm
    reads a conv output)."""
    from sdeflow_light_amd import ops
    torch.manual_seed(N + Ci + Co)
    kk = (k, k) if H > 1 else (k,)
    W = torch.randn(Co, Ci + C1, *kk) * 0.1
    b = torch.randn(Co) * 0.3 + 0.5
    op = mk(W, b, "conv", kk, st, (k - 1) // 2, [Ci] + ([C1] if C1 else []))
    srcs = [torch.randn(N * H * W_ * Ci, device=DEV) + 0.3] + ([torch.randn(N * H * W_ * C1, device=DEV)] if C1 else [])
    geom, Ho, Wo = op._geom(N, H, W_)
    S = ops.conv_chanstats_slots(geom, Ci, C1, Co, op.CoutP)
    assert S > 0
    P = Ho * Wo
    r = torch.randn(N * P * Co, device=DEV) if res else None
    out = torch.randn(N * P * Co, device=DEV) if acc else None
    out, _, _ = op.forward(srcs, N, H, W_, n_bias=N, residual=r, out=out, accumulate=acc, stats=True)
    cs, S2 = out._msgm_cs
    assert S2 == S and cs.numel() == N * S * 2 * Co
    o = out.view(N, P, Co).double()
    tot = cs.view(N, S, 2, Co).double().sum(1)
    e1, e2 = rel_l2(tot[:, 0].cpu(), o.sum(1).cpu()), rel_l2(tot[:, 1].cpu(), (o * o).sum(1).cpu())
    print(f"chanstats N={N} {Ci}+{C1}->{Co} {H}x{W_} k={k} stride {st} S={S}: sum rel-L2 {e1:.2e}, sum of squares {e2:.2e}")
    assert e1 <= 2e-6 and e2 <= 2e-6
    G = 32
    gamma, beta = torch.randn(Co, device=DEV), torch.randn(Co, device=DEV)
    sc0, sh0 = ops.groupnorm_affine(out, Co, gamma, beta, N, P, G)
    sc1, sh1 = ops.groupnorm_affine_cs(cs, S, Co, gamma, beta, N, P, G)
    e3, e4 = rel_l2(sc1.cpu(), sc0.cpu()), rel_l2(sh1.cpu(), sh0.cpu())
    print(f"   GroupNorm affine from chanstats vs from the tensor: scale {e3:.2e}, shift {e4:.2e}")
    assert e3 <= 2e-6 and e4 <= 2e-6
    # a second forward without stats clears the description of the overwritten values
    out2, _, _ = op.forward(srcs, N, H, W_, n_bias=N, out=out)
    assert out2._msgm_cs is None


def test_groupnorm_affine_chanstats_two_sources():
    """The decoder's GroupNorm over cat([h, skip]) from the two producers' statistics (different slot counts, a group
    straddling the two sources: 128 + 64 channels in 32 groups of 6)."""
    from sdeflow_light_amd import ops
    torch.manual_seed(4)
    N, H = 6, 32
    P = H * H
    outs = []
    for Ci, Co, k in ((64, 128, 1), (32, 64, 3)):
        kk = (k, k) if k == 3 else (1,)
        op = mk(torch.randn(Co, Ci, *kk) * 0.1, torch.randn(Co), "conv", kk, 1, (k - 1) // 2, [Ci])
        hw = (H, H) if k == 3 else (1, P)
        o, _, _ = op.forward([torch.randn(N * P * Ci, device=DEV)], N, hw[0], hw[1], n_bias=N, stats=True)
        assert o._msgm_cs is not None
        outs.append((o, Co))
    (h, C0), (s, C1) = outs
    assert h._msgm_cs[1] != s._msgm_cs[1]
    gamma, beta = torch.randn(C0 + C1, device=DEV), torch.randn(C0 + C1, device=DEV)
    sc0, sh0 = ops.groupnorm_affine(h, C0, gamma, beta, N, P, 32, x1=s, C1=C1)
    sc1, sh1 = ops.groupnorm_affine_cs(h._msgm_cs[0], h._msgm_cs[1], C0, gamma, beta, N, P, 32, cs1=s._msgm_cs[0],
                                       S1=s._msgm_cs[1], C1=C1)
    e3, e4 = rel_l2(sc1.cpu(), sc0.cpu()), rel_l2(sh1.cpu(), sh0.cpu())
    print(f"two-source GroupNorm affine from chanstats vs from the tensors: scale {e3:.2e}, shift {e4:.2e}")
    assert e3 <= 2e-6 and e4 <= 2e-6


# ------------------------------------------------------------------ the U-Net's input / output convolutions (vector-ALU kernels)
@pytest.mark.parametrize("N,nb,Cin,H,W_", [(5, 3, 3, 32, 32), (4, 4, 1, 20, 24), (3, 2, 3, 64, 64)])
def test_unet_input_conv_small_cin(N, nb, Cin, H, W_):
    """model/unet.py:353-359, conv3x3(image channels -> 32) on k_conv3x3_cin_small vs F.conv2d (bias on the first nb rows
    only: the tangent rows of the dual batch carry none); with MSGM_NO_CONV_SMALL the MFMA implicit GEMM served it."""
    torch.manual_seed(N + Cin + H)
    x = torch.randn(N, Cin, H, W_)
    w, b = torch.randn(32, Cin, 3, 3) * 0.3, torch.randn(32)
    ref = F.conv2d(x, w, None, 1, 1)
    ref[:nb] += b[None, :, None, None]
    op = mk(w, b, "conv", (3, 3), 1, 1, [Cin])
    out, _, _ = op.forward([cl2(x).reshape(-1).to(DEV)], N, H, W_, n_bias=nb, stats=True)
    e = rel_l2(out.view(N, H, W_, 32).cpu(), cl2(ref))
    print(f"input conv {Cin}->32 {H}x{W_}: rel-L2 {e:.2e}")
    assert e <= 2e-7
    if (H * W_) % 64 == 0:                                  # statistics by-product: one slot per 64 pixels
        cs, S = out._msgm_cs
        assert S == H * W_ // 64
        o = out.view(N, H * W_, 32).double()
        tot = cs.view(N, S, 2, 32).double().sum(1)
        assert rel_l2(tot[:, 0].cpu(), o.sum(1).cpu()) <= 2e-6 and rel_l2(tot[:, 1].cpu(), (o * o).sum(1).cpu()) <= 2e-6
    else:
        assert out._msgm_cs is None
    r = torch.randn(N * H * W_ * 32, device=DEV)
    out2, _, _ = op.forward([cl2(x).reshape(-1).to(DEV)], N, H, W_, n_bias=nb, residual=r)
    assert rel_l2((out2 - r).cpu(), out.cpu()) <= 1e-6


@pytest.mark.parametrize("N,nb,Cout,H,W_,aff", [(5, 3, 3, 32, 32, True), (4, 4, 1, 20, 24, True), (3, 3, 3, 64, 64, False)])
def test_unet_output_conv_small_cout(N, nb, Cout, H, W_, aff):
    """model/unet.py:442-446, GroupNorm -> SiLU -> conv3x3(32 -> image channels) on k_conv3x3_cout_small (GroupNorm + SiLU
    applied while the halo tile is staged) vs plain PyTorch fp32."""
    from sdeflow_light_amd import ops
    torch.manual_seed(N + Cout + H)
    x = torch.randn(N, 32, H, W_) * 1.5 + 0.3
    gam, bet = torch.randn(32) * 0.5 + 1.0, torch.randn(32) * 0.2
    w, b = torch.randn(Cout, 32, 3, 3) * 0.1, torch.randn(Cout)
    h = F.silu(F.group_norm(x, 32, gam, bet, 1e-5)) if aff else x
    ref = F.conv2d(h, w, None, 1, 1)
    ref[:nb] += b[None, :, None, None]
    op = mk(w, b, "conv", (3, 3), 1, 1, [32])
    assert op.can_transform_input(N, H, W_)
    xs = cl2(x).reshape(-1).to(DEV)
    ab = ops.groupnorm_affine(xs, 32, gam.to(DEV), bet.to(DEV), N, H * W_, 32) if aff else None
    out, _, _ = op.forward([xs], N, H, W_, n_bias=nb, in_affine=ab, in_act=1 if aff else 0)
    e = rel_l2(out.view(N, H, W_, Cout).cpu(), cl2(ref))
    print(f"output conv 32->{Cout} {H}x{W_} (GroupNorm+SiLU folded: {aff}): rel-L2 {e:.2e}")
    assert e <= 1e-6
    out2 = torch.ones(N * H * W_ * Cout, device=DEV)
    op.forward([xs], N, H, W_, n_bias=nb, in_affine=ab, in_act=1 if aff else 0, out=out2, accumulate=True)
    assert rel_l2((out2 - 1.0).cpu(), out.cpu()) <= 1e-6


@pytest.mark.parametrize("N,C0,C1,Cout,T,aff,extra", [(20, 64, 0, 192, 256, True, None),        # qkv with the folded GroupNorm
                                                        (6, 64, 0, 64, 1024, False, "residual"),   # proj_out + x
                                                        (24, 128, 64, 128, 256, False, None),      # decoder skip conv, two sources
                                                        (10, 128, 0, 384, 512, True, "accumulate"),
                                                        (3, 32, 0, 32, 4096, False, "residual"),
                                                        (20, 192, 0, 64, 272, False, None),        # 192 channels, 16 pixels per wave (T % 32 != 0)
                                                        (16, 192, 0, 64, 256, True, "residual")])  # ... and 32 per wave
def test_pixel_stationary_1x1_kernel(N, C0, C1, Cout, T, aff, extra):
    """k_conv1x1 (1x1 convolutions with >= 4096 pixels: qkv / proj_out / skip convs, model/unet.py:216-232,158) with its
    fused options — folded GroupNorm(+SiLU) on the input, second source, residual / accumulate in the epilogue, per-sample
    bias on the primal rows — against plain PyTorch fp32, and against the halo-tile kernel that served these shapes
    before (MSGM_NO_CONV1X1 is read once per process, so that comparison lives in tools/bench_1x1.py)."""
    from sdeflow_light_amd import ops
    torch.manual_seed(N + C0 + Cout)
    C = C0 + C1
    x = torch.randn(N, C, T) * 1.3 + 0.2
    w, b = torch.randn(Cout, C, 1) * 0.1, torch.randn(Cout)
    nb = N - 2                                               # the last two rows play tangent rows: no bias
    sb = torch.randn(nb, Cout)
    h = x
    ab = None
    if aff:
        G = 32
        gam, bet = torch.randn(C) * 0.5 + 1.0, torch.randn(C) * 0.2
        h = F.silu(F.group_norm(x, G, gam, bet, 1e-5))
    ref = F.conv1d(h, w, None)
    ref[:nb] += (b[None, :] + sb)[:, :, None]
    op = mk(w, b, "conv", (1,), 1, 0, [C0] + ([C1] if C1 else []))
    x0 = cl1(x[:, :C0]).reshape(-1).to(DEV)
    x1 = cl1(x[:, C0:]).reshape(-1).to(DEV) if C1 else None
    if aff:
        ab = ops.groupnorm_affine(x0, C0, gam.to(DEV), bet.to(DEV), N, T, G, x1=x1, C1=C1)
    srcs = [x0] + ([x1] if C1 else [])
    kw = dict(n_bias=nb, samp_bias=sb.reshape(-1).to(DEV), emb_rows=nb, in_affine=ab, in_act=1 if aff else 0)
    if extra == "residual":
        r = torch.randn(N, Cout, T)
        out, _, _ = op.forward(srcs, N, 1, T, residual=cl1(r).reshape(-1).to(DEV), stats=True, **kw)
        ref = ref + r
    elif extra == "accumulate":
        r = torch.randn(N, Cout, T)
        out = cl1(r).reshape(-1).to(DEV).clone()
        op.forward(srcs, N, 1, T, out=out, accumulate=True, stats=True, **kw)
        ref = ref + r
    else:
        out, _, _ = op.forward(srcs, N, 1, T, stats=True, **kw)
    e = rel_l2(out.view(N, T, Cout).cpu(), cl1(ref))
    print(f"1x1 {C0}+{C1}->{Cout} T={T} N={N} affine={aff} {extra}: rel-L2 {e:.2e}")
    assert e <= 4e-7                                         # measured 0.6-1.5e-7
    cs, S = out._msgm_cs                                     # and the statistics by-product describes the final values
    o = out.view(N, T, Cout).double()
    tot = cs.view(N, S, 2, Cout).double().sum(1)
    assert rel_l2(tot[:, 0].cpu(), o.sum(1).cpu()) <= 2e-6 and rel_l2(tot[:, 1].cpu(), (o * o).sum(1).cpu()) <= 2e-6


# every (MT, NT, WM) instance of the pixel-streaming 1x1 wgrad (k_wgrad1x1), 1-D and 2-D pixel grids, pixel counts that are
# not a multiple of the staged tile, primal rows < N (the bias gradient stops at n_bias), both the slot-ordered
# (deterministic, default) and the float-atomic form, accumulation into a non-zero packed image at a K offset
@pytest.mark.parametrize("N,nb,C,Cout,H,W_", [(3, 2, 32, 32, 8, 8), (2, 1, 64, 32, 16, 16), (2, 2, 128, 32, 8, 8),
                                              (3, 2, 32, 64, 1, 80), (2, 1, 32, 128, 8, 8), (2, 1, 32, 192, 4, 4),
                                              (5, 3, 64, 64, 1, 48), (2, 1, 64, 128, 8, 8), (3, 2, 64, 192, 1, 272),
                                              (2, 1, 128, 64, 8, 8), (3, 2, 128, 128, 16, 16), (2, 1, 128, 384, 1, 256),
                                              (2, 1, 256, 64, 8, 8), (2, 2, 256, 128, 8, 8)])
@pytest.mark.parametrize("atomic", [False, True])
def test_wgrad_1x1_pixel_streaming(N, nb, C, Cout, H, W_, atomic, monkeypatch):
    from sdeflow_light_amd import ops
    if atomic:
        monkeypatch.setenv("MSGM_ATOMIC_WGRAD", "1")
    torch.manual_seed(C + Cout + H)
    P = H * W_
    gy = torch.randn(N * P, Cout)
    x = torch.randn(N * P, C)
    koff, Ktot = 16, C + 32                                   # this source sits at K offset 16 of a wider packed image
    geom = ops.conv_geom(N, H, W_, H, W_, 1, 1, 1, 0, 0, 0)
    dWp0 = torch.randn(Cout * Ktot)
    db0 = torch.randn(Cout)
    dWp, db = dWp0.clone().to(DEV), db0.clone().to(DEV)
    ops.conv_wgrad(geom, gy.to(DEV), x.to(DEV), C, koff, dWp, Cout, Cout, Ktot, dbias=db, n_bias=nb)
    want = dWp0.view(Cout, Ktot).double().clone()
    want[:, koff:koff + C] += gy.double().t() @ x.double()
    assert rel_l2(dWp.view(Cout, Ktot).cpu().double(), want) <= 1e-6
    # columns outside this source's K range are untouched
    got = dWp.view(Cout, Ktot).cpu()
    assert torch.equal(got[:, :koff], dWp0.view(Cout, Ktot)[:, :koff]) and torch.equal(got[:, koff + C:], dWp0.view(Cout, Ktot)[:, koff + C:])
    wb = db0.double() + gy[: nb * P].double().sum(0)
    assert rel_l2(db.cpu().double(), wb) <= 1e-6
    if not atomic:                                            # slot order: the same bits every time
        dWp2, db2 = dWp0.clone().to(DEV), db0.clone().to(DEV)
        ops.conv_wgrad(geom, gy.to(DEV), x.to(DEV), C, koff, dWp2, Cout, Cout, Ktot, dbias=db2, n_bias=nb)
        assert torch.equal(dWp2, dWp) and torch.equal(db2, db)


# opt-in experiment: the sampler's 3x3 convolution in bf16-SPLIT arithmetic (three bf16 pieces per fp32 operand,
# six bf16 MFMA products, fp32 accumulate) against the fp32 direct kernel and a float64 convolution: it must be fp32-grade
@pytest.mark.parametrize("N,H,Ci,Co,two,ups,aff", [(3, 32, 64, 64, False, False, False), (2, 64, 32, 32, False, False, True),
                                                   (2, 32, 64, 32, True, False, True), (2, 16, 128, 128, False, False, False),
                                                   (2, 16, 64, 64, False, True, False), (1, 16, 32, 96, False, False, True),
                                                   (2, 16, 128, 96, True, True, True)])
def test_bf16_split_conv_is_fp32_grade(N, H, Ci, Co, two, ups, aff):
    from sdeflow_light_amd import ops
    from sdeflow_light_amd.convnet import ConvOp
    torch.manual_seed(N * 1000 + H + Ci + Co)
    dev = "cuda"
    C0, C1 = (Ci, 32) if two else (Ci, 0)
    w = torch.nn.Parameter(torch.randn(Co, C0 + C1, 3, 3, device=dev) * (2.0 / (9 * (C0 + C1))) ** 0.5)
    b = torch.nn.Parameter(torch.randn(Co, device=dev) * 0.1)
    op = ConvOp(w, b, "conv", (3, 3), 1, 1, [C0, C1] if two else [C0], ups=ups)
    op.pack()
    assert op.b6_capable()
    op.pack_b6()
    Hi = H // 2 if ups else H
    srcs = [torch.randn(N * Hi * Hi * C0, device=dev)] + ([torch.randn(N * Hi * Hi * C1, device=dev)] if two else [])
    sb = torch.randn(N * Co, device=dev) * 0.1
    res = torch.randn(N * H * H * Co, device=dev)
    in_aff = (1 + 0.3 * torch.randn(N * (C0 + C1), device=dev), 0.2 * torch.randn(N * (C0 + C1), device=dev)) if aff else None
    kw = dict(samp_bias=sb, residual=res, in_affine=in_aff, in_act=1 if aff else 0)
    ref, _, _ = op.forward(srcs, N, Hi, Hi, N, **kw)
    got, _, _ = op.forward(srcs, N, Hi, Hi, N, b6=True, stats=True, **kw)
    assert not torch.equal(got, ref)                       # it really took the other kernel
    # float64 yardstick: the same convolution on the CPU in double precision
    xs = [s_.view(N, Hi, Hi, c).permute(0, 3, 1, 2).double().cpu() for s_, c in zip(srcs, [C0, C1] if two else [C0])]
    xin = torch.cat(xs, 1)
    if aff:
        a_, b_ = in_aff[0].view(N, -1, 1, 1).double().cpu(), in_aff[1].view(N, -1, 1, 1).double().cpu()
        xin = torch.nn.functional.silu(xin * a_ + b_)
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    y64 = F.conv2d(xin, w.detach().double().cpu(), b.detach().double().cpu(), padding=1) + sb.view(N, Co, 1, 1).double().cpu()
    y64 = cl2(y64) + res.view(N, H, H, Co).double().cpu()
    e_ref, e_got = rel_l2(ref.view(N, H, H, Co).cpu().double(), y64), rel_l2(got.view(N, H, H, Co).cpu().double(), y64)
    print(f"bf16-split 3x3 conv N={N} {H}x{H} {C0}+{C1}->{Co} ups={ups} affine={aff}: vs float64: fp32 direct {e_ref:.2e}, bf16 split {e_got:.2e}; "
          f"split vs direct {rel_l2(got.cpu(), ref.cpu()):.2e}")
    assert e_got <= max(2.0 * e_ref, 3e-7)                 # as accurate as the fp32 MFMA kernel (the folded SiLU dominates both)
    cs, S = got._msgm_cs                                   # statistics by-product: one slot per (16x16 tile, wave)
    assert S == (H // 16) ** 2 * 4
    o = got.view(N, H * H, Co).double()
    tot = cs.view(N, S, 2, Co).double().sum(1)
    assert rel_l2(tot[:, 0].cpu(), o.sum(1).cpu()) <= 2e-6 and rel_l2(tot[:, 1].cpu(), (o * o).sum(1).cpu()) <= 2e-6


@pytest.mark.parametrize("N,H,C0,C1,Co,ups", [(3, 32, 64, 0, 64, False), (2, 16, 128, 64, 128, False), (2, 64, 32, 0, 32, False),
                                              (2, 16, 64, 0, 64, True), (2, 32, 64, 32, 32, False),
                                              (128, 64, 32, 0, 32, False)])        # the persistent 32-channel form, forward and dgrad
def test_winograd_training_forward_and_dgrad(N, H, C0, C1, Co, ups):
    """MSGM_TRAIN_WINO (ConvOpSet.pack_wino(train=True)): forward AND dgrad of a 3x3 stride-1 convolution on the Winograd
    kernel — the dgrad as a Winograd forward of gy with the flipped, transposed kernels — against the direct kernels of the same
    op and against PyTorch fp32; the weight gradient path is untouched."""
    from sdeflow_light_amd.convnet import ConvOp, ConvOpSet
    torch.manual_seed(N + H + C0 + Co)
    dev = "cuda"
    w = torch.nn.Parameter(torch.randn(Co, C0 + C1, 3, 3, device=dev) * (2.0 / (9 * (C0 + C1))) ** 0.5)
    b = torch.nn.Parameter(torch.randn(Co, device=dev) * 0.1)
    w.grad, b.grad = torch.zeros_like(w), torch.zeros_like(b)
    op = ConvOp(w, b, "conv", (3, 3), 1, 1, [C0, C1] if C1 else [C0], ups=ups)
    cs = ConvOpSet([op])
    Hi = H // 2 if ups else H
    srcs = [torch.randn(N * Hi * Hi * C0, device=dev)] + ([torch.randn(N * Hi * Hi * C1, device=dev)] if C1 else [])
    gy = torch.randn(N * H * H * Co, device=dev)

    def run(train_wino):
        cs.pack()
        if train_wino:
            cs.pack_wino(train=True)
        else:
            cs.clear_train_wino()
        cs.zero_grad_images()
        b.grad.zero_()
        out, _, _ = op.forward(srcs, N, Hi, Hi, N)
        if ups:
            dx = [op.backward_ups(gy, srcs[0], N, Hi, Hi, N)]
        else:
            dx = op.backward(gy, srcs, N, Hi, Hi, N)
        return out.clone(), [d.clone() for d in dx]

    o0, d0 = run(False)
    o1, d1 = run(True)
    assert op.train_wino and not torch.equal(o0, o1)
    e = [rel_l2(o1.cpu(), o0.cpu())] + [rel_l2(a.cpu(), b_.cpu()) for a, b_ in zip(d1, d0)]
    print(f"training Winograd N={N} {H}x{H} {C0}+{C1}->{Co} ups={ups}: forward / dgrad vs the direct kernels rel-L2 " + " ".join(f"{v:.1e}" for v in e))
    assert max(e) <= 2e-6
    assert not torch.equal(d0[0], d1[0])                   # the dgrad took the Winograd kernel too (folded upsample: on the 2x grid)
    # and against PyTorch fp32 (CPU)
    xs = [s_.view(N, Hi, Hi, c).permute(0, 3, 1, 2).cpu() for s_, c in zip(srcs, [C0, C1] if C1 else [C0])]
    xin = torch.cat(xs, 1).requires_grad_(True)
    xx = F.interpolate(xin, scale_factor=2, mode="nearest") if ups else xin
    y = F.conv2d(xx, w.detach().cpu(), b.detach().cpu(), padding=1)
    y.backward(gy.view(N, H, H, Co).permute(0, 3, 1, 2).cpu())
    assert rel_l2(o1.view(N, H, H, Co).cpu(), cl2(y.detach())) <= 1e-5
    dx_ref = cl2(xin.grad)
    got = torch.cat([d.view(N, Hi, Hi, -1).cpu() for d in d1], -1)
    assert rel_l2(got, dx_ref) <= 1e-5
