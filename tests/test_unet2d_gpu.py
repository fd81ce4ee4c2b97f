"""GPU parity of the 2-D U-Net pieces (K7 GroupNorm+SiLU dual, K8 attention
pieces, K9/K10 glue) against plain PyTorch fp32 CPU, and of the assembled
VorticityUNet against the golden vectors recorded from the reference."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, rel_l2
from sdeflow_light_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def cl(x):    # (N,C,H,W) -> [N][H*W][C] flat
    return x.permute(0, 2, 3, 1).contiguous().reshape(-1)


@pytest.mark.parametrize("C,H,silu", [(32, 8, True), (64, 4, True), (96, 8, True), (192, 4, True), (256, 4, False), (128, 16, False)])
def test_groupnorm_dual_forward_backward(C, H, silu):
    from sdeflow_light_amd import ops
    torch.manual_seed(C)
    B, G = 3, min(C, 32)
    x, xd = torch.randn(B, C, H, H) * 1.5 + 0.3, torch.randn(B, C, H, H)
    gam, bet = 1 + 0.2 * torch.randn(C), 0.2 * torch.randn(C)

    def f(xx, g_, b_):
        y = F.group_norm(xx, G, g_, b_, eps=1e-5)
        return torch.sigmoid(y) * y if silu else y
    xg, xdg = x.clone().requires_grad_(True), xd.clone().requires_grad_(True)
    gg, bg = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    yp, yt = torch.func.jvp(lambda a: f(a, gg, bg), (xg,), (xdg,))
    P = H * H
    xs = torch.cat([cl(x), cl(xd)]).to(DEV)
    stats = torch.empty(B * G * 4, device=DEV)
    out = ops.groupnorm_dual_forward(xs, gam.to(DEV), bet.to(DEV), B, P, C, G, True, silu, stats=stats)
    half = B * P * C
    assert rel_l2(out[:half].cpu(), cl(yp.detach())) <= 1e-5
    assert rel_l2(out[half:].cpu(), cl(yt.detach())) <= 1e-5
    out1 = ops.groupnorm_dual_forward(cl(x).to(DEV), gam.to(DEV), bet.to(DEV), B, P, C, G, False, silu)
    assert rel_l2(out1.cpu(), cl(yp.detach())) <= 1e-5
    gp, gt = torch.randn_like(yp), torch.randn_like(yt)
    ((yp * gp).sum() + (yt * gt).sum()).backward()
    gout = torch.cat([cl(gp), cl(gt)]).to(DEV)
    dga, dbe = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    gx = ops.groupnorm_dual_backward(xs, gam.to(DEV), bet.to(DEV), stats, gout, dga, dbe, B, P, C, G, silu)
    assert rel_l2(gx[:half].cpu(), cl(xg.grad)) <= 2e-5, rel_l2(gx[:half].cpu(), cl(xg.grad))
    assert rel_l2(gx[half:].cpu(), cl(xdg.grad)) <= 2e-5
    assert rel_l2(dga.cpu(), gg.grad) <= 2e-5 and rel_l2(dbe.cpu(), bg.grad) <= 2e-5
    # skip-branch cotangent added in the apply pass: gx + residual, same bits as a separate add
    res = torch.randn(2 * half, device=DEV)
    dga2, dbe2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    gx2 = ops.groupnorm_dual_backward(xs, gam.to(DEV), bet.to(DEV), stats, gout.clone(), dga2, dbe2, B, P, C, G, silu, residual=res)
    dga1 = torch.zeros(C, device=DEV)
    gx1 = ops.groupnorm_dual_backward(xs, gam.to(DEV), bet.to(DEV), stats, gout.clone(), dga1, torch.zeros(C, device=DEV), B, P, C, G, silu)
    assert torch.equal(gx2, gx1 + res) and torch.equal(dga2, dga1)     # (gout was overwritten in place by the first call above)
    # a SECOND addend (the skip-stack cotangent of the U-Net's encoder tensors): in the same apply pass inside a batched-
    # reduction backward, by one lincomb outside it — the same bits either way
    res2 = torch.randn(2 * half, device=DEV)
    dga3, dbe3 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    with ops.DeferredReduces.on(DEV):
        gx3 = ops.groupnorm_dual_backward(xs, gam.to(DEV), bet.to(DEV), stats, gout.clone(), dga3, dbe3, B, P, C, G, silu,
                                          residual=res, residual2=res2)
    gx4 = ops.groupnorm_dual_backward(xs, gam.to(DEV), bet.to(DEV), stats, gout.clone(), torch.zeros(C, device=DEV),
                                      torch.zeros(C, device=DEV), B, P, C, G, silu, residual=res, residual2=res2)
    assert torch.equal(gx3, (gx1 + res) + res2) and torch.equal(gx4, gx3)
    assert rel_l2(dga3.cpu(), dga1.cpu()) <= 1e-6 and rel_l2(dbe3.cpu(), dbe2.cpu()) <= 1e-6      # batched slot sums: another order


def test_bmm_strided_and_softmax_dual():
    from sdeflow_light_amd import ops
    torch.manual_seed(1)
    Bn, T, C = 3, 80, 32
    qkv = torch.randn(Bn, T, 3 * C)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    ld = 3 * C
    qd = qkv.to(DEV).contiguous()
    S = torch.empty(Bn * T * T, device=DEV)
    ops.bmm(qd, 0, qd, C, S, 0, T, T, C, Bn, (T * ld, ld, 1), (T * ld, 1, ld), (T * T, T, 1), alpha=0.5)
    ref = 0.5 * torch.einsum("btc,bsc->bts", q, k)
    assert rel_l2(S.view(Bn, T, T).cpu(), ref) <= 1e-5
    W = torch.softmax(ref, -1)
    att = torch.zeros(Bn * T * C, device=DEV)
    ops.bmm(W.to(DEV).contiguous(), 0, qd, 2 * C, att, 0, T, C, T, Bn, (T * T, T, 1), (T * ld, ld, 1), (T * C, C, 1))
    assert rel_l2(att.view(Bn, T, C).cpu(), torch.einsum("bts,bsc->btc", W, v)) <= 1e-5
    tr = torch.zeros(Bn * T * C, device=DEV)                   # transposed A, accumulate twice
    for _ in range(2):
        ops.bmm(W.to(DEV).contiguous(), 0, qd, 0, tr, 0, T, C, T, Bn, (T * T, 1, T), (T * ld, ld, 1), (T * C, C, 1), accumulate=True)
    assert rel_l2(tr.view(Bn, T, C).cpu(), 2 * torch.einsum("bts,btc->bsc", W, q)) <= 1e-5
    # dual softmax rows
    Wl, Wdl = torch.randn(Bn * T, T) * 2, torch.randn(Bn * T, T)
    Wg, Wdg = Wl.clone().requires_grad_(True), Wdl.clone().requires_grad_(True)
    Pp, Pt = torch.func.jvp(lambda a: torch.softmax(a, -1), (Wg,), (Wdg,))
    Sd, Wd_, Pd = Wl.to(DEV).clone(), Wdl.to(DEV).clone(), torch.empty(Bn * T, T, device=DEV)
    ops.softmax_dual_forward(Sd, T, Wd_, Pd)
    assert rel_l2(Sd.cpu(), Pp.detach()) <= 1e-6 and rel_l2(Pd.cpu(), Pt.detach()) <= 1e-5
    Pb, Pdb = torch.randn_like(Pp), torch.randn_like(Pt)
    ((Pp * Pb).sum() + (Pt * Pdb).sum()).backward()
    pb, pdb = Pb.to(DEV).clone(), Pdb.to(DEV).clone()
    ops.softmax_dual_backward(Sd, Wd_, pb, pdb, T)
    assert rel_l2(pb.cpu(), Wg.grad) <= 2e-5 and rel_l2(pdb.cpu(), Wdg.grad) <= 2e-5


@pytest.mark.parametrize("Bn,T,C", [(3, 64, 32), (2, 128, 64), (2, 320, 64), (5, 256, 128), (1, 1024, 64), (2, 192, 128)])
def test_fused_attention_forward(Bn, T, C):
    """K8b (sampler path): softmax(scale q k^T) v without the (T,T) matrix, vs plain PyTorch fp32 of
    QKVAttention.forward (model/unet.py:236-250).  Tolerance 2e-5 rel-L2 (online softmax re-association, __expf)."""
    from sdeflow_light_amd import ops
    torch.manual_seed(T + C)
    qkv = torch.randn(Bn, T, 3 * C) * 1.5
    qkv[0, : T // 2, :C] *= 4.0                        # a few peaked rows: exercises the running-max rescale
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    s = C ** -0.25
    ref = torch.einsum("bts,bsc->btc", torch.softmax(torch.einsum("btc,bsc->bts", q * s, k * s), -1), v)
    assert ops.attention_supported(T, C)
    out = torch.full((Bn * T * C,), float("nan"), device=DEV)
    ops.attention_forward(qkv.to(DEV).contiguous().view(-1), out, Bn, T, C, 1.0 / math.sqrt(C))
    assert rel_l2(out.view(Bn, T, C).cpu(), ref) <= 2e-5


def test_fused_attention_unsupported_shapes_fail_loudly():
    from sdeflow_light_amd import ops
    from sdeflow_light_amd._lib import MsgmError
    assert not ops.attention_supported(80, 32) and not ops.attention_supported(64, 48) and not ops.attention_supported(16, 64)
    qkv = torch.zeros(2 * 80 * 96, device=DEV)
    with pytest.raises(MsgmError):
        ops.attention_forward(qkv, torch.zeros(2 * 80 * 32, device=DEV), 2, 80, 32, 1.0)


def test_unet2d_sampler_forward_fused_equals_composed(monkeypatch):
    """The no-tangent forward (sampler) takes the fused attention where the shape allows; it must agree with the
    composed bmm/softmax/bmm path the training step uses (same net, same input)."""
    from sdeflow_light_amd import ops
    from sdeflow_light_amd.NNUnet import VorticityUNet
    from oracle.det_params import load_det_
    torch.manual_seed(3)
    net = VorticityUNet(base_channels=32, channel_mults=(1, 2), num_res_blocks=1, in_space=16, attention_resolutions=(1, 2),
                        flatten_order="F").to(DEV)
    load_det_(net.core)
    x = torch.randn(4, 256, device=DEV)
    t = torch.rand(4, device=DEV)
    calls = []
    real = ops.attention_forward
    monkeypatch.setattr(ops, "attention_forward", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    fused = net(x, t).clone()
    assert calls, "fused attention was not taken for T=256/64, C=32/64"
    monkeypatch.setattr(ops, "attention_supported", lambda T, C: False)
    composed = net(x, t)
    from conftest import within
    within(rel_l2(fused.cpu(), composed.cpu()), 2e-4, "sampler forward, fused vs composed attention")


def test_glue_kernels():
    from sdeflow_light_amd import ops
    from oracle import nets_ref as N
    g = load_golden("g12_embedding")
    emb = ops.timestep_embedding(g["t"].to(DEV), 32)
    assert rel_l2(emb.cpu(), g["emb32"]) <= 1e-6
    torch.manual_seed(2)
    B, C, H = 3, 2, 6
    x = torch.randn(B, C * H * H)
    for order in ("C", "F"):
        img = ops.flat_to_image(x.to(DEV), B, C, H, H, order == "F", 0.2)
        ref = torch.stack([N.flat_to_image(x[:, c * H * H:(c + 1) * H * H], H, H, order)[:, 0] for c in range(C)], -1)   # [B][H][W][C]
        assert rel_l2(img.view(B, H, H, C).cpu(), ref) <= 1e-6
        back = ops.image_to_flat(img, B, C, H, H, order == "F", 5.0)
        assert rel_l2(back.cpu(), x) <= 1e-6
    up = torch.randn(B, 2 * H, 2 * H, C)
    s = ops.sum2x2(up.to(DEV).reshape(-1), B, H, H, C)
    ref = up.view(B, H, 2, H, 2, C).sum((2, 4))
    assert rel_l2(s.view(B, H, H, C).cpu(), ref) <= 1e-6


def _vunet(S_, order, channels=1):
    from sdeflow_light_amd.NNUnet import VorticityUNet
    from oracle.det_params import load_det_
    net = VorticityUNet(base_channels=32, channel_mults=(1, 2, 4), num_res_blocks=2, premodule=None, in_space=S_,
                        attention_resolutions=(2, 4), flatten_order=order, channels=channels)
    if channels == 1:
        load_det_(net)
    else:
        load_det_(net.core)
    return net.to(DEV)


def test_unet2d_state_dict_matches_reference_layout():
    from oracle.shapes import unet2d_shapes
    from oracle import nets_ref as N
    net = _vunet(16, "C")
    want = unet2d_shapes(N.UNet2DConfig(in_space=16), "core.")
    got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert got == {k: tuple(v) for k, v in want.items()}
    assert sum(v.numel() for v in net.state_dict().values()) == 4023233


@pytest.mark.parametrize("tag,S_,order", [("u2d16C", 16, "C"), ("u2d16F", 16, "F"), ("u2d32F", 32, "F")])
def test_unet2d_forward_golden(tag, S_, order):
    g = load_golden("g09_unet2d")
    net = _vunet(S_, order)
    out = net(g[tag + "_x"].to(DEV), g[tag + "_t"].to(DEV))
    assert out.shape == g[tag + "_out"].shape
    from conftest import within
    within(rel_l2(out.cpu(), g[tag + "_out"]), 7e-6, f"VorticityUNet forward vs reference ({tag})")


def test_unet2d_core64_three_channels_golden():
    """64x64x3 (config C4 shape): 4-D image input through the 3-channel core."""
    g = load_golden("g09_unet2d")
    net = _vunet(64, "C", channels=3)
    out = net(g["core64_x"].to(DEV), g["core64_t"].to(DEV))
    from conftest import within
    within(rel_l2(out.cpu(), g["core64_out"]), 5e-5, "64x64x3 core forward vs reference")


def test_unet2d_ssm_golden():
    """SSM loss + all parameter gradients (digest) vs the reference's double backward (g10)."""
    from conftest import check_digest as _check_digest
    from test_host_gpu import make_gen
    g = load_golden("g10_ssm_unets")
    net = _vunet(16, "F")
    gen = make_gen("sgm", net)
    gen.zero_grad()
    per = gen.ssm(g["u2d_x"].to(DEV), u=g["u2d_u_t"].reshape(-1).to(DEV), eps=g["u2d_eps"].to(DEV), u_v=g["u2d_u_v"].to(DEV))
    from conftest import within
    within(rel_l2(per.detach().cpu(), g["u2d_per"]), 8e-6, "2-D U-Net 16x16 per-sample SSM loss vs reference")   # measured 4.1e-06
    per.mean().backward()
    grads = {k: p.grad.cpu() for k, p in gen.a.named_parameters()}
    _check_digest(g, "u2d", grads, "a.", 7e-5)            # measured 3.3e-05 (r2)


def test_unet2d_ssm_vs_oracle_32():
    """Second size (32x32, attention at T = 256 (fused dual kernel, C = 64) and T = 64) against the CPU oracle: HIP no
    further from the float64 oracle than twice the float32 oracle (conftest.parity_vs_fp64)."""
    from oracle.shapes import unet2d_shapes
    from test_host_gpu import make_gen
    from test_round2_gpu import ssm_parity_vs_fp64
    from oracle import nets_ref as N
    from oracle.det_params import det_state_dict
    net = _vunet(32, "F")
    gen = make_gen("sgm", net)
    torch.manual_seed(0)
    B, d = 2, 1024
    x, u, eps, uv = torch.randn(B, d) * 3, torch.rand(B), torch.randn(B, d), torch.rand(B, d)
    cfg = N.UNet2DConfig(in_space=32)
    p = det_state_dict(unet2d_shapes(cfg, "core."))
    score = lambda prm, yy, tt: N.vorticity_unet_forward(prm, yy, tt, cfg, None, "F")
    # slack 6 / 10 (not 2 / 4): at this size the deterministic test parameters amplify rounding by ~1e3, so two correct fp32
    # evaluations land anywhere within a few x of each other.  Measured in round 3 (tools/debug_grad32.py): flat gradient
    # error vs float64 3.5e-05 with the round-2 embedding GEMMs, 1.0e-04 with the embedding bank, 7.8e-05 with the embedding
    # projections replaced by their CORRECTLY ROUNDED values — the float32 oracle itself: 2.0e-05.  The well-conditioned
    # pin is test_round3_gpu.py::test_unet2d_ssm_reference_init_fixture (absolute tolerance).
    ssm_parity_vs_fp64(gen, score, p, x, u, eps, uv, "VorticityUNet 32x32, B=2", slack=6.0, slack_worst=10.0)


def test_unet2d_reference_loop_adam_steps_vs_oracle():
    """zero_grad / ssm(x).mean() / backward / torch.optim.Adam.step (MSGM_higherDim.py:803-809) for two iterations on
    the 2-D U-Net (16x16, attention at T = 64 / 16) against the oracle doing the same on the CPU."""
    from oracle.shapes import unet2d_shapes
    from test_host_gpu import make_gen
    from oracle import sde_ref as S, nets_ref as N, ssm_ref as LR
    torch.manual_seed(4)
    net = _vunet(16, "F")
    gen = make_gen("sgm", net)
    opt = torch.optim.Adam(gen.a.parameters(), lr=1e-3)
    B, d = 2, 256
    cfg = N.UNet2DConfig(in_space=16)
    ref = {k: v.detach().cpu().clone() for k, v in net.named_parameters()}
    assert set(ref) == set(unet2d_shapes(cfg, "core."))
    m = {k: torch.zeros_like(v) for k, v in ref.items()}
    vv = {k: torch.zeros_like(v) for k, v in ref.items()}
    sp = S.SdeSpec()
    score = lambda prm, yy, tt: N.vorticity_unet_forward(prm, yy, tt, cfg, None, "F")
    losses, losses_ref = [], []
    for it in range(2):
        x, u, eps, uv = torch.randn(B, d) * 3, torch.rand(B), torch.randn(B, d), torch.rand(B, d)
        gen.zero_grad()
        loss = gen.ssm(x.to(DEV), u=u.to(DEV), eps=eps.to(DEV), u_v=uv.to(DEV)).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        t = S.clamp_time(sp, u.reshape(B, 1)); y = S.vp_perturb(sp, t, x, eps); v = S.rademacher_from_uniform(uv)
        lref, _, gref = LR.ssm_mean_and_grads(sp, score, ref, t, y, v)
        losses_ref.append(float(lref))
        for k in ref:
            ref[k], m[k], vv[k] = LR.adam_step(ref[k], gref[k], m[k], vv[k], it + 1)
    from conftest import within
    within(max(abs(a_ - b_) / abs(b_) for a_, b_ in zip(losses, losses_ref)), 6e-4, "2 Adam steps, 2-D U-Net 16x16: loss sequence rel. error")
    flat = torch.cat([p_.detach().reshape(-1).cpu() for _, p_ in net.named_parameters()])
    flat_ref = torch.cat([ref[k].reshape(-1) for k, _ in net.named_parameters()])
    # Parameters: conv biases that feed a GroupNorm have an analytically ZERO gradient; Adam turns their rounding noise
    # into +-lr steps (in any implementation, the reference included), so ~1 % of the entries legitimately differ by
    # 2*lr per step — bounded here, while the losses above pin the parameters that matter.
    within(rel_l2(flat, flat_ref), 4e-3, "2 Adam steps, 2-D U-Net 16x16: parameters rel-L2")
    assert float((flat - flat_ref).abs().max()) <= 2 * 2 * 1e-3 + 1e-6


def test_graphed_step_sampler_unet2d_equals_eager():
    """C5 path: one hipGraph-captured EM step (device-side clock) replayed N times == the eager integrator."""
    from sdeflow_light_amd import sde_scheme as SS
    from test_host_gpu import make_gen
    torch.manual_seed(0)
    net = _vunet(16, "F")
    gen = make_gen("sgm", net)
    B, n, N = 6, 256, 5
    x0 = torch.randn(B, n, device=DEV)
    gs = SS.GraphedStepSampler(gen, B, n, N)
    st = gen.base_sde.rng.state.clone()
    a = gs.run(x0).clone()
    gen.base_sde.rng.state.copy_(st)
    b = SS.euler_maruyama_sampler(gen, x0, num_steps=N, keep_all_samples=False)
    assert rel_l2(a.cpu(), b) <= 1e-5, rel_l2(a.cpu(), b)
    c = gs.run(x0).clone()
    assert not torch.equal(a, c) and torch.isfinite(c).all()


# ------------------------------------------------------------------ NormalizeLogRadius conditioning (MSGM default)
def test_normalize_and_embedding_dual_kernels():
    from sdeflow_light_amd import ops
    from oracle import nets_ref as N
    torch.manual_seed(3)
    B, n = 5, 300
    x, xd = torch.randn(B, n) * 2, torch.randn(B, n)
    f = lambda a: tuple(N.normalize_log_radius(a))
    (xn, lr), (xnd, lrd) = torch.func.jvp(f, (x,), (xd,))
    out, logr = ops.normalize_dual(torch.cat([x, xd]).to(DEV), B, n, True, math.sqrt(n))
    assert rel_l2(out[:B].cpu(), xn * math.sqrt(n)) <= 1e-6 and rel_l2(out[B:].cpu(), xnd * math.sqrt(n)) <= 1e-5
    assert rel_l2(logr[:B].cpu(), lr.reshape(-1)) <= 1e-6 and rel_l2(logr[B:].cpu(), lrd.reshape(-1)) <= 1e-5
    t, td = torch.randn(B) * 3, torch.randn(B)
    e, ed = torch.func.jvp(lambda a: N.sinusoidal_embedding(a, 32), (t,), (td,))
    emb = ops.timestep_embedding_dual(torch.cat([t, td]).to(DEV), B, 32)
    assert rel_l2(emb[:B].cpu(), e) <= 1e-6 and rel_l2(emb[B:].cpu(), ed) <= 1e-5


def test_unet_premodule_forward_golden():
    from sdeflow_light_amd.NNUnet import VorticityUNet
    from sdeflow_light_amd.NNUnet1D import UNet1D
    from oracle.det_params import load_det_
    g2, g1 = load_golden("g09_unet2d"), load_golden("g09_unet1d")
    net = VorticityUNet(base_channels=32, channel_mults=(1, 2, 4), num_res_blocks=2, premodule="NormalizeLogRadius", in_space=16,
                        attention_resolutions=(2, 4), flatten_order="F")
    load_det_(net)
    out = net.to(DEV)(g2["u2d16Fn_x"].to(DEV), g2["u2d16Fn_t"].to(DEV))
    assert rel_l2(out.cpu(), g2["u2d16Fn_out"]) <= 1e-4, rel_l2(out.cpu(), g2["u2d16Fn_out"])
    n1 = UNet1D(input_dim=1024, premodule="NormalizeLogRadius")
    load_det_(n1)
    out = n1.to(DEV)(g1["u1dn_x"].to(DEV), g1["u1dn_t"].to(DEV))
    assert rel_l2(out.cpu(), g1["u1dn_out"]) <= 1e-4, rel_l2(out.cpu(), g1["u1dn_out"])


@pytest.mark.parametrize("which", ["1d", "2d"])
def test_unet_premodule_ssm_msgm_vs_oracle(which):
    """The reference's MSGM default (sparse tensor + NormalizeLogRadius U-Net): the log-radius embedding has a
    tangent, so the embedding path itself runs on dual numbers."""
    from sdeflow_light_amd.NNUnet import VorticityUNet
    from sdeflow_light_amd.NNUnet1D import UNet1D
    from oracle.shapes import unet1d_shapes, unet2d_shapes
    from test_host_gpu import make_gen
    from oracle import sde_ref as S, nets_ref as N, ssm_ref as LR
    from oracle.det_params import det_state_dict, load_det_
    torch.manual_seed(1)
    pre = "NormalizeLogRadius"
    if which == "1d":
        n = 64
        net = UNet1D(input_dim=n, premodule=pre)
        p = det_state_dict(unet1d_shapes(n, pre))
        score = lambda prm, yy, tt: N.unet1d_forward(prm, yy, tt, pre)
    else:
        n = 256
        net = VorticityUNet(base_channels=32, channel_mults=(1, 2, 4), num_res_blocks=2, premodule=pre, in_space=16,
                            attention_resolutions=(2, 4), flatten_order="F")
        cfg = N.UNet2DConfig(in_space=16, use_log_norm=True)
        p = det_state_dict(unet2d_shapes(cfg, "core."))
        score = lambda prm, yy, tt: N.vorticity_unet_forward(prm, yy, tt, cfg, pre, "F")
    load_det_(net)
    gen = make_gen("sparse", net.to(DEV), n=n, nsf=4)
    B = 2
    t, y, uv = torch.rand(B, 1).clamp_min(1e-3), torch.randn(B, n) * 2, torch.rand(B, n)
    gen.zero_grad()
    per = gen.ssm_loss(t.to(DEV), y.to(DEV), y.to(DEV), u_v=uv.to(DEV))
    per.mean().backward()
    sp = S.SdeSpec(kind=S.MSGM_SPARSE, n=n)
    loss, per_ref, gref = LR.ssm_mean_and_grads(sp, score, p, t, y, S.rademacher_from_uniform(uv))
    from conftest import within
    tol = {"1d": (5e-7, 3e-6, 2e-5), "2d": (5e-5, 8e-5, 1.5e-4)}[which]          # <= 2x the values measured in round 2
    within(rel_l2(per.detach().cpu(), per_ref), tol[0], f"MSGM + NormalizeLogRadius U-Net ({which}): per-sample loss rel-L2")
    names = [k for k, _ in gen.a.named_parameters()]
    flat = torch.cat([pp.grad.reshape(-1).cpu() for _, pp in gen.a.named_parameters()])
    ref = torch.cat([gref[k].reshape(-1) for k in names])
    within(rel_l2(flat, ref), tol[1], f"MSGM + NormalizeLogRadius U-Net ({which}): flat gradient rel-L2")
    worst = max(rel_l2(pp.grad.cpu(), gref[k]) for k, pp in gen.a.named_parameters() if "scale_embed" in k)
    within(worst, tol[2], f"MSGM + NormalizeLogRadius U-Net ({which}): worst scale_embed tensor")   # they must receive their tangent contributions


@pytest.mark.parametrize("kind", ["unet2d", "unet1d"])
def test_unet_trainer_graph_replay_equals_eager(kind):
    """UNetScoreTrainer(use_graph=True) replays one captured hipGraph per step; losses and parameters after 4 steps at
    the driver's order of learning rate (1e-3) must EQUAL the eager trainer's bit for bit: no kernel on the step uses
    float atomics any more (weight / bias / GroupNorm-parameter gradients and the GroupNorm moments go through
    per-workgroup slots added in a fixed order), so there is no atomics-order noise for Adam to amplify."""
    from sdeflow_light_amd.SDEs import SGMsde, PluginReverseSDE
    from sdeflow_light_amd.train import UNetScoreTrainer
    from oracle.det_params import load_det_

    def make():
        torch.manual_seed(11)
        if kind == "unet2d":
            from sdeflow_light_amd.NNUnet import VorticityUNet
            net = VorticityUNet(base_channels=32, channel_mults=(1, 2), num_res_blocks=1, in_space=16,
                                attention_resolutions=(2,), flatten_order="F").to(DEV)
            load_det_(net.core)
            d = 256
        else:
            from sdeflow_light_amd.NNUnet1D import UNet1D
            net = UNet1D(input_dim=128, base_channels=16, channel_mults=(1, 2), emb_dim=32).to(DEV)
            d = 128
        T = torch.nn.Parameter(torch.FloatTensor([1.0]), requires_grad=False)
        gen = PluginReverseSDE(SGMsde(T=T, num_steps_forward=16, device=DEV), net, T, deviceReverseSDE=DEV).to(DEV)
        return gen, net, d

    out = {}
    for use_graph in (False, True):
        gen, net, d = make()
        tr = UNetScoreTrainer(gen, 8, d, lr=1e-3, use_graph=use_graph, seed=5)
        p0 = net.flat_parameters()[0].clone()
        torch.manual_seed(0)
        tr.set_data(torch.randn(8, d, device=DEV))
        losses = [float(tr.step()) for _ in range(4)]
        assert (tr.graph is not None) == use_graph
        if use_graph:                                       # kernel nodes only: no memset / memcpy node in the captured step
            kinds = ops.graph_node_kinds(tr.graph)
            print(f"{kind}: captured train step = {kinds}")
            assert set(kinds) == {"kernel"} and kinds["kernel"] > 50, kinds
        flat, _ = net.flat_parameters()
        assert float((flat - p0).abs().max()) > 1e-6, "parameters did not move"
        out[use_graph] = (losses, flat.clone().cpu())
    assert all(math.isfinite(l) for l in out[True][0])
    e = rel_l2(out[True][1], out[False][1])
    print(f"{kind}: graph replay vs eager after 4 Adam steps at lr 1e-3: parameters rel-L2 {e:.2e}, losses {out[True][0]} vs {out[False][0]}")
    assert out[True][0] == out[False][0]
    assert torch.equal(out[True][1], out[False][1])


@pytest.mark.parametrize("kind", ["unet2d", "unet1d"])
def test_unet_gradients_are_bitwise_reproducible(kind):
    """Two evaluations of ssm(x).mean().backward() on the same inputs give the SAME bits in every parameter gradient
    (float atomics in k_wgrad_tile / k_conv_wgrad / GroupNorm parameter gradients, double atomics in the
    GroupNorm moments — all replaced by slot-ordered sums), including the fused dual attention's slab-reduced qbar."""
    from test_host_gpu import make_gen, _unet1d
    torch.manual_seed(3)
    if kind == "unet2d":
        gen, B, d = make_gen("sgm", _vunet(32, "F")), 4, 1024          # attention at T = 256 (C = 64, fused) and T = 64 (C = 128)
    else:
        gen, B, d = make_gen("sgm", _unet1d(256)), 4, 256
    x, u, eps, uv = torch.randn(B, d, device=DEV), torch.rand(B, device=DEV), torch.randn(B, d, device=DEV), torch.rand(B, d, device=DEV)
    runs = []
    for _ in range(3):
        gen.zero_grad()
        per = gen.ssm(x, u=u, eps=eps, u_v=uv)
        per.mean().backward()
        runs.append((per.detach().clone(), torch.cat([p.grad.reshape(-1) for p in gen.a.parameters()]).clone()))
        torch.cuda.synchronize()
    for per, g in runs[1:]:
        assert torch.equal(per, runs[0][0])
        assert torch.equal(g, runs[0][1]), float((g - runs[0][1]).abs().max())
    assert float(runs[0][1].abs().max()) > 0


def test_unet2d_sampler_forward_gn_fold_equals_unfused(monkeypatch):
    """The no-tangent forward folds GroupNorm(+SiLU) into the consuming conv and the residual into its epilogue;
    with MSGM_NO_GN_FOLD the separate GroupNorm kernels run.  Same net, same input: 2e-4 (whole network)."""
    from sdeflow_light_amd.NNUnet import VorticityUNet
    from oracle.det_params import load_det_
    torch.manual_seed(4)
    net = VorticityUNet(base_channels=32, channel_mults=(1, 2), num_res_blocks=2, in_space=16, attention_resolutions=(1, 2),
                        flatten_order="F", premodule="NormalizeLogRadius").to(DEV)
    load_det_(net.core)
    x, t = torch.randn(5, 256, device=DEV), torch.rand(5, device=DEV)
    fused = net(x, t).clone()
    monkeypatch.setenv("MSGM_NO_GN_FOLD", "1")
    plain = net(x, t)
    from conftest import within
    within(rel_l2(fused.cpu(), plain.cpu()), 1.5e-4, "sampler forward, GroupNorm folded into the conv vs separate")


def test_bmm_dual_output_shares_the_big_operand():
    """msgm_bmm_dual: C = A.B + A2.B2 and C3 = A.B3 in one pass over A (the (T,T) operand of the dual attention),
    plain and transposed A, channel-sliced operands — vs torch einsum, 1e-5."""
    from sdeflow_light_amd import ops
    torch.manual_seed(9)
    Bn, T, C = 3, 96, 32
    ld = 3 * C
    P, Pd = torch.randn(Bn, T, T), torch.randn(Bn, T, T)
    qkv = torch.randn(2 * Bn, T, ld)                                   # primal | tangent rows
    v, vd = qkv[:Bn, :, 2 * C:], qkv[Bn:, :, 2 * C:]
    half = Bn * T * ld
    qd, Pg, Pdg = qkv.to(DEV).contiguous().view(-1), P.to(DEV).contiguous().view(-1), Pd.to(DEV).contiguous().view(-1)
    att = torch.full((2 * Bn * T * C,), float("nan"), device=DEV)
    sP, sPt, sv, sa = (T * T, T, 1), (T * T, 1, T), (T * ld, ld, 1), (T * C, C, 1)
    ops.bmm(Pg, 0, qd, half + 2 * C, att, Bn * T * C, T, C, T, Bn, sP, sv, sa, pair2=(Pdg, 0, qd, 2 * C), third=(qd, 2 * C, att, 0))
    a_ref = torch.einsum("bts,bsc->btc", P, v)
    ad_ref = torch.einsum("bts,bsc->btc", P, vd) + torch.einsum("bts,bsc->btc", Pd, v)
    out = att.view(2 * Bn, T, C).cpu()
    assert rel_l2(out[:Bn], a_ref) <= 1e-5 and rel_l2(out[Bn:], ad_ref) <= 1e-5
    # transposed shared operand, alpha, outputs written into channel slices of a wider tensor
    dq = torch.zeros(2 * Bn * T * ld, device=DEV)
    ops.bmm(Pdg, 0, qd, half, dq, C, T, C, T, Bn, sPt, sv, sv, alpha=0.5, pair2=(Pg, 0, qd, 0), third=(qd, 0, dq, half + C))
    q_, qdot = qkv[:Bn, :, :C], qkv[Bn:, :, :C]
    k_ref = 0.5 * (torch.einsum("bts,btc->bsc", Pd, qdot) + torch.einsum("bts,btc->bsc", P, q_))
    kd_ref = 0.5 * torch.einsum("bts,btc->bsc", Pd, q_)
    o = dq.view(2 * Bn, T, ld).cpu()
    assert rel_l2(o[:Bn, :, C:2 * C], k_ref) <= 1e-5 and rel_l2(o[Bn:, :, C:2 * C], kd_ref) <= 1e-5
    assert float(o[:, :, :C].abs().max()) == 0.0 and float(o[:, :, 2 * C:].abs().max()) == 0.0


@pytest.mark.parametrize("order", ["C", "F"])
def test_flat_img_helpers_match_reference_semantics(order):
    """flat_to_img / img_to_flat (NNUnet.py:26-77): /5 and view (C) or view(B,1,W,H).transpose (F); inverse x5."""
    from sdeflow_light_amd.NNUnet import flat_to_img, img_to_flat
    torch.manual_seed(2)
    B, H, W = 3, 6, 10
    x = torch.randn(B, H * W)
    ref = (x / 5).view(B, 1, H, W) if order == "C" else (x / 5).view(B, 1, W, H).transpose(2, 3).contiguous()
    img = flat_to_img(x.to(DEV), H, W, order)
    assert img.shape == (B, 1, H, W) and rel_l2(img.cpu(), ref) <= 1e-7
    back = img_to_flat(img, order)
    assert back.shape == (B, H * W) and rel_l2(back.cpu(), x) <= 1e-6


@pytest.mark.parametrize("Bp,T,C", [(3, 64, 32), (2, 128, 64), (2, 320, 64), (1, 1024, 64), (5, 192, 32),
                                    (2, 256, 128), (3, 64, 128), (1, 32, 128), (9, 96, 128), (40, 256, 128),   # C = 128: the 16x16 / 8x8 blocks (r3)
                                    # launches big enough for the backward's TWO passes over the key blocks (the second adds to
                                    # the first's query-gradient slab rows): Bp * T / keys-per-workgroup >= 2048
                                    (1024, 128, 32), (512, 256, 64), (1024, 64, 128)])
def test_fused_dual_attention_forward_backward(Bp, T, C):
    """K8 (training path): attention on dual numbers, forward and backward, without the (T,T) tensors — vs plain PyTorch
    fp32 (torch.func.jvp of QKVAttention.forward's arithmetic, model/unet.py:236-250, then autograd of the pair)."""
    from sdeflow_light_amd import ops
    torch.manual_seed(T + C + Bp)
    qkv = torch.randn(2 * Bp, T, 3 * C) * 1.2
    qkv[0, : T // 2, :C] *= 3.0                          # a few peaked rows: exercises the running-max rescale
    s = C ** -0.25

    def attn(x):
        q, k, v = x[..., :C], x[..., C:2 * C], x[..., 2 * C:]
        return torch.einsum("bts,bsc->btc", torch.softmax(torch.einsum("btc,bsc->bts", q * s, k * s), -1), v)
    xp, xt = qkv[:Bp].clone().requires_grad_(True), qkv[Bp:].clone().requires_grad_(True)
    o, od = torch.func.jvp(attn, (xp,), (xt,))
    assert ops.attention_dual_supported(T, C)
    dev_qkv = qkv.to(DEV).contiguous().view(-1)
    att, stats = ops.attention_dual_forward(dev_qkv, Bp, T, C, 1.0 / math.sqrt(C))
    a = att.view(2 * Bp, T, C).cpu()
    e_o, e_od = rel_l2(a[:Bp], o.detach()), rel_l2(a[Bp:], od.detach())
    g = torch.randn(2 * Bp, T, C)
    ((o * g[:Bp]).sum() + (od * g[Bp:]).sum()).backward()
    dq = ops.attention_dual_backward(dev_qkv, att, g.to(DEV).contiguous().view(-1), stats, Bp, T, C, 1.0 / math.sqrt(C))
    d = dq.view(2 * Bp, T, 3 * C).cpu()
    names = ("q", "k", "v")
    errs = {}
    for i, nm in enumerate(names):
        errs[nm] = rel_l2(d[:Bp, :, i * C:(i + 1) * C], xp.grad[..., i * C:(i + 1) * C])
        errs[nm + "dot"] = rel_l2(d[Bp:, :, i * C:(i + 1) * C], xt.grad[..., i * C:(i + 1) * C])
    print(f"dual attention Bp={Bp} T={T} C={C}: o {e_o:.1e} odot {e_od:.1e} | " + " ".join(f"{k}bar {v:.1e}" for k, v in errs.items()))
    assert e_o <= 2e-5 and e_od <= 2e-5
    assert max(errs.values()) <= 2e-5
    # deterministic: a second backward gives the same bits (slab reduction in block order, no atomics)
    dq2 = ops.attention_dual_backward(dev_qkv, att, g.to(DEV).contiguous().view(-1), stats, Bp, T, C, 1.0 / math.sqrt(C))
    assert torch.equal(dq, dq2)


def test_training_decoder_without_cat_equals_cat_path(monkeypatch):
    """Training path: the decoder's cat([h, skip]) (model/unet.py:514) is never materialised — GroupNorm, the 1x1 skip conv
    and their backward read the two tensors as two sources.  Same loss and gradients as the path that concatenates."""
    from test_host_gpu import make_gen
    torch.manual_seed(5)
    gen = make_gen("sgm", _vunet(32, "F"))
    B, d = 3, 1024
    x, u, eps, uv = torch.randn(B, d, device=DEV), torch.rand(B, device=DEV), torch.randn(B, d, device=DEV), torch.rand(B, d, device=DEV)
    res = {}
    for cat in (False, True):
        if cat:
            monkeypatch.setenv("MSGM_TRAIN_CAT", "1")
        gen.zero_grad()
        per = gen.ssm(x, u=u, eps=eps, u_v=uv)
        per.mean().backward()
        res[cat] = (per.detach().clone(), {k: p.grad.detach().clone() for k, p in gen.a.named_parameters()})
    from conftest import within
    within(rel_l2(res[False][0].cpu(), res[True][0].cpu()), 1e-6, "no-cat vs cat training path: per-sample loss")
    worst = max(rel_l2(res[False][1][k].cpu(), g.cpu()) for k, g in res[True][1].items() if float(g.norm()) > 1e-3 * max(float(v.norm()) for v in res[True][1].values()))
    within(worst, 1e-5, "no-cat vs cat training path: worst parameter-gradient tensor")
    skipw = [k for k in res[True][1] if "output_blocks" in k and "skip_connection.weight" in k]
    assert skipw and all(float(res[False][1][k].abs().max()) > 0 for k in skipw)       # the twin's gradient reached the parameter


@pytest.mark.parametrize("T,C", [(256, 128), (64, 128), (96, 32)])
def test_bmm_lds_staged_equals_register_direct(T, C, monkeypatch):
    """The LDS-staged batched GEMM (k_bmm_lds) against torch einsum on the eight operand layouts the composed dual attention
    uses (K-contiguous / transposed A and B, second operand pair, second output sharing the big operand)."""
    from sdeflow_light_amd import ops
    torch.manual_seed(T + C)
    Bn, ld = 2, 3 * C
    qkv = torch.randn(2 * Bn, T, ld)
    P, Pd = torch.randn(Bn, T, T), torch.randn(Bn, T, T)
    half = Bn * T * ld
    qd, Pg, Pdg = qkv.to(DEV).contiguous().view(-1), P.to(DEV).contiguous().view(-1), Pd.to(DEV).contiguous().view(-1)
    q, k, v = qkv[:Bn, :, :C], qkv[:Bn, :, C:2 * C], qkv[:Bn, :, 2 * C:]
    qdot, kdot, vdot = qkv[Bn:, :, :C], qkv[Bn:, :, C:2 * C], qkv[Bn:, :, 2 * C:]
    sq, sk, sS, sSt, sa = (T * ld, ld, 1), (T * ld, 1, ld), (T * T, T, 1), (T * T, 1, T), (T * C, C, 1)
    S = torch.empty(Bn * T * T, device=DEV)
    ops.bmm(qd, half, qd, C, S, 0, T, T, C, Bn, sq, sk, sS, alpha=0.5, pair2=(qd, 0, qd, half + C))      # qdot k^T + q kdot^T
    ref = 0.5 * (torch.einsum("btc,bsc->bts", qdot, k) + torch.einsum("btc,bsc->bts", q, kdot))
    e1 = rel_l2(S.view(Bn, T, T).cpu(), ref)
    att = torch.full((2 * Bn * T * C,), float("nan"), device=DEV)
    ops.bmm(Pg, 0, qd, half + 2 * C, att, Bn * T * C, T, C, T, Bn, sS, sq, sa, pair2=(Pdg, 0, qd, 2 * C), third=(qd, 2 * C, att, 0))
    a = att.view(2 * Bn, T, C).cpu()
    e2 = max(rel_l2(a[:Bn], torch.einsum("bts,bsc->btc", P, v)),
             rel_l2(a[Bn:], torch.einsum("bts,bsc->btc", P, vdot) + torch.einsum("bts,bsc->btc", Pd, v)))
    dq = torch.zeros(2 * Bn * T * ld, device=DEV)
    ops.bmm(Pdg, 0, qd, half, dq, C, T, C, T, Bn, sSt, sq, sq, alpha=0.5, pair2=(Pg, 0, qd, 0), third=(qd, 0, dq, half + C))   # transposed A
    o = dq.view(2 * Bn, T, ld).cpu()
    e3 = max(rel_l2(o[:Bn, :, C:2 * C], 0.5 * (torch.einsum("bts,btc->bsc", Pd, qdot) + torch.einsum("bts,btc->bsc", P, q))),
             rel_l2(o[Bn:, :, C:2 * C], 0.5 * torch.einsum("bts,btc->bsc", Pd, q)))
    print(f"k_bmm_lds T={T} C={C}: NT pair {e1:.1e}, NN dual {e2:.1e}, TN dual {e3:.1e}")
    assert max(e1, e2, e3) <= 2e-6


def test_sampler_forward_groupnorm_statistics_from_conv_epilogues(monkeypatch):
    """Sampler path: the GroupNorm statistics come from the per-channel sums the producing convolutions leave behind
    (no pass over the tensor).  Same score as with the statistics pass (MSGM_NO_CHANSTATS) up to fp32 summation order, and
    the stand-alone statistics kernel no longer runs for the layers whose producer has the by-product."""
    net = _vunet(32, "F")
    torch.manual_seed(2)
    B = 40                                                  # 16x16 tiles at 32x32, 8x16 tiles below
    x = torch.randn(B, 32 * 32, device=DEV)
    s = torch.rand(B, device=DEV) * 0.9 + 0.05
    calls = {"cs": 0, "full": 0}
    real_cs, real_full = ops.groupnorm_affine_cs, ops.groupnorm_affine
    monkeypatch.setattr(ops, "groupnorm_affine_cs", lambda *a, **k: (calls.__setitem__("cs", calls["cs"] + 1), real_cs(*a, **k))[1])
    monkeypatch.setattr(ops, "groupnorm_affine", lambda *a, **k: (calls.__setitem__("full", calls["full"] + 1), real_full(*a, **k))[1])
    a = net(x, s)
    n_cs, n_full = calls["cs"], calls["full"]
    monkeypatch.setenv("MSGM_NO_CHANSTATS", "1")
    b = net(x, s)
    e = rel_l2(a.cpu(), b.cpu())
    print(f"sampler forward, GroupNorm statistics from conv epilogues ({n_cs} layers; {n_full} by a pass over the tensor) "
          f"vs all by passes: rel-L2 {e:.2e}")
    assert n_cs >= 40 and n_full <= 2
    assert calls["full"] - n_full == n_cs + n_full          # the switch sends every layer through the statistics pass
    # the statistics themselves agree to ~1e-8 (tests/test_conv_gpu.py::test_conv_channel_statistics_byproduct); the
    # untrained deterministic-parameter U-Net amplifies ANY fp32 reordering by ~100x over its 45 normalisations — the same
    # 1-2e-5 separates the HIP forward from the fp32 oracle.  Measured 1.5e-5.
    assert e <= 3e-5
