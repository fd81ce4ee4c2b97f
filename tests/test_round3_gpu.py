"""-m gpu: round-3 pins.

(a) hipGraph replay after an idle gap: the captured U-Net train step must give the same gradients —
    bit for bit — back to back and after the GPU went idle (the round-2 finding: hipMemsetAsync nodes lost their
    ordering after an idle gap; every zero-fill on the path is a kernel node since);
(b) trainer checkpoints interchange with torch.optim.Adam / FusedAdam built over gen_sde.parameters();
(c) the MSGM (multiplicative SDE) graph-captured trainer: graph == eager bit for bit, per-sample loss + gradients against
    the CPU oracle.
"""
import time

import pytest
import torch

from conftest import rel_l2
from sdeflow_light_amd import ops
from test_host_gpu import make_gen

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _unet1d_small():
    from sdeflow_light_amd.NNUnet1D import UNet1D
    return UNet1D(input_dim=128, base_channels=16, channel_mults=(1, 2), emb_dim=32)


@pytest.mark.parametrize("net_kind", ["unet1d", "unet2d"])
def test_graph_replay_after_idle_gap_is_bitwise(net_kind):
    """capture -> replay -> synchronise + 0.5 s idle -> replay: gradients torch.equal (lr = 0 and a pinned Philox
    offset make every step the same computation)."""
    from sdeflow_light_amd.train import UNetScoreTrainer
    torch.manual_seed(1)
    if net_kind == "unet1d":
        net, B, d = _unet1d_small(), 8, 128
    else:
        from test_unet2d_gpu import _vunet
        net, B, d = _vunet(16, "F"), 4, 256
    gen = make_gen("sgm", net)
    tr = UNetScoreTrainer(gen, B, d, lr=0.0, seed=4, use_graph=True)
    torch.manual_seed(0)
    tr.set_data(torch.randn(B, d, device=DEV))

    def step():
        tr.rng.state[1] = 7                     # same noise every step; lr = 0: same parameters
        loss = float(tr.step())
        return loss, tr.gbuf.clone()
    l0, g0 = step()                             # eager (the capture's warm-up step)
    l1, g1 = step()                             # replay
    l2, g2 = step()                             # replay, back to back
    torch.cuda.synchronize()
    time.sleep(0.5)
    l3, g3 = step()                             # replay that starts on an idle GPU
    torch.cuda.synchronize()
    time.sleep(0.5)
    l4, g4 = step()
    assert set(ops.graph_node_kinds(tr.graph)) == {"kernel"}
    assert torch.isfinite(g0).all() and float(g0.abs().max()) > 0
    for tag, g in (("replay", g1), ("back-to-back replay", g2), ("replay after 0.5 s idle", g3), ("second idle replay", g4)):
        assert torch.equal(g, g0), f"{tag}: max abs diff {float((g - g0).abs().max()):.3e}"
    assert l0 == l1 == l2 == l3 == l4


def test_trainer_checkpoint_interchanges_with_adam_and_fused_adam(tmp_path):
    """the reference optimizer is torch.optim.Adam(gen_sde.parameters()) whose index 0 is the
    non-trainable T.  A trainer checkpoint must load into Adam / FusedAdam over gen_sde.parameters() and back."""
    from sdeflow_light_amd.NN import MLP
    from sdeflow_light_amd.optim import FusedAdam
    from sdeflow_light_amd.train import MLPScoreTrainer
    torch.manual_seed(0)
    gen = make_gen("sgm", MLP(2))
    tr = MLPScoreTrainer(gen, 1024, lr=1e-3, seed=3, use_graph=False)
    tr.set_data(torch.randn(1024, 2, device=DEV))
    for _ in range(3):
        tr.step()
    sd = tr.state_dict()
    allp = list(gen.parameters())
    assert not allp[0].requires_grad and allp[0].numel() == 1              # T first, as upstream
    assert sd["param_groups"][0]["params"] == list(range(len(allp))) and 0 not in sd["state"]
    assert sorted(sd["state"]) == list(range(1, len(allp)))
    # -> stock Adam over same-shaped parameters in the same order
    ref_params = [torch.nn.Parameter(p.detach().cpu().clone(), requires_grad=p.requires_grad) for p in allp]
    adam = torch.optim.Adam(ref_params, lr=1e-3)
    adam.load_state_dict({k: v for k, v in sd.items() if k != "philox"})
    assert len(adam.state) == len(allp) - 1 and all(int(s["step"]) == 3 for s in adam.state.values())
    off = 0
    for p_ref, p in zip(ref_params[1:], allp[1:]):
        k = p.numel()
        assert torch.equal(adam.state[p_ref]["exp_avg"].reshape(-1), tr.m[off:off + k].cpu())
        off += k
    # -> FusedAdam over gen.parameters()
    fa = FusedAdam(gen.parameters(), lr=1e-3)
    fa.load_state_dict({k: v for k, v in sd.items() if k != "philox"})
    assert fa._step_host == 3
    off = 0
    for p in allp[1:]:
        k = p.numel()
        assert torch.equal(fa.state[p]["exp_avg_sq"].reshape(-1), tr.v[off:off + k])
        off += k
    # <- and an Adam / FusedAdam-written state back into a fresh trainer
    torch.manual_seed(0)
    gen2 = make_gen("sgm", MLP(2))
    tr2 = MLPScoreTrainer(gen2, 1024, lr=1e-3, seed=3, use_graph=False)
    tr2.load_state_dict(fa.state_dict())
    assert torch.equal(tr2.m, tr.m) and torch.equal(tr2.v, tr.v) and int(tr2.step_dev.item()) == 3
    tr3 = MLPScoreTrainer(make_gen("sgm", MLP(2)), 1024, lr=1e-3, seed=3, use_graph=False)
    tr3.load_state_dict(adam.state_dict())
    assert torch.equal(tr3.m.cpu(), tr.m.cpu()) and int(tr3.step_dev.item()) == 3


@pytest.mark.parametrize("rows,n_bias,cos,K", [(2, 2, [32, 64, 128], 128), (64, 32, [32, 96, 128, 40], 128), (200, 100, [64, 128], 64)])
def test_embedding_bank_forward_backward_vs_fp64(rows, n_bias, cos, K):
    """ops.EmbBank (all ResBlocks' emb_layers[1] in one launch per direction, model/unet.py:145-151,176-180) against
    float64 torch: forward rows, weight / bias gradients (bias over the primal rows only, also written to the partner conv
    bias) and the embedding cotangent."""
    torch.manual_seed(rows + K)
    items = []
    for co in cos:
        w = torch.nn.Parameter(torch.randn(co, K, device=DEV) / K ** 0.5)
        b = torch.nn.Parameter(torch.randn(co, device=DEV))
        cb = torch.nn.Parameter(torch.randn(co, device=DEV))
        for t in (w, b, cb):
            t.grad = torch.full_like(t, 7.0)                       # must be overwritten, not accumulated into
        items.append((w, b, cb))
    bank = ops.EmbBank(items, K)
    semb = torch.randn(rows, K, device=DEV)
    outs = bank.forward(semb, rows, n_bias)
    worst = 0.0
    for (w, b, _), o in zip(items, outs):
        ref = semb.double() @ w.double().t()
        ref[:n_bias] += b.double()
        worst = max(worst, rel_l2(o.view(rows, -1).cpu(), ref.cpu()))
    douts = [torch.randn(rows, co, device=DEV) for co in cos]
    for d, dd in zip(bank.dout, douts):
        d.copy_(dd.reshape(-1))
    dsemb = torch.full((rows, K), 3.0, device=DEV)
    bank.backward(semb, dsemb, rows, n_bias)
    ref_ds = torch.zeros(rows, K, dtype=torch.float64, device=DEV)
    for (w, b, cb), dd in zip(items, douts):
        worst = max(worst, rel_l2(w.grad.cpu(), (dd.double().t() @ semb.double()).cpu()))
        rb = dd[:n_bias].double().sum(0).cpu()
        worst = max(worst, rel_l2(b.grad.cpu(), rb), rel_l2(cb.grad.cpu(), rb))
        ref_ds += dd.double() @ w.double()
    worst = max(worst, rel_l2(dsemb.cpu(), ref_ds.cpu()))
    print(f"embedding bank rows={rows} cos={cos} K={K}: worst rel-L2 vs float64 {worst:.2e}")
    assert worst <= 1e-6


@pytest.mark.parametrize("tag,S_", [("w32", 32), ("w64", 64)])
def test_unet2d_ssm_reference_init_fixture(tag, S_):
    """VERDICT r2 #5a: the HIP training path against the REFERENCE's own double-backward SSM (tests/golden/g17, recorded
    by tools/make_golden.py) on a well-conditioned parameter set — default-init statistics, zero-init layers re-randomised
    small (oracle.det_params.load_init_like_) — at 32x32 (attention T = 256 / 64) and 64x64 (T = 1024 / 256, the C4 network's
    attention shapes), B = 2.  ABSOLUTE tolerances (rel-L2, <= 3x what round 3 measured): forward 3e-6 (measured 1.1e-6),
    per-sample loss 6e-7 (8.8e-8 / 2.1e-7), per-tensor gradient digest 3e-5 (1.2e-5 / 5.8e-6; floor 1e-3 of the largest
    tensor norm, conftest.check_digest)."""
    from conftest import check_digest, load_golden, within
    from oracle.det_params import load_init_like_
    from sdeflow_light_amd.NNUnet import VorticityUNet
    g = load_golden("g17_ssm_wellconditioned")
    net = VorticityUNet(base_channels=32, channel_mults=(1, 2, 4), num_res_blocks=2, premodule=None, in_space=S_,
                        attention_resolutions=(2, 4), flatten_order="F")
    load_init_like_(net)
    gen = make_gen("sgm", net)
    out = gen.a(g[tag + "_x"].to(DEV), g[tag + "_fwd_t"].to(DEV))
    within(rel_l2(out.cpu(), g[tag + "_fwd"]), 3e-6, f"VorticityUNet {S_}x{S_} forward vs reference (well-conditioned fill)")
    gen.zero_grad()
    per = gen.ssm(g[tag + "_x"].to(DEV), u=g[tag + "_u_t"].reshape(-1).to(DEV), eps=g[tag + "_eps"].to(DEV),
                  u_v=g[tag + "_u_v"].to(DEV))
    within(rel_l2(per.detach().cpu(), g[tag + "_per"]), 6e-7, f"VorticityUNet {S_}x{S_} per-sample SSM loss vs reference")
    per.mean().backward()
    grads = {k: p.grad.cpu() for k, p in gen.a.named_parameters()}
    check_digest(g, tag, grads, "a.", 3e-5)


def _msgm_trainer(kind, use_graph, seed=11):
    """(trainer, gen) for the multiplicative SDE: 'sparse' = UNet1D (d = 128, NormalizeLogRadius as the driver uses with
    MSGM, MSGM_higherDim.py:704-725) on the sparse rotation tensor, 'dense' = MLP (d = 6) on the dense rank-3 tensor."""
    from sdeflow_light_amd.train import MLPScoreTrainer, UNetScoreTrainer
    torch.manual_seed(3)
    if kind == "sparse":
        from sdeflow_light_amd.NNUnet1D import UNet1D
        net, B, d = UNet1D(input_dim=128, base_channels=16, channel_mults=(1, 2), emb_dim=32, premodule="NormalizeLogRadius"), 8, 128
        gen = make_gen("sparse", net, n=d, nsf=4)
        tr = UNetScoreTrainer(gen, B, d, lr=1e-3, seed=seed, use_graph=use_graph)
    else:
        from sdeflow_light_amd.NN import MLP
        net, B, d = MLP(6, premodule="NormalizeLogRadius"), 256, 6
        gen = make_gen("dense", net, n=d, nsf=4)
        tr = MLPScoreTrainer(gen, B, lr=1e-3, seed=seed, use_graph=use_graph)
    torch.manual_seed(5)
    tr.set_data(torch.randn(B, d, device=DEV) * 1.5)
    return tr, gen


@pytest.mark.parametrize("kind", ["sparse", "dense"])
def test_msgm_trainer_graph_equals_eager_and_matches_ssm(kind):
    """the multiplicative SDE (SDEs.py:78-132,221-509; MSGM_higherDim.py:733-746) through the graph-captured
    trainers.  (i) the captured step holds kernel nodes only and replays to the eager run's parameters BIT FOR BIT;
    (ii) the trainer's first step computes the loss ``PluginReverseSDE.ssm`` (pinned against the reference by the g14 /
    g08 fixtures) computes from the same stream state — same draws, same kernels."""
    out = {}
    for use_graph in (False, True):
        tr, gen = _msgm_trainer(kind, use_graph)
        st0 = tr.rng.state.clone()
        x = tr.x.clone()
        p0 = tr.flat.clone()
        losses = [float(tr.step()) for _ in range(4)]
        out[use_graph] = (losses, tr.flat.clone(), tr.rng.state.clone())
        if use_graph:
            assert set(ops.graph_node_kinds(tr.graph)) == {"kernel"}
        else:
            # (ii) replay the first step through the reference-surface entry point on a twin with the same start state
            tr2, gen2 = _msgm_trainer(kind, False)
            assert torch.equal(tr2.flat, p0) and torch.equal(tr2.rng.state, st0)
            gen2.zero_grad()
            per = gen2.ssm(x)
            loss_ssm = float(per.mean())
            print(f"MSGM {kind}: trainer step-1 loss {losses[0]:.6f} vs PluginReverseSDE.ssm {loss_ssm:.6f}")
            assert abs(losses[0] - loss_ssm) <= 2e-6 * max(1.0, abs(loss_ssm))
            nsf = gen2.base_sde.num_steps_forward
            assert int(tr2.rng.state[1]) - int(st0[1]) == 2 * nsf + 3            # forward perturbation + the probe
    (l0, f0, s0), (l1, f1, s1) = out[False], out[True]
    assert l0 == l1 and torch.equal(f0, f1) and torch.equal(s0, s1)
    assert all(v == v for v in l0) and float((f0 - p0).abs().max()) > 0


def test_mu_with_a_device_time_tensor_equals_the_scalar_call():
    """The reference calls sde.mu(t, x) with t a (B,1) DEVICE tensor filled with one value (sde_scheme.py:81-83).  That form
    stays on the device (the stage kernel reads its clock from the tensor: no .item() round trip per call) and equals the
    host-scalar form bit for bit."""
    from sdeflow_light_amd.NN import MLP
    torch.manual_seed(2)
    gen = make_gen("sgm", MLP(2))
    y = torch.randn(64, 2, device=DEV)
    t = 0.3125
    for fn in (gen.mu, gen.mu_Strato):
        a = fn(t, y, 0.5)
        b = fn(torch.full((64, 1), t, device=DEV), y, 0.5)
        assert torch.equal(a, b)


def test_unet2d_reference_loop_adam_steps_wellconditioned():
    """The reference loop (zero_grad / ssm(x).mean() / backward / Adam.step, MSGM_higherDim.py:803-809) for two iterations
    on the 2-D U-Net at 32x32 (fused dual attention at T = 256, C = 64 and T = 64, C = 128) with the WELL-CONDITIONED fill,
    against the oracle doing the same on the CPU.  With the sinusoidal fill of test_unet2d_gpu.py this comparison is the
    loosest pin of the suite (loss sequence 2.4e-4); here the loss sequence is held to 1e-5 — it sees every parameter that
    matters after each update.  The measured value moves with ROUNDING on either side, because Adam's first steps are
    lr * sign(g) wherever |g| >> eps: 5.6e-7 with the oracle on all host threads, 3.6e-6 with the oracle on 16 threads (an
    earlier test of the suite sets that; the CPU convolutions then add in another order) and the pixel-streaming 1x1 weight-gradient
    kernel of r3 — same HIP bits in both runs (tools/debug_poison.py: no kernel of the pass reads memory it did not write).
    The oracle's thread count is pinned here so that the number does not depend on which tests ran before.  (Parameters themselves: tensors with an analytically zero gradient — conv biases in front
    of a GroupNorm — take +-lr rounding-noise steps under Adam in ANY implementation, so they are bounded, not compared.)"""
    from conftest import within
    from oracle import sde_ref as S, nets_ref as N, ssm_ref as LR
    from oracle.det_params import load_init_like_
    from sdeflow_light_amd.NNUnet import VorticityUNet
    torch.set_num_threads(min(16, torch.get_num_threads()))
    torch.manual_seed(8)
    net = VorticityUNet(base_channels=32, channel_mults=(1, 2, 4), num_res_blocks=2, premodule=None, in_space=32,
                        attention_resolutions=(2, 4), flatten_order="F")
    load_init_like_(net)
    gen = make_gen("sgm", net)
    opt = torch.optim.Adam(gen.a.parameters(), lr=1e-4)
    B, d = 2, 1024
    cfg = N.UNet2DConfig(in_space=32)
    ref = {k: v.detach().cpu().clone() for k, v in net.named_parameters()}
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
            ref[k], m[k], vv[k] = LR.adam_step(ref[k], gref[k], m[k], vv[k], it + 1, lr=1e-4)
    within(max(abs(a_ - b_) / abs(b_) for a_, b_ in zip(losses, losses_ref)), 1e-5,
           "2 Adam steps, 2-D U-Net 32x32, well-conditioned fill: loss sequence rel. error")          # measured 3.6e-06 (see above)
    flat = torch.cat([p_.detach().reshape(-1).cpu() for _, p_ in net.named_parameters()])
    flat_ref = torch.cat([ref[k].reshape(-1) for k, _ in net.named_parameters()])
    assert float((flat - flat_ref).abs().max()) <= 2 * 2 * 1e-4 + 1e-7          # nobody moved further than Adam allows
    print(f"  parameters after 2 steps: rel-L2 {rel_l2(flat, flat_ref):.2e}, max abs diff {float((flat - flat_ref).abs().max()):.2e}")


def test_sampler_bf16_split_opt_in_matches_the_fp32_sampler(monkeypatch):
    """Opt-in experiment: MSGM_SAMPLER_BF16X3=1 sends the sampler's 3x3 convolutions through the
    bf16-split kernel (six bf16 MFMA products per fp32 product, fp32 accumulate).  A 16-step Euler-Maruyama run through the
    2-D U-Net (32x32, attention at 16x16 and 8x8) must stay fp32-grade: against the default sampler (Winograd convs) with the
    same injected noise, far inside north_star's 1e-4."""
    from sdeflow_light_amd import sde_scheme as SS
    from oracle.det_params import load_init_like_
    from sdeflow_light_amd.NNUnet import VorticityUNet
    torch.manual_seed(21)
    net = VorticityUNet(base_channels=32, channel_mults=(1, 2, 4), num_res_blocks=2, premodule=None, in_space=32,
                        attention_resolutions=(2, 4), flatten_order="F")
    load_init_like_(net)
    gen = make_gen("sgm", net)
    B, n, steps = 4, 1024, 16
    x0, z = torch.randn(B, n), torch.randn(steps, B, n)
    ref = SS.euler_maruyama_sampler(gen, x0.to(DEV), num_steps=steps, keep_all_samples=True, include_t0=True, noise=z)
    monkeypatch.setenv("MSGM_SAMPLER_BF16X3", "1")
    got = SS.euler_maruyama_sampler(gen, x0.to(DEV), num_steps=steps, keep_all_samples=True, include_t0=True, noise=z)
    monkeypatch.delenv("MSGM_SAMPLER_BF16X3")
    back = SS.euler_maruyama_sampler(gen, x0.to(DEV), num_steps=steps, keep_all_samples=True, include_t0=True, noise=z)
    e = [rel_l2(got[i], ref[i]) for i in (1, 4, 8, 16)]
    print("bf16-split sampler vs fp32 sampler, 16-step U-Net EM, rel-L2 at steps 1,4,8,16: " + " ".join(f"{v:.1e}" for v in e))
    assert not torch.equal(torch.as_tensor(got[16]), torch.as_tensor(ref[16]))      # the switch took the other kernels
    assert max(e) <= 1e-5
    assert torch.equal(torch.as_tensor(back[16]), torch.as_tensor(ref[16]))         # and switching it off restores the default bits
