"""GPU parity of the host mirrors (same call surface as the reference's
NN / SDEs / sde_scheme) against the golden vectors recorded from the reference.
Sampler tolerance: 1e-4 rel-L2 (north_star); index tensors bit-exact."""
import pytest
import torch

from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def Tp():
    return torch.nn.Parameter(torch.FloatTensor([1.0]), requires_grad=False)


def make_gen(base_kind, net, g=None, prefix=None, nsf=16, G=None, n=None):
    from sdeflow_light_amd.SDEs import SGMsde, MSGMsde, PluginReverseSDE
    T = Tp()
    if base_kind == "sgm":
        base = SGMsde(beta_min=0.1, beta_max=20.0, t_epsilon=1e-3, T=T, num_steps_forward=nsf, device=DEV)
    else:
        base = MSGMsde(torch.randn(64, n), beta_min=0.1, beta_max=20.0, t_epsilon=1e-3, T=T, num_steps_forward=nsf,
                       device=DEV, denseTensor=(base_kind == "dense"), norm_map="log", G=G)
    gen = PluginReverseSDE(base, net.to(DEV), T, deviceReverseSDE=DEV).to(DEV)
    if g is not None:
        sd = {k: v for k, v in g.sub(prefix).items()}
        missing = gen.load_state_dict(sd, strict=False)
        assert set(missing.missing_keys) <= {"T", "base_sde.T"} and not missing.unexpected_keys
    return gen


def test_mlp_module_forward_golden():
    from sdeflow_light_amd.NN import MLP
    g = load_golden("g09_mlp")
    for tag, d, pre in (("mlp2", 2, None), ("mlp6n", 6, "NormalizeLogRadius")):
        net = MLP(d, premodule=pre).to(DEV)
        net.load_state_dict(g.sub(tag + "::"))
        out = net(g[tag + "_x"].to(DEV), g[tag + "_t"].to(DEV))
        assert rel_l2(out.cpu(), g[tag + "_out"]) <= 1e-5


@pytest.mark.parametrize("tag,d,pre", [("mlp2", 2, None), ("mlp6n", 6, "NormalizeLogRadius")])
def test_ssm_reference_loop_surface(tag, d, pre):
    """gen_sde.ssm(x).mean().backward() as the driver writes it (MSGM_higherDim.py:807-808)."""
    from sdeflow_light_amd.NN import MLP
    g = load_golden("g10_ssm_mlp")
    gen = make_gen("sgm", MLP(d, premodule=pre), g, tag + "::")
    gen.zero_grad()
    per = gen.ssm(g[tag + "_x"].to(DEV), u=g[tag + "_u_t"].reshape(-1).to(DEV), eps=g[tag + "_eps"].to(DEV),
                  u_v=g[tag + "_u_v"].to(DEV))
    assert rel_l2(per.detach().cpu(), g[tag + "_per"]) <= 1e-5
    per.mean().backward()
    from conftest import within
    within(max(rel_l2(p.grad.cpu(), g[f"{tag}_grad::{k}"]) for k, p in gen.named_parameters() if p.requires_grad), 5e-7,
           f"MLP {tag}: worst per-tensor gradient rel-L2 vs the reference's double backward")
    # a second backward pass accumulates (autograd semantics)
    per2 = gen.ssm(g[tag + "_x"].to(DEV), u=g[tag + "_u_t"].reshape(-1).to(DEV), eps=g[tag + "_eps"].to(DEV),
                   u_v=g[tag + "_u_v"].to(DEV))
    per2.sum().backward()
    B = per.shape[0]
    k = "a.main.2.weight"
    assert rel_l2(dict(gen.named_parameters())[k].grad.cpu(), (1 + B) * g[f"{tag}_grad::{k}"]) <= 2e-4


def test_three_train_steps_golden_fused_adam():
    """3 optimizer steps (SSM + Adam) against the reference's loss sequence and final parameters (g11)."""
    from sdeflow_light_amd.NN import MLP
    from sdeflow_light_amd.optim import FusedAdam
    g = load_golden("g11_train3")
    gen = make_gen("sgm", MLP(2), g, "init::")
    opt = FusedAdam(gen.parameters(), lr=1e-3)
    for i in range(3):
        opt.zero_grad()
        loss = gen.ssm(g["x"][i].to(DEV), u=g["u_t"][i].reshape(-1).to(DEV), eps=g["eps"][i].to(DEV),
                       u_v=g["u_v"][i].to(DEV)).mean()
        loss.backward()
        opt.step()
        assert float(loss.detach()) == pytest.approx(float(g["loss"][i]), rel=2e-5)
    sd = gen.state_dict()
    for k, v in g.sub("final::").items():
        assert rel_l2(sd[k].cpu(), v) <= 2e-5, k
    osd = opt.state_dict()
    assert len(osd["state"]) == 8                                               # torch.optim.Adam layout
    assert all(float(st["step"]) == 3.0 and set(st) == {"step", "exp_avg", "exp_avg_sq"} for st in osd["state"].values())


def test_trainer_graph_equals_eager():
    from sdeflow_light_amd.NN import MLP
    from sdeflow_light_amd.train import MLPScoreTrainer
    from sdeflow_light_amd.data import gaussian_mixture_2d
    outs = []
    for use_graph in (False, True):
        torch.manual_seed(0)
        gen = make_gen("sgm", MLP(2))
        tr = MLPScoreTrainer(gen, 4096, lr=1e-3, use_graph=use_graph, seed=3)
        tr.set_data(gaussian_mixture_2d(4096, device=DEV))
        losses = [float(tr.step()) for _ in range(6)]
        outs.append((losses, tr.flat.clone()))
        if use_graph:
            from sdeflow_light_amd import ops
            kinds = ops.graph_node_kinds(tr.graph)
            print(f"captured MLP train step = {kinds}")
            assert set(kinds) == {"kernel"}, kinds
    # ONE optimizer update per step() call in both modes (capture()'s warm-up is the first call's step): the two runs
    # match step for step, and end with the same parameters
    assert all(abs(l) < 1e6 for l in outs[0][0])
    assert outs[0][0] == pytest.approx(outs[1][0], rel=1e-6)
    e = rel_l2(outs[1][1], outs[0][1])
    print(f"graph vs eager parameters after 6 steps: rel-L2 {e:.2e}")
    assert e <= 1e-6


def _mlp_gen(g, prefix, d, pre, kind, **kw):
    from sdeflow_light_amd.NN import MLP
    return make_gen(kind, MLP(d, premodule=pre), g, prefix, **kw)


def test_samplers_sgm_golden():
    from sdeflow_light_amd import sde_scheme as SS
    g = load_golden("g07_samplers")
    gen = _mlp_gen(g, "sgm::", 2, None, "sgm")
    x0 = g["sgm_x0"]
    for tag, fn in (("em", SS.euler_maruyama_sampler), ("heun", SS.heun_sampler), ("rk4", SS.rk4_stratonovich_sampler)):
        z = g[f"sgm_{tag}_z"]
        xs = fn(gen, x0.to(DEV), num_steps=z.shape[0], keep_all_samples=True, include_t0=True, noise=z)
        assert xs.device.type == "cpu" and xs.shape == g[f"sgm_{tag}_traj"].shape
        assert rel_l2(xs, g[f"sgm_{tag}_traj"]) <= 1e-4, tag
    xs = SS.euler_maruyama_sampler(gen, x0.to(DEV), num_steps=8, lmbd=0.5, keep_all_samples=True, noise=g["sgm_em_l05_z"])
    assert rel_l2(xs, g["sgm_em_l05_traj"]) <= 1e-4
    xs = SS.euler_maruyama_sampler(gen, x0.to(DEV), num_steps=8, keep_all_samples=False, noise=g["sgm_em_final_z"])
    assert rel_l2(xs, g["sgm_em_final"]) <= 1e-4
    xs = SS.euler_maruyama_sampler(gen, x0.to(DEV), num_steps=8, keep_all_samples=False, samplesToKeep=g["sgm_em_keep_idx"],
                                   noise=g["sgm_em_keep_z"])
    assert rel_l2(xs, g["sgm_em_keep"]) <= 1e-4
    with pytest.raises(ValueError, match="samplesToKeep"):
        SS.euler_maruyama_sampler(gen, x0.to(DEV), num_steps=8, keep_all_samples=False, samplesToKeep=[1, 2, 3])


def test_samplers_msgm_sparse_golden():
    from sdeflow_light_amd import sde_scheme as SS
    from sdeflow_light_amd.SDEs import forward_SDE
    g = load_golden("g07_samplers")
    gen = _mlp_gen(g, "sp::", 6, "NormalizeLogRadius", "sparse", n=6)
    x0 = g["sp_x0"]
    for nc in (0, 1):
        for tag, fn in (("em", SS.euler_maruyama_sampler), ("heun", SS.heun_sampler), ("rk4", SS.rk4_stratonovich_sampler)):
            z = g[f"sp_{tag}_nc{nc}_z"]
            xs = fn(gen, x0.to(DEV), num_steps=z.shape[0], keep_all_samples=True, include_t0=True,
                    norm_correction=bool(nc), noise=z)
            assert rel_l2(xs, g[f"sp_{tag}_nc{nc}_traj"]) <= 1e-4, (tag, nc)
    z = g["sp_fwd_rk4_z"]
    xs = SS.rk4_stratonovich_sampler(forward_SDE(gen.base_sde, gen.T), x0.to(DEV), 4, lmbd=0., keep_all_samples=True,
                                     include_t0=True, norm_correction=True, noise=z)
    assert rel_l2(xs, g["sp_fwd_rk4_traj"]) <= 1e-4


def test_samplers_msgm_dense_golden():
    from sdeflow_light_amd import sde_scheme as SS
    g = load_golden("g07_samplers")
    gen = _mlp_gen(g, "dn::", 4, None, "dense", n=4, G=g["dn_G"])
    for tag, fn in (("em", SS.euler_maruyama_sampler), ("rk4", SS.rk4_stratonovich_sampler)):
        z = g[f"dn_{tag}_z"]
        xs = fn(gen, g["dn_x0"].to(DEV), num_steps=z.shape[0], keep_all_samples=True, include_t0=True,
                norm_correction=True, noise=z)
        assert rel_l2(xs, g[f"dn_{tag}_traj"]) <= 1e-4, tag


@pytest.mark.parametrize("tag,kind,n", [("sp", "sparse", 6), ("dn", "dense", 4)])
def test_msgm_forward_perturb_golden(tag, kind, n):
    """Device-resident masked RK4 (no per-row Python loop) vs SDE.sample_scheme (g08)."""
    from sdeflow_light_amd.SDEs import MSGMsde
    from sdeflow_light_amd import ops
    g = load_golden("g08_sample_scheme")
    T = Tp()
    base = MSGMsde(torch.randn(64, n), T=T, num_steps_forward=4, device=DEV, denseTensor=(kind == "dense"),
                   norm_map="log", G=g.get("dn_G") if kind == "dense" else None).to(DEV)
    t = g[f"{tag}_t"].to(DEV)
    k = ops.forward_step_index(t, 4, 1.0)
    assert torch.equal(k.cpu(), g[f"{tag}_k"])                                  # bit-exact timestep indexing
    y = base.sample(t, g[f"{tag}_x0"].to(DEV), noise_main=g[f"{tag}_z_main"], noise_short=g[f"{tag}_z_short"])
    assert rel_l2(y.cpu(), g[f"{tag}_y"]) <= 1e-4
    y2 = base.sample(t, g[f"{tag}_x0"].to(DEV))                                 # Philox path: fresh noise every call
    y3 = base.sample(t, g[f"{tag}_x0"].to(DEV))
    assert torch.isfinite(y2).all() and y2.shape == y.shape and not torch.equal(y2, y3)


def test_latent_samples():
    from sdeflow_light_amd.SDEs import MSGMsde, SGMsde
    g = load_golden("g13_misc")
    T = Tp()
    ms = MSGMsde(g["lat_xinit"], T=T, denseTensor=False, norm_map="log", device=DEV).to(DEV)
    assert rel_l2(ms.r_T.cpu(), g["lat_rT"]) <= 1e-6
    x0 = ms.latent_sample(40, 6, u=g["lat_u"].to(DEV), z=g["lat_z"].to(DEV))
    assert rel_l2(x0.cpu(), g["lat_x0"]) <= 1e-5
    x1 = ms.latent_sample(1000, 6)
    assert torch.isfinite(x1).all() and x1.shape == (1000, 6)
    s = SGMsde(T=T, device=DEV)
    z = s.latent_sample(1 << 16, 4)
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1) < 0.02
    assert not torch.equal(z, s.latent_sample(1 << 16, 4))


def test_graphed_sampler_equals_eager():
    from sdeflow_light_amd.NN import MLP
    from sdeflow_light_amd import sde_scheme as SS
    from sdeflow_light_amd import _lib
    torch.manual_seed(1)
    gen = make_gen("sgm", MLP(2))
    B, N = 4096, 25
    x0 = torch.randn(B, 2, device=DEV)
    gs = SS.GraphedEMSampler(gen, B, N)
    st = gen.base_sde.rng.state.clone()
    a = gs.run(x0).clone()
    gen.base_sde.rng.state.copy_(st)
    b = SS.euler_maruyama_sampler(gen, x0, num_steps=N, keep_all_samples=False)
    assert rel_l2(a.cpu(), b) <= 1e-6
    c = gs.run(x0).clone()                                    # offset advanced inside the graph: fresh noise
    assert not torch.equal(a, c) and torch.isfinite(c).all()


@pytest.mark.parametrize("B,d,pre,N", [(65, 2, None, 7), (1000, 2, None, 12), (4100, 6, "NormalizeLogRadius", 5),
                                      (300, 20, None, 4), (32, 2, None, 3), (70000, 2, None, 3)])
def test_em_loop_one_launch_equals_per_step_kernels(B, d, pre, N):
    """msgm_mlp_em_loop (all N steps in one launch, each workgroup carrying its rows through every step) must give
    the numbers of N successive msgm_mlp_em_step launches: same time grid, same Philox draws.  Ragged B, several
    tiles per workgroup, wide and premodule variants; B <= 32 is refused and the sampler falls back."""
    from sdeflow_light_amd.NN import MLP
    from sdeflow_light_amd import ops
    from sdeflow_light_amd._lib import MsgmError
    torch.manual_seed(B + d)
    gen = make_gen("sgm", MLP(d, premodule=pre))
    base = gen.base_sde
    P, st = gen.a.kernel_params(), base.struct()
    T = base.T_float()
    ts = torch.linspace(0, 1, N + 1) * T
    delta = T / N
    x0 = torch.randn(B, d, device=DEV)
    rng = base.philox(DEV)
    ref = x0.clone()
    for i in range(N):
        ops.mlp_em_step(P, ref, st, ts[i].item(), delta, 0.0, rng=rng, rng_step=i)
    one = x0.clone()
    if B <= 32:
        with pytest.raises(MsgmError):
            ops.mlp_em_loop(P, one, st, ts.to(DEV), delta, 0.0, rng, 0)
        return
    ops.mlp_em_loop(P, one, st, ts.to(DEV), delta, 0.0, rng, 0)
    assert torch.isfinite(one).all()
    assert rel_l2(one.cpu(), ref.cpu()) <= 1e-6


def test_trainer_fused_launches_match_separate_kernels():
    """prep (K1+probe+tick) and reduce+Adam fused launches == the separate C-ABI kernels, two steps, ragged batch."""
    from sdeflow_light_amd.NN import MLP
    from sdeflow_light_amd.train import MLPScoreTrainer
    from sdeflow_light_amd.data import gaussian_mixture_2d
    from sdeflow_light_amd import ops, _lib
    torch.manual_seed(0)
    B = 1003
    gen = make_gen("sgm", MLP(2))
    tr = MLPScoreTrainer(gen, B, lr=1e-3, use_graph=False, seed=5)
    x = gaussian_mixture_2d(B, device=DEV)
    tr.set_data(x)
    p = tr.flat.clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    rng = _lib.PhiloxState(5 * 1000003 + 17, DEV)
    st = gen.base_sde.struct()
    ws = ops.mlp_ssm_workspace(2, False, DEV)
    shapes = [(128, 3), (128,), (128, 128), (128,), (128, 128), (128,), (2, 128), (2,)]
    for step in (1, 2):
        off, views = 0, []
        for s in shapes:
            k = 1
            for q in s:
                k *= q
            views.append(p[off:off + k].view(s)); off += k
        P = ops.mlp_params(*views, premodule=False)
        y, t = ops.perturb_vp(x, st, rng=rng)
        vv = ops.rademacher((B, 2), DEV, rng=rng)
        g = torch.empty_like(p)
        lsum = torch.empty(1, device=DEV)
        ops.mlp_ssm_grad(P, y, t, vv, st, 1.0 / B, g, ws, loss_sum=lsum)
        ops.adam_step(p, g, m, v, step=step, lr=1e-3)
        rng.advance(1)
        loss = tr.step()
        assert float(loss) == pytest.approx(float(lsum), rel=1e-6)
        assert rel_l2(tr.gflat.cpu(), g.cpu()) <= 1e-6
        assert rel_l2(tr.flat.cpu(), p.cpu()) <= 1e-7
    assert int(tr.step_dev) == 2


# ------------------------------------------------------------------ 1-D U-Net (C3 path)
def _unet1d(L, dev=DEV):
    from sdeflow_light_amd.NNUnet1D import UNet1D
    from oracle.det_params import load_det_
    net = UNet1D(input_dim=L, base_channels=32, channel_mults=(1, 2, 4), num_res_blocks=2, premodule=None, emb_dim=128)
    load_det_(net)
    return net.to(dev)


def test_unet1d_state_dict_keys_match_reference_layout():
    from oracle.shapes import unet1d_shapes
    net = _unet1d(64, "cpu")
    want = unet1d_shapes(64, None)
    got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert got == {k: tuple(v) for k, v in want.items()}
    assert sum(v.numel() for v in net.state_dict().values()) == 819265


@pytest.mark.parametrize("tag,L", [("u1d", 1024), ("u1d_odd", 1001), ("u1d_small", 64)])
def test_unet1d_forward_golden(tag, L):
    g = load_golden("g09_unet1d")
    net = _unet1d(L)
    out = net(g[tag + "_x"].to(DEV), g[tag + "_t"].to(DEV))
    assert out.shape == g[tag + "_out"].shape
    assert rel_l2(out.cpu(), g[tag + "_out"]) <= 1e-4, rel_l2(out.cpu(), g[tag + "_out"])


def test_unet1d_ssm_golden():
    """Per-sample SSM loss and every parameter gradient (digest) vs the reference's double-backward (g10)."""
    from conftest import check_digest as _check_digest
    g = load_golden("g10_ssm_unets")
    net = _unet1d(256)
    gen = make_gen("sgm", net)
    gen.zero_grad()
    per = gen.ssm(g["u1d_x"].to(DEV), u=g["u1d_u_t"].reshape(-1).to(DEV), eps=g["u1d_eps"].to(DEV), u_v=g["u1d_u_v"].to(DEV))
    from conftest import within
    within(rel_l2(per.detach().cpu(), g["u1d_per"]), 1e-7, "UNet1D L=256 per-sample SSM loss vs reference")
    per.mean().backward()
    grads = {k: p.grad.cpu() for k, p in gen.a.named_parameters()}
    _check_digest(g, "u1d", grads, "a.", 5e-7)            # measured 2.4e-07 (r2)


def test_unet_ssm_backward_respects_zero_grad_and_accumulation():
    """The reference loop is zero_grad(); ssm(x).mean().backward(); step() (MSGM_higherDim.py:803-809): a second
    iteration must start from zero after zero_grad() (set_to_none), and accumulate when zero_grad() is not called."""
    torch.manual_seed(3)
    net = _unet1d(64)
    gen = make_gen("sgm", net)
    B, d = 3, 64
    x, u, eps, uv = torch.randn(B, d).to(DEV), torch.rand(B).to(DEV), torch.randn(B, d).to(DEV), torch.rand(B, d).to(DEV)

    def grads():
        return torch.cat([p.grad.reshape(-1) for p in gen.a.parameters()]).cpu().clone()

    gen.zero_grad()
    gen.ssm(x, u=u, eps=eps, u_v=uv).mean().backward()
    g1 = grads()
    gen.zero_grad()
    gen.ssm(x, u=u, eps=eps, u_v=uv).mean().backward()
    g2 = grads()
    assert float(g1.abs().max()) > 0 and rel_l2(g2, g1) <= 1e-5, rel_l2(g2, g1)
    gen.ssm(x, u=u, eps=eps, u_v=uv).mean().backward()          # no zero_grad: accumulates
    assert rel_l2(grads(), 2.0 * g1) <= 1e-5


def test_unet1d_reference_loop_adam_steps_vs_oracle():
    """The reference's loop body (MSGM_higherDim.py:803-809) through the mirror API with torch.optim.Adam, three
    iterations on a 1-D U-Net, against the oracle doing the same on the CPU: the loss of iteration k+1 only matches if
    the kernels see the parameters the optimizer just wrote (weight images are re-packed per forward)."""
    from oracle import sde_ref as S, nets_ref as N, ssm_ref as LR
    torch.manual_seed(21)
    L_, B = 64, 4
    net = _unet1d(L_)
    gen = make_gen("sgm", net)
    opt = torch.optim.Adam(gen.a.parameters(), lr=1e-3)
    ref = {k: v.detach().cpu().clone() for k, v in net.named_parameters()}
    m = {k: torch.zeros_like(v) for k, v in ref.items()}
    vv = {k: torch.zeros_like(v) for k, v in ref.items()}
    sp = S.SdeSpec()
    score = lambda prm, yy, tt: N.unet1d_forward(prm, yy, tt)
    losses, losses_ref = [], []
    for it in range(3):
        x, u, eps, uv = torch.randn(B, L_), torch.rand(B), torch.randn(B, L_), torch.rand(B, L_)
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
    within(max(abs(a_ - b_) / abs(b_) for a_, b_ in zip(losses, losses_ref)), 1e-6, "3 Adam steps, UNet1D: loss sequence rel. error")
    # the iterations are different problems, so matching losses 2 and 3 needs the updated parameters
    assert abs(losses_ref[1] - losses_ref[0]) > 1e-2 * abs(losses_ref[0])
    flat = torch.cat([p_.detach().reshape(-1).cpu() for _, p_ in net.named_parameters()])
    flat_ref = torch.cat([ref[k].reshape(-1) for k, _ in net.named_parameters()])
    within(rel_l2(flat, flat_ref), 6e-5, "3 Adam steps, UNet1D: parameters rel-L2")


def test_unet1d_sampler_runs_and_matches_oracle():
    from sdeflow_light_amd import sde_scheme as SS
    from oracle import sde_ref as S, nets_ref as N
    torch.manual_seed(0)
    L = 64
    net = _unet1d(L)
    gen = make_gen("sgm", net)
    x0, z = torch.randn(5, L), torch.randn(6, 5, L)
    xs = SS.euler_maruyama_sampler(gen, x0.to(DEV), num_steps=6, keep_all_samples=False, noise=z)
    p = {k: v.cpu() for k, v in net.state_dict().items()}
    ref = S.euler_maruyama(S.ReverseProcess(S.SdeSpec(), lambda y, s: N.unet1d_forward(p, y, s)), x0, 6, z)
    assert rel_l2(xs, ref) <= 1e-4, rel_l2(xs, ref)


# ------------------------------------------------------------------ multiplicative SDE training (MSGM)
@pytest.mark.parametrize("tag,kind,d,pre", [("sp", "sparse", 6, "NormalizeLogRadius"), ("dn", "dense", 4, None)])
def test_ssm_msgm_mlp_golden(tag, kind, d, pre):
    """SSM loss + parameter gradients for the multiplicative SDE (u = G(y)^T v) vs the reference (g14)."""
    from sdeflow_light_amd.NN import MLP
    g = load_golden("g14_ssm_msgm")
    gen = make_gen(kind, MLP(d, premodule=pre), g, tag + "::", nsf=4, n=d, G=g.get("dn_G") if kind == "dense" else None)
    gen.zero_grad()
    per = gen.ssm_loss(g[tag + "_t"].to(DEV), g[tag + "_y"].to(DEV), g[tag + "_y"].to(DEV), u_v=g[tag + "_u_v"].to(DEV))
    assert rel_l2(per.detach().cpu(), g[tag + "_per"]) <= 2e-5, rel_l2(per.detach().cpu(), g[tag + "_per"])
    per.mean().backward()
    for k, p in gen.named_parameters():
        if p.requires_grad:
            assert rel_l2(p.grad.cpu(), g[f"{tag}_grad::{k}"]) <= 3e-4, (k, rel_l2(p.grad.cpu(), g[f"{tag}_grad::{k}"]))


def test_ssm_msgm_end_to_end_runs():
    """ssm(x) for MSGM: device-resident RK4 perturbation + probe + fused kernel, Philox noise; loss finite, grads flow."""
    from sdeflow_light_amd.NN import MLP
    from sdeflow_light_amd.optim import FusedAdam
    torch.manual_seed(0)
    gen = make_gen("sparse", MLP(6, premodule="NormalizeLogRadius"), n=6, nsf=8)
    opt = FusedAdam(gen.parameters(), lr=1e-3)
    x = torch.randn(256, 6, device=DEV)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = gen.ssm(x).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(abs(l) < 1e6 and l == l for l in losses) and len(set(losses)) == 3


def test_ssm_unet1d_msgm_sparse_vs_oracle():
    """U-Net + multiplicative SDE through the general (u, cst) loss form, against the CPU oracle."""
    from oracle.shapes import unet1d_shapes
    from oracle import sde_ref as S, nets_ref as N, ssm_ref as LR
    from oracle.det_params import det_state_dict
    torch.manual_seed(0)
    L_ = 64
    net = _unet1d(L_)
    gen = make_gen("sparse", net, n=L_, nsf=4)
    B = 3
    t, y, uv = torch.rand(B, 1).clamp_min(1e-3), torch.randn(B, L_), torch.rand(B, L_)
    gen.zero_grad()
    per = gen.ssm_loss(t.to(DEV), y.to(DEV), y.to(DEV), u_v=uv.to(DEV))
    per.mean().backward()
    sp = S.SdeSpec(kind=S.MSGM_SPARSE, n=L_)
    p = det_state_dict(unet1d_shapes(L_, None))
    score = lambda prm, yy, tt: N.unet1d_forward(prm, yy, tt, None)
    loss, per_ref, gref = LR.ssm_mean_and_grads(sp, score, p, t, y, S.rademacher_from_uniform(uv))
    from conftest import within
    within(rel_l2(per.detach().cpu(), per_ref), 2e-6, "MSGM sparse + UNet1D: per-sample loss rel-L2")
    flat = torch.cat([pp.grad.reshape(-1).cpu() for _, pp in gen.a.named_parameters()])
    ref = torch.cat([gref[k].reshape(-1) for k, _ in gen.a.named_parameters()])
    within(rel_l2(flat, ref), 2e-6, "MSGM sparse + UNet1D: flat gradient rel-L2")


def test_checkpoint_roundtrip_and_torch_adam_compat(tmp_path):
    """N3: save_checkpoint / load_checkpoint keep the reference's dictionary layout (NN.py:13-42); the optimizer
    state loads into torch.optim.Adam and back, and a resumed run continues bit-identically."""
    from sdeflow_light_amd.NN import MLP, save_checkpoint, load_checkpoint
    from sdeflow_light_amd.optim import FusedAdam
    torch.manual_seed(0)
    x = torch.randn(512, 2, device=DEV)
    noise = [(torch.rand(512, device=DEV), torch.randn(512, 2, device=DEV), torch.rand(512, 2, device=DEV)) for _ in range(4)]

    def run(gen, opt, steps):
        for u, e, uv in steps:
            opt.zero_grad()
            gen.ssm(x, u=u, eps=e, u_v=uv).mean().backward()
            opt.step()
    torch.manual_seed(1)
    gen = make_gen("sgm", MLP(2))
    opt = FusedAdam(gen.parameters(), lr=1e-3)
    run(gen, opt, noise[:2])
    path = str(tmp_path / "ck.pt")
    save_checkpoint(path, gen, opt, 1)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"iteration", "model", "optimizer", "torch_rng", "numpy_rng", "python_rng", "msgm_hip"}   # reference keys + ONE
    assert {"T", "base_sde.T", "a.main.0.weight", "a.main.6.bias"} <= set(ck["model"])
    # the optimizer state is torch.optim.Adam's: it loads into a stock Adam over same-shaped parameters
    ref_params = [torch.nn.Parameter(p_.detach().cpu().clone(), requires_grad=p_.requires_grad) for p_ in gen.parameters()]
    ref_opt = torch.optim.Adam(ref_params, lr=1e-3)
    ref_opt.load_state_dict(ck["optimizer"])
    assert len(ref_opt.state) == 8 and all(int(st["step"]) == 2 for st in ref_opt.state.values())
    run(gen, opt, noise[2:])
    final = gen.a.flat_parameters()[0].clone()
    # resume from the checkpoint in a fresh object graph
    torch.manual_seed(2)
    gen2 = make_gen("sgm", MLP(2))
    opt2 = FusedAdam(gen2.parameters(), lr=1e-3)
    it = load_checkpoint(path, gen2, opt2, DEV)
    assert it == 1
    run(gen2, opt2, noise[2:])
    assert torch.equal(gen2.a.flat_parameters()[0], final)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_mmd_golden(tag):
    """compute_kernel / compute_mmd (quantitative_comparison.py:22-46) on the device vs the reference's values
    (g15, generated by importing the reference) — fp32, 1e-5."""
    from sdeflow_light_amd.quantitative_comparison import compute_kernel, compute_mmd
    g = load_golden("g15_metrics")
    x, y = g[f"mmd_{tag}_x"], g[f"mmd_{tag}_y"]
    K = compute_kernel(x.to(DEV), y.to(DEV))
    assert K.shape == (x.shape[0], y.shape[0])
    assert rel_l2(K.cpu(), g[f"mmd_{tag}_Kxy"]) <= 1e-5
    assert abs(float(compute_mmd(x.to(DEV), y.to(DEV))) - float(g[f"mmd_{tag}"])) <= 1e-5
    assert abs(float(compute_mmd(x, y)) - float(g[f"mmd_{tag}"])) <= 1e-5          # host tensors are moved over


def test_mmd_large_matches_oracle_and_properties():
    """Ragged sizes across several tiles; MMD(x, x) = 0 and symmetry (size-independent properties)."""
    from oracle import metrics_ref as M
    from sdeflow_light_amd.quantitative_comparison import compute_mmd
    torch.manual_seed(3)
    x, y = torch.randn(1000, 5), torch.randn(777, 5) * 0.7 + 0.3
    a = float(compute_mmd(x.to(DEV), y.to(DEV)))
    assert abs(a - float(M.mmd(x, y))) <= 1e-5
    assert abs(a - float(compute_mmd(y.to(DEV), x.to(DEV)))) <= 1e-6
    assert abs(float(compute_mmd(x.to(DEV), x.to(DEV)))) <= 1e-6


def test_elbo_and_evaluate_golden():
    """elbo_random_t_slice (SDEs.py:708-721) with the reference's recorded draws, and NN.evaluate's contract."""
    from sdeflow_light_amd.NN import MLP, evaluate
    g = load_golden("g15_metrics")
    gen = make_gen("sgm", MLP(2), g, "elbo::")
    assert rel_l2(gen.base_sde.log_latent_pdf(g["lp_y"].to(DEV)).cpu(), g["lp"]) <= 1e-6
    elbo = gen.elbo_random_t_slice(g["elbo_x"].to(DEV), u=g["elbo_draw0_rand"].to(DEV), eps=g["elbo_draw1_randn_like"].to(DEV),
                                   u_v=g["elbo_draw2_rand"].to(DEV), eps_T=g["elbo_draw5_randn_like"].to(DEV))
    assert rel_l2(elbo.cpu(), g["elbo"]) <= 1e-4
    mean, se = evaluate(gen, g["elbo_x"].to(DEV))
    assert mean.dim() == 0 and se.dim() == 0 and torch.isfinite(mean) and float(se) > 0
    assert gen.training


def test_reference_api_surface_wrappers():
    """Names the reference exposes next to the hot path and that a driver may call (SDEs.py:78-146,171-175,343-367,
    560-580,695-706): thin wrappers over the same kernels — checked against the accessor-level formulas."""
    from sdeflow_light_amd.NN import MLP
    torch.manual_seed(5)
    gen = make_gen("sgm", MLP(2))
    base = gen.base_sde
    B = 64
    x = torch.randn(B, 2, device=DEV)
    t = torch.rand(B, 1, device=DEV).clamp_min(1e-3)
    eps = torch.randn(B, 2, device=DEV)
    y, e, std, g = base.sample_Song_et_al(t, x, return_noise=True, eps=eps)
    assert torch.equal(e, eps) and rel_l2(y.cpu(), base.sample(t, x, eps=eps).cpu()) == 0.0
    assert rel_l2(std.cpu(), (base.var(t) ** 0.5).cpu()) == 0.0 and g.shape == y.shape
    assert rel_l2(y.cpu(), (base.mean_weight(t) * x + std * eps).cpu()) <= 1e-6
    lv, mn = base.logvar_mean_T
    assert float(lv) == 0.0 and float(mn) == 0.0
    # reverse drift pieces: mu(t) = ga_m_drift(T - t); ga = g . a
    s = torch.rand(B, 1, device=DEV) * 0.8 + 0.1
    yy = torch.randn(B, 2, device=DEV)
    a = gen.a(yy, s.reshape(-1))
    assert rel_l2(gen.ga(s, yy).cpu(), (base.g(s, yy) * a).cpu()) <= 1e-5
    lm = 0.3
    ref = (1 - 0.5 * lm) * base.g(s, yy) * a - base.f(s, yy) + (1 - lm) * base.div_Sigma(s, yy)
    assert rel_l2(gen.ga_m_drift(s, yy, lm).cpu(), ref.cpu()) <= 1e-5
    su = torch.full((B, 1), 0.37, device=DEV)                        # uniform time: the fused stage kernel agrees
    assert rel_l2(gen.mu(1.0 - su, yy, lm).cpu(), gen.ga_m_drift(su, yy, lm).cpu()) <= 1e-5
    tl, mask = gen.sample_t_linspace(x)
    nsf = base.num_steps_forward
    full = torch.linspace(1.0 / nsf, 1.0, nsf)
    assert torch.equal(mask.cpu(), full <= base.t_epsilon) and rel_l2(tl.cpu(), full[full > base.t_epsilon]) <= 1e-7
    # multiplicative SDE: sample_scheme is the device-resident forward perturbation, sample_scheme_allt the RK4 path
    gm = make_gen("sparse", MLP(6, premodule="NormalizeLogRadius"), n=6, nsf=4)
    bm = gm.base_sde
    x6 = torch.randn(10, 6, device=DEV)
    t6 = torch.tensor([[0.05], [0.3], [0.5], [0.26], [0.76], [1.0], [0.1], [0.9], [0.25], [0.6]], device=DEV)
    st = bm.philox(DEV).state.clone()
    ya = bm.sample_scheme(t6, x6, keep_all_samples=False)
    bm.rng.state.copy_(st)
    yb = bm.sample(t6, x6)
    assert torch.equal(ya, yb)
    traj = bm.sample_scheme_allt(x6, include_t0=True, keep_all_samples=True)
    assert tuple(traj.shape) == (5, 10, 6) and rel_l2(traj[0], x6.cpu()) == 0.0
    Gd = bm.sparse_G_full(6)
    dense = torch.einsum("ijk,bj->bik", Gd, x6)                       # g with beta = 1
    v = torch.randn(10, 6, device=DEV)
    I, J, K = bm.IJK()
    sp = torch.zeros(10, 6, device=DEV).scatter_add_(1, I.unsqueeze(0).expand(10, -1), (bm.G_V * x6[:, J]) * v[:, K])
    assert rel_l2(torch.einsum("bik,bk->bi", dense, v).cpu(), sp.cpu()) <= 1e-6
