"""-m gpu: round-2 parity pins.

(a) the HIP integrators driving the 2-D U-Net against trajectories recorded from the reference
    (sde_scheme.py:43-269 x NNUnet.py:195-245; tests/golden/g16_round2.npz), and a 256-step MLP Euler-Maruyama run
    (error growth over a long reverse-SDE trajectory) — tolerance 1e-4 rel-L2 (north_star);
(b) SSM loss + EVERY parameter gradient at the BASELINE shapes against the CPU oracle: VorticityUNet 64x64x3 (T = 1024
    attention, 3-channel in/out convs; config C4) and UNet1D L = 1024 (config C3), batch 2 — per tensor;
(c) the Gaussian and the sphere probe (SDEs.py:517-536) through ``ssm``.
Every test prints the rel-L2 it measured; tolerances are set to <= 2x the measured value (rounded up to one digit),
never above the north-star bound.
"""
import pytest
import torch

from conftest import load_golden, rel_l2
from sdeflow_light_amd import ops
from test_host_gpu import make_gen, _unet1d
from test_unet2d_gpu import _vunet

pytestmark = pytest.mark.gpu
DEV = "cuda"


def per_tensor_check(named_grads, ref, tol, floor_frac=1e-3, what=""):
    """Per-tensor rel-L2 of gradients.  A tensor whose reference norm is below floor_frac x the largest tensor norm
    holds rounding noise only (biases in front of a GroupNorm have an analytically zero gradient): for those the error
    is measured against the floor instead of their own norm.  Returns / prints the worst tensor."""
    top = max(float(r.double().norm()) for r in ref.values())
    floor = floor_frac * top
    worst, wname, nfloor = 0.0, "", 0
    for k, g in named_grads.items():
        r = ref[k].double().reshape(-1)
        scale = max(float(r.norm()), floor)
        nfloor += float(r.norm()) < floor
        e = float((g.double().reshape(-1).cpu() - r).norm()) / scale
        if e > worst:
            worst, wname = e, k
    print(f"{what}: worst per-tensor gradient rel-L2 {worst:.2e} ({wname}); {nfloor}/{len(ref)} tensors under the "
          f"{floor_frac:g} x max-norm floor; tolerance {tol:.1e}")
    assert worst <= tol, (wname, worst)
    return worst


# ------------------------------------------------------------------------------------------ (a) trajectories
@pytest.mark.parametrize("tag,steps,tol", [("em", 8, 1e-5), ("rk4", 4, 1e-5), ("heun", 4, 1e-5)])   # measured 5.4-6.6e-6 (Winograd sampler)
def test_unet2d_reverse_sde_trajectory_vs_reference(tag, steps, tol):
    from sdeflow_light_amd import sde_scheme as SS
    g = load_golden("g16_round2")
    gen = make_gen("sgm", _vunet(16, "F"))
    fn = {"em": SS.euler_maruyama_sampler, "rk4": SS.rk4_stratonovich_sampler, "heun": SS.heun_sampler}[tag]
    xs = fn(gen, g["u2d_x0"].to(DEV), num_steps=steps, keep_all_samples=True, include_t0=True, noise=g[f"u2d_{tag}_z"])
    ref = g[f"u2d_{tag}_traj"]
    assert xs.shape == ref.shape and xs.device.type == "cpu"
    e, e_last = rel_l2(xs, ref), rel_l2(xs[-1], ref[-1])
    print(f"HIP vs reference, VorticityUNet 16x16 {tag} x{steps}: trajectory rel-L2 {e:.2e}, final state {e_last:.2e}")
    assert e <= tol and e_last <= tol


def test_mlp_em_256_steps_vs_reference():
    from sdeflow_light_amd import sde_scheme as SS
    from sdeflow_light_amd.NN import MLP
    g = load_golden("g16_round2")
    gen = make_gen("sgm", MLP(2), g, "mlp::")
    xs = SS.euler_maruyama_sampler(gen, g["mlp_x0"].to(DEV), num_steps=256, keep_all_samples=True, include_t0=True,
                                   noise=g["mlp_em256_z"])
    ref = g["mlp_em256_every32"]
    e = [rel_l2(xs[32 * i], ref[i]) for i in range(ref.shape[0])]
    print("HIP vs reference, 256-step MLP EM, rel-L2 at steps 0,32,..,256: " + " ".join(f"{v:.1e}" for v in e))
    assert max(e) <= 2e-7


# ------------------------------------------------------------------------------------------ (b) BASELINE shapes
def ssm_parity_vs_fp64(gen, score, p, x, u, eps, uv, what, grad_key=lambda k: k, **kw):
    """SGM + (u, eps, u_v) injected: HIP ``ssm`` against the float32 / float64 CPU oracle (conftest.parity_vs_fp64)."""
    from oracle import sde_ref as S, ssm_ref as LR
    from conftest import parity_vs_fp64
    B = x.shape[0]
    sp = S.SdeSpec()
    torch.set_num_threads(min(16, torch.get_num_threads()))

    def hip():
        gen.zero_grad()
        per = gen.ssm(x.to(DEV), u=u.to(DEV), eps=eps.to(DEV), u_v=uv.to(DEV))
        per.mean().backward()
        return per.detach(), {grad_key(k): pp.grad.detach() for k, pp in gen.a.named_parameters()}

    def oracle(dt):
        t = S.clamp_time(sp, u.reshape(B, 1))
        y = S.vp_perturb(sp, t, x, eps)
        v = S.rademacher_from_uniform(uv)
        _, per, g = LR.ssm_mean_and_grads(sp, score, {k: w.to(dt) for k, w in p.items()}, t.to(dt), y.to(dt), v.to(dt))
        return per, g
    return parity_vs_fp64(hip, oracle, what, **kw)


@pytest.mark.parametrize("convs", ["winograd (default)", "direct"])
def test_c4_shape_ssm_loss_and_all_gradients_vs_oracle(convs, monkeypatch):
    """VorticityUNet 64x64x3, B = 2: attention at T = 1024 (C = 64, fused dual kernel) and T = 256 (C = 128), 3-channel
    convs — the C4 network at a batch the float64 oracle evaluates in seconds.  Both convolution paths of the training step:
    the default (3x3 forward + dgrad on the Winograd kernel, r3) and the direct kernels (MSGM_TRAIN_WINO=0).  On this
    ill-conditioned det_params fill (it amplifies every rounding ~1e3x) the direct kernels land at 1.07x the fp32 oracle's
    own distance from float64 (bound 2x), the Winograd transforms — which round about twice as much per convolution, still
    in fp32 — at 2.2x (bound 2.5x; worst single tensor 1.8x the oracle's worst, bound 4x as before); the well-conditioned reference fixture (g17, test_round3_gpu.py) holds its absolute
    tolerances on the default path."""
    if convs == "direct":
        monkeypatch.setenv("MSGM_TRAIN_WINO", "0")
    from oracle import nets_ref as N
    from oracle.det_params import det_state_dict
    from oracle.shapes import unet2d_shapes
    B, d = 2, 3 * 64 * 64
    gen = make_gen("sgm", _vunet(64, "F", channels=3))
    torch.manual_seed(21)
    x, u, eps, uv = torch.randn(B, d), torch.rand(B), torch.randn(B, d), torch.rand(B, d)
    cfg = N.UNet2DConfig(in_channels=3, out_channels=3, in_space=64)
    p = det_state_dict(unet2d_shapes(cfg))
    score = lambda prm, yy, tt: N.image_to_flat(N.unet2d_core_forward(prm, N.flat_to_image(yy, 64, 64, "F", 3), tt.reshape(-1), cfg), "F")
    ssm_parity_vs_fp64(gen, score, p, x, u, eps, uv, f"C4 shape (2-D U-Net 64x64x3, B=2), {convs} convolutions",
                       grad_key=lambda k: k[len("core."):], **({} if convs == "direct" else dict(slack=2.5)))


def test_c3_shape_ssm_loss_and_all_gradients_vs_oracle():
    """UNet1D L = 1024, B = 2 (the C3 network)."""
    from oracle import nets_ref as N
    from oracle.det_params import det_state_dict
    from oracle.shapes import unet1d_shapes
    B, d = 2, 1024
    gen = make_gen("sgm", _unet1d(1024))
    torch.manual_seed(22)
    x, u, eps, uv = torch.randn(B, d), torch.rand(B), torch.randn(B, d), torch.rand(B, d)
    p = det_state_dict(unet1d_shapes(1024, None))
    ssm_parity_vs_fp64(gen, lambda prm, yy, tt: N.unet1d_forward(prm, yy, tt, None), p, x, u, eps, uv, "C3 shape (UNet1D L=1024, B=2)")


# ------------------------------------------------------------------------------------------ (c) probe types
@pytest.mark.parametrize("vt", ["gaussian", "uniform"])
def test_ssm_gaussian_and_sphere_probes_vs_reference(vt):
    from sdeflow_light_amd.NN import MLP
    from sdeflow_light_amd import _lib as L
    g = load_golden("g16_round2")
    gen = make_gen("sgm", MLP(2), g, "mlp::")
    gen.vtype = vt
    z = g[f"ssm_{vt}_zv"]
    v = z if vt == "gaussian" else z / torch.linalg.norm(z, dim=1, keepdim=True)
    gen.zero_grad()
    per = gen.ssm(g[f"ssm_{vt}_x"].to(DEV), u=g[f"ssm_{vt}_u_t"].reshape(-1).to(DEV), eps=g[f"ssm_{vt}_eps"].to(DEV), v=v.to(DEV))
    e = rel_l2(per.detach().cpu(), g[f"ssm_{vt}_per"])
    per.mean().backward()
    eg = max(rel_l2(p.grad.cpu(), g[f"ssm_{vt}_grad::{k}"]) for k, p in gen.named_parameters() if p.requires_grad)
    print(f"SSM with the {vt} probe: per-sample rel-L2 {e:.2e}, worst parameter-gradient rel-L2 {eg:.2e}")
    assert e <= 2e-7 and eg <= 5e-7
    # drawn (not injected) probes: right law, fresh every call, same Philox numbers as the standalone draw
    gen.zero_grad()
    st = gen.base_sde.philox(torch.device(DEV)).state.clone()
    p1 = gen.ssm(g[f"ssm_{vt}_x"].to(DEV)).detach().clone()
    p2 = gen.ssm(g[f"ssm_{vt}_x"].to(DEV)).detach()
    assert torch.isfinite(p1).all() and not torch.equal(p1, p2)
    assert int(gen.base_sde.rng.state[1]) > int(st[1])


# ------------------------------------------------------------------------------------------ N3: resume with device noise
@pytest.mark.parametrize("how", ["fused_adam_loop", "mlp_trainer", "unet_trainer"])
def test_resume_continues_the_philox_noise_stream(how, tmp_path):
    """Nothing injected: (t, eps, v) come from the device Philox streams.  Train 2 steps, save, train 2 more; reload in
    a fresh object graph and train 2: the parameters must be THE SAME, bit for bit.  Without the Philox state in the checkpoint the resumed run would replay
    the draws of iterations 0-1 and end elsewhere (checked)."""
    from sdeflow_light_amd.NN import MLP, save_checkpoint, load_checkpoint
    from sdeflow_light_amd.optim import FusedAdam
    from sdeflow_light_amd.train import MLPScoreTrainer, UNetScoreTrainer
    B = 512
    torch.manual_seed(0)
    d = 2 if how != "unet_trainer" else 128
    x = torch.randn(B if d == 2 else 8, d, device=DEV)

    def make(seed):
        torch.manual_seed(seed)                       # different init in the fresh graph: everything must come from the file
        if how == "unet_trainer":
            from sdeflow_light_amd.NNUnet1D import UNet1D
            gen = make_gen("sgm", UNet1D(input_dim=128, base_channels=16, channel_mults=(1, 2), emb_dim=32))
            opt = UNetScoreTrainer(gen, 8, d, lr=1e-3, seed=4)
        else:
            gen = make_gen("sgm", MLP(2))
            opt = MLPScoreTrainer(gen, B, lr=1e-3, seed=4) if how == "mlp_trainer" else FusedAdam(gen.parameters(), lr=1e-3)
        if how != "fused_adam_loop":
            opt.set_data(x)
        return gen, opt

    def run(gen, opt, n):
        for _ in range(n):
            if how == "fused_adam_loop":              # the reference loop (MSGM_higherDim.py:803-809)
                opt.zero_grad()
                gen.ssm(x).mean().backward()
                opt.step()
            else:
                opt.step()

    gen, opt = make(1)
    run(gen, opt, 2)
    path = str(tmp_path / "ck.pt")
    save_checkpoint(path, gen, opt, 2)
    run(gen, opt, 2)
    final = gen.a.flat_parameters()[0].clone()
    gen2, opt2 = make(2)
    assert load_checkpoint(path, gen2, opt2, DEV) == 2
    run(gen2, opt2, 2)
    got = gen2.a.flat_parameters()[0]
    e = rel_l2(got.cpu(), final.cpu())
    print(f"resume ({how}): parameters after 2+2 steps vs uninterrupted run: rel-L2 {e:.2e}, equal={torch.equal(got, final)}")
    assert torch.equal(got, final)          # every path is bitwise reproducible (no float atomics on the training step)
    # negative control: dropping the Philox entry replays the first draws -> a different end point
    ck = torch.load(path, map_location=DEV, weights_only=False)
    ck.pop("msgm_hip"); ck["optimizer"].pop("philox", None)
    torch.save(ck, path)
    gen3, opt3 = make(2)
    load_checkpoint(path, gen3, opt3, DEV)
    run(gen3, opt3, 2)
    assert rel_l2(gen3.a.flat_parameters()[0].cpu(), final.cpu()) > 10 * max(e, 1e-7)


# ------------------------------------------------------------------------------------------ graph-captured Heun / RK4 step
@pytest.mark.parametrize("method", ["em", "heun", "rk4"])
@pytest.mark.parametrize("base_kind", ["sgm", "sparse"])
def test_graphed_step_sampler_equals_eager_integrator(method, base_kind):
    """One hipGraph-captured step with device-side stage clocks (t, t + delta/2, t + delta), replayed N times, must give
    what the eager integrator gives from the same Philox state (the eager integrators are pinned against the reference's
    trajectories above and in test_host_gpu.py).  RK4 is what the driver generates with (MSGM_higherDim.py:903)."""
    from sdeflow_light_amd import sde_scheme as SS
    torch.manual_seed(0)
    if base_kind == "sgm":
        gen, n = make_gen("sgm", _vunet(16, "F")), 256
    else:                                                  # multiplicative SDE, sparse tensor, norm correction
        from sdeflow_light_amd.NNUnet1D import UNet1D
        from oracle.det_params import load_det_
        net = UNet1D(input_dim=64, premodule="NormalizeLogRadius")
        load_det_(net)
        gen, n = make_gen("sparse", net, n=64, nsf=4), 64
    nc = base_kind != "sgm"
    B, N = 6, 5
    x0 = torch.randn(B, n, device=DEV)
    gs = SS.GraphedStepSampler(gen, B, n, N, method=method, norm_correction=nc)
    kinds = ops.graph_node_kinds(gs.graph)
    print(f"graphed {method} ({base_kind}) step = {kinds}")
    assert set(kinds) == {"kernel"}, kinds                 # no memset / memcpy nodes (see ops.graph_node_kinds)
    st = gen.base_sde.rng.state.clone()
    a = gs.run(x0).clone()
    gen.base_sde.rng.state.copy_(st)
    fn = {"em": SS.euler_maruyama_sampler, "heun": SS.heun_sampler, "rk4": SS.rk4_stratonovich_sampler}[method]
    b = fn(gen, x0, num_steps=N, keep_all_samples=False, norm_correction=nc)
    e = rel_l2(a.cpu(), b)
    print(f"graphed {method} ({base_kind}) vs eager integrator after {N} steps: rel-L2 {e:.2e}")
    if e > 2e-6:                                           # diagnosis: which of the two is not reproducible?
        gen.base_sde.rng.state.copy_(st)
        a2 = gs.run(x0).clone()
        gen.base_sde.rng.state.copy_(st)
        b2 = fn(gen, x0, num_steps=N, keep_all_samples=False, norm_correction=nc)
        print(f"  rerun: graph vs graph {rel_l2(a2.cpu(), a.cpu()):.2e}, eager vs eager {rel_l2(b2, b):.2e}, "
              f"graph2 vs eager2 {rel_l2(a2.cpu(), b2):.2e}")
    assert e <= 2e-6
    c = gs.run(x0).clone()                                 # the stream advanced: a second run is a fresh sample
    assert not torch.equal(a, c) and torch.isfinite(c).all()


def test_sgm_sample_uses_t_as_given_below_t_epsilon():
    """SGMsde.sample(t, y0) went through the clamping kernel (t <= t_epsilon was raised to t_epsilon and
    t/T*T could move t by an ulp).  It now uses t exactly as passed, like the reference (SDEs.py:134-146)."""
    from sdeflow_light_amd.NN import MLP
    g = load_golden("g16_round2")
    gen = make_gen("sgm", MLP(2))
    y = gen.base_sde.sample(g["smallt_t"].to(DEV), g["smallt_x0"].to(DEV), eps=g["smallt_eps"].to(DEV))
    e = rel_l2(y.cpu(), g["smallt_y"])
    print(f"SGMsde.sample at t in [1e-5, 1]: rel-L2 {e:.2e}")
    # measured 2.3e-06: at t = 1e-5 the variance 1 - exp(-1e-6) keeps ~4 significant bits in fp32 (upstream's formula,
    # SDEs.py:180-181), so one ulp of difference between the two exp implementations moves sqrt(var) by percents
    assert e <= 5e-6
    y2, eps, std, gg = gen.base_sde.sample_Song_et_al(g["smallt_t"].to(DEV), g["smallt_x0"].to(DEV), return_noise=True,
                                                      eps=g["smallt_eps"].to(DEV))
    assert torch.equal(y2, y) and torch.equal(eps.cpu(), g["smallt_eps"])


def test_unet2d_long_reverse_sde_run_vs_oracle():
    """Error growth over a LONG reverse-SDE run through the 2-D U-Net: 128 Euler-Maruyama steps (the driver's default
    num_steps, MSGM_higherDim.py:108) of VorticityUNet 16x16 with injected noise, HIP vs the CPU oracle (which equals the
    reference bit for bit on the 8-step golden trajectory, tests/test_oracle_golden.py)."""
    from sdeflow_light_amd import sde_scheme as SS
    from oracle import sde_ref as S, nets_ref as N
    from oracle.det_params import det_state_dict
    from oracle.shapes import unet2d_shapes
    torch.manual_seed(7)
    B, n, steps = 3, 256, 128
    gen = make_gen("sgm", _vunet(16, "F"))
    x0, z = torch.randn(B, n), torch.randn(steps, B, n)
    xs = SS.euler_maruyama_sampler(gen, x0.to(DEV), num_steps=steps, keep_all_samples=True, include_t0=True, noise=z)
    cfg = N.UNet2DConfig(in_space=16)
    p = det_state_dict(unet2d_shapes(cfg, "core."))
    proc = S.ReverseProcess(S.SdeSpec(), lambda y, s: N.vorticity_unet_forward(p, y, s, cfg, None, "F"))
    with torch.no_grad():
        ref = S.euler_maruyama(proc, x0, steps, z, keep_all=True, include_t0=True)
    e = [rel_l2(xs[i], ref[i]) for i in (1, 16, 32, 64, 96, 128)]
    print("HIP vs oracle, 128-step U-Net EM, rel-L2 at steps 1,16,32,64,96,128: " + " ".join(f"{v:.1e}" for v in e))
    # measured (r3, embedding bank): 4.5e-06 run alone, 6.6e-06 after the other tests of this file (r2: 1.4e-06 / 3.5e-06 —
    # the det_params net amplifies a one-ulp difference in the embedding projection ~10x per evaluation); north_star bound 1e-4
    assert max(e) <= 1.5e-5
