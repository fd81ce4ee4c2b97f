"""Pin the CPU oracle (oracle/) against golden vectors produced by the actual
reference (tools/make_golden.py).  CPU only; never imports /root/reference."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2, check_digest as _check_digest
from oracle import sde_ref as S
from oracle import nets_ref as N
from oracle import ssm_ref as L
from oracle.det_params import det_state_dict, det_tensor
from oracle.shapes import unet1d_shapes, unet2d_shapes

TOL = 2e-6   # fp32 re-association slack between two CPU statements of the same formula


def spec(kind=S.SGM, **kw):
    return S.SdeSpec(kind=kind, **kw)


def close(a, b, tol=TOL):
    assert a.shape == b.shape, (a.shape, b.shape)
    assert rel_l2(a, b) <= tol, rel_l2(a, b)


def test_g01_schedule():
    g = load_golden("g01_schedule")
    sp = spec()
    assert torch.equal(S.beta(sp, g["t"]), g["beta"])
    close(S.vp_mean_weight(sp, g["t"]), g["mean_weight"], 1e-7)
    close(S.vp_var(sp, g["t"]), g["var"], 1e-7)


def test_g02_sample_t_clamp():
    g = load_golden("g02_sample_t")
    sp = spec()
    assert torch.equal(S.clamp_time(sp, g["u"]), g["t"])
    assert torch.equal(S.clamp_time(sp, g["u_edge"]), g["t_edge"])
    assert float(g["t_edge"].min()) == pytest.approx(1e-3)


@pytest.mark.parametrize("nsf", [4, 16, 128])
def test_g03_step_index_bit_exact(nsf):
    g = load_golden("g03_step_index")
    k = S.forward_step_index(spec(num_steps_forward=nsf), g[f"t_{nsf}"])
    assert k.dtype == torch.int32
    assert torch.equal(k, g[f"k_{nsf}"])


@pytest.mark.parametrize("d", [2, 1024])
def test_g04_vp_perturb(d):
    g = load_golden("g04_vp_perturb")
    y = S.vp_perturb(spec(), g[f"t_{d}"], g[f"x0_{d}"], g[f"eps_{d}"])
    close(y, g[f"y_{d}"], 1e-7)


def test_g05_drift_diffusion():
    g = load_golden("g05_drift_diffusion")
    sp = spec()
    for name, fn in (("f", S.drift_f), ("fs", S.drift_f_strato), ("div", S.div_sigma), ("g", S.diffusion_g)):
        close(fn(sp, g["sgm_t"], g["sgm_y"]), g["sgm_" + name], 1e-7)
    G = S.make_dense_G(4, g["dense_gen"])
    close(G, g["dense_G"], 1e-6)
    sp = spec(S.MSGM_DENSE, n=4, G=g["dense_G"])
    close(sp.L_G, g["dense_LG"], 1e-6)
    assert float(torch.trace(sp.L_G)) == pytest.approx(-2.0, abs=1e-5)
    for name, fn in (("f", S.drift_f), ("fs", S.drift_f_strato), ("div", S.div_sigma), ("g", S.diffusion_g)):
        close(fn(sp, g["dense_t"], g["dense_y"]), g["dense_" + name])
    for n in (6, 1024):
        I, J, K, V = S.sparse_ijkv(n)
        for a, b in ((I, "I"), (J, "J"), (K, "K")):
            assert torch.equal(a, g[f"sp{n}_{b}"])
        assert torch.equal(V, g[f"sp{n}_V"])
        sp = spec(S.MSGM_SPARSE, n=n)
        for name, fn in (("f", S.drift_f), ("div", S.div_sigma), ("g", S.diffusion_g)):
            close(fn(sp, g[f"sp{n}_t"], g[f"sp{n}_y"]), g[f"sp{n}_{name}"], 1e-7)


def test_g06_emstep_three_layouts():
    g = load_golden("g06_emstep")
    d = float(g["delta"])
    close(S.em_increment(spec(), g["mu"], d, g["sig_diag"], g["dW"]), g["out_diag"], 1e-7)
    close(S.em_increment(spec(S.MSGM_DENSE), g["mu"], d, g["sig_dense"], g["dW"]), g["out_dense"])
    close(S.em_increment(spec(S.MSGM_SPARSE), g["mu"], d, g["sig_sparse"], g["dW"]), g["out_sparse"], 1e-7)


def _mlp_score(p, pre=None):
    p = {k[2:] if k.startswith("a.") else k: v for k, v in p.items()}
    return lambda y, s: N.mlp_forward(p, y, s, pre)


def test_g07_samplers():
    g = load_golden("g07_samplers")
    # SGM + MLP(2)
    sp = spec()
    proc = S.ReverseProcess(sp, _mlp_score(g.sub("sgm::")))
    for tag, fn in (("em", S.euler_maruyama), ("heun", S.heun), ("rk4", S.rk4_stratonovich)):
        z = g[f"sgm_{tag}_z"]
        tr = fn(proc, g["sgm_x0"], z.shape[0], z, keep_all=True, include_t0=True)
        close(tr, g[f"sgm_{tag}_traj"], 5e-6)
    z = g["sgm_em_l05_z"]
    tr = S.euler_maruyama(S.ReverseProcess(sp, _mlp_score(g.sub("sgm::")), lmbd=0.5), g["sgm_x0"], 8, z, keep_all=True)
    close(tr, g["sgm_em_l05_traj"], 5e-6)
    close(S.euler_maruyama(proc, g["sgm_x0"], 8, g["sgm_em_final_z"]), g["sgm_em_final"], 5e-6)
    kept = S.euler_maruyama(proc, g["sgm_x0"], 8, g["sgm_em_keep_z"], stop_index=g["sgm_em_keep_idx"])
    close(kept, g["sgm_em_keep"], 5e-6)
    # MSGM sparse + MLP(6, NormalizeLogRadius)
    sp = spec(S.MSGM_SPARSE, n=6)
    proc = S.ReverseProcess(sp, _mlp_score(g.sub("sp::"), "NormalizeLogRadius"))
    for nc in (0, 1):
        for tag, fn in (("em", S.euler_maruyama), ("heun", S.heun), ("rk4", S.rk4_stratonovich)):
            z = g[f"sp_{tag}_nc{nc}_z"]
            tr = fn(proc, g["sp_x0"], z.shape[0], z, keep_all=True, include_t0=True, norm_correction=bool(nc))
            close(tr, g[f"sp_{tag}_nc{nc}_traj"], 2e-5)
    z = g["sp_fwd_rk4_z"]
    tr = S.rk4_stratonovich(S.ForwardProcess(sp), g["sp_x0"], 4, z, keep_all=True, include_t0=True, norm_correction=True)
    close(tr, g["sp_fwd_rk4_traj"], 5e-6)
    # MSGM dense + MLP(4)
    sp = spec(S.MSGM_DENSE, n=4, G=g["dn_G"])
    proc = S.ReverseProcess(sp, _mlp_score(g.sub("dn::")))
    for tag, fn in (("em", S.euler_maruyama), ("rk4", S.rk4_stratonovich)):
        z = g[f"dn_{tag}_z"]
        tr = fn(proc, g["dn_x0"], z.shape[0], z, keep_all=True, include_t0=True, norm_correction=True)
        close(tr, g[f"dn_{tag}_traj"], 2e-5)


@pytest.mark.parametrize("tag,kind,n", [("sp", S.MSGM_SPARSE, 6), ("dn", S.MSGM_DENSE, 4)])
def test_g08_msgm_forward_perturb(tag, kind, n):
    g = load_golden("g08_sample_scheme")
    sp = spec(kind, n=n, num_steps_forward=4, G=g.get("dn_G") if kind == S.MSGM_DENSE else None)
    y, k = S.msgm_forward_perturb(sp, g[f"{tag}_t"], g[f"{tag}_x0"], g[f"{tag}_z_main"], g[f"{tag}_z_short"])
    assert torch.equal(k, g[f"{tag}_k"])
    assert int(k.min()) == 0 and int(k.max()) == 4
    close(y, g[f"{tag}_y"], 5e-6)


@pytest.mark.parametrize("tag,pre", [("mlp2", None), ("mlp2n", "NormalizeLogRadius"), ("mlp6n", "NormalizeLogRadius"), ("mlp16", None)])
def test_g09_mlp(tag, pre):
    g = load_golden("g09_mlp")
    close(N.mlp_forward(g.sub(tag + "::"), g[tag + "_x"], g[tag + "_t"], pre), g[tag + "_out"], 2e-6)


@pytest.mark.parametrize("tag,L,pre", [("u1d", 1024, None), ("u1dn", 1024, "NormalizeLogRadius"), ("u1d_odd", 1001, None), ("u1d_small", 64, None)])
def test_g09_unet1d(tag, L, pre):
    g = load_golden("g09_unet1d")
    p = det_state_dict(unet1d_shapes(L, pre))
    with torch.no_grad():
        y = N.unet1d_forward(p, g[tag + "_x"], g[tag + "_t"], pre)
    close(y, g[tag + "_out"], 1e-5)


def test_unet2d_param_count():
    cfg = N.UNet2DConfig(in_space=64)
    n = sum(math.prod(s) for s in unet2d_shapes(cfg).values())
    assert n == 4023233


@pytest.mark.parametrize("tag,S_,pre,order", [("u2d16C", 16, None, "C"), ("u2d16F", 16, None, "F"),
                                               ("u2d16Fn", 16, "NormalizeLogRadius", "F"), ("u2d32F", 32, None, "F")])
def test_g09_unet2d(tag, S_, pre, order):
    g = load_golden("g09_unet2d")
    cfg = N.UNet2DConfig(in_space=S_, use_log_norm=pre is not None)
    p = det_state_dict(unet2d_shapes(cfg, "core."))
    with torch.no_grad():
        y = N.vorticity_unet_forward(p, g[tag + "_x"], g[tag + "_t"], cfg, pre, order)
    close(y, g[tag + "_out"], 2e-5)


def test_g09_unet2d_core64_three_channels():
    g = load_golden("g09_unet2d")
    cfg = N.UNet2DConfig(in_channels=3, out_channels=3, in_space=64)
    p = det_state_dict(unet2d_shapes(cfg))
    with torch.no_grad():
        y = N.unet2d_core_forward(p, g["core64_x"], g["core64_t"], cfg)
    close(y, g["core64_out"], 2e-5)


def _ssm_inputs(g, tag, sp):
    t = S.clamp_time(sp, g[tag + "_u_t"])
    y = S.vp_perturb(sp, t, g[tag + "_x"], g[tag + "_eps"])
    v = S.rademacher_from_uniform(g[tag + "_u_v"])
    return t, y, v


@pytest.mark.parametrize("tag,pre", [("mlp2", None), ("mlp6n", "NormalizeLogRadius")])
@pytest.mark.parametrize("form", ["jvp", "double_backward"])
def test_g10_ssm_mlp(tag, pre, form):
    g = load_golden("g10_ssm_mlp")
    sp = spec()
    p = {k[2:]: v for k, v in g.sub(tag + "::").items() if k.startswith("a.")}
    t, y, v = _ssm_inputs(g, tag, sp)
    score = lambda prm, yy, tt: N.mlp_forward(prm, yy, tt, pre)
    loss, per, grads = L.ssm_mean_and_grads(sp, score, p, t, y, v, form=form)
    close(per, g[tag + "_per"], 5e-6)
    for k, gr in grads.items():
        close(gr, g[f"{tag}_grad::a.{k}"], 2e-5)


def test_g10_ssm_unet1d():
    g = load_golden("g10_ssm_unets")
    sp = spec()
    p = det_state_dict(unet1d_shapes(256, None))
    t, y, v = _ssm_inputs(g, "u1d", sp)
    score = lambda prm, yy, tt: N.unet1d_forward(prm, yy, tt, None)
    loss, per, grads = L.ssm_mean_and_grads(sp, score, p, t, y, v, form="jvp")
    close(per, g["u1d_per"], 2e-5)
    _check_digest(g, "u1d", grads, "a.", 1e-4)


def test_g10_ssm_unet2d():
    g = load_golden("g10_ssm_unets")
    sp = spec()
    cfg = N.UNet2DConfig(in_space=16)
    p = det_state_dict(unet2d_shapes(cfg, "core."))
    t, y, v = _ssm_inputs(g, "u2d", sp)
    score = lambda prm, yy, tt: N.vorticity_unet_forward(prm, yy, tt, cfg, None, "F")
    loss, per, grads = L.ssm_mean_and_grads(sp, score, p, t, y, v, form="jvp")
    close(per, g["u2d_per"], 5e-5)
    _check_digest(g, "u2d", grads, "a.", 2e-4)


@pytest.mark.parametrize("tag,S_", [("w32", 32), ("w64", 64)])
def test_g17_ssm_unet2d_wellconditioned(tag, S_):
    """The oracle against the reference on the WELL-CONDITIONED parameter set (oracle.det_params.init_like_*): forward,
    per-sample SSM loss and the gradient digest at 32x32 and 64x64 (attention at T = 1024 / 256 — the C4 shapes)."""
    from oracle.det_params import init_like_state_dict
    g = load_golden("g17_ssm_wellconditioned")
    sp = spec()
    cfg = N.UNet2DConfig(in_space=S_)
    p = init_like_state_dict(unet2d_shapes(cfg, "core."))
    score = lambda prm, yy, tt: N.vorticity_unet_forward(prm, yy, tt, cfg, None, "F")
    with torch.no_grad():
        close(score(p, g[tag + "_x"], g[tag + "_fwd_t"]), g[tag + "_fwd"], 5e-6)
    t, y, v = _ssm_inputs(g, tag, sp)
    loss, per, grads = L.ssm_mean_and_grads(sp, score, p, t, y, v, form="jvp")
    close(per, g[tag + "_per"], 1e-5)
    _check_digest(g, tag, grads, "a.", 1e-4)


def test_g11_three_train_steps_adam():
    g = load_golden("g11_train3")
    sp = spec()
    p = {k[2:]: v.clone() for k, v in g.sub("init::").items() if k.startswith("a.")}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    vv = {k: torch.zeros_like(v) for k, v in p.items()}
    score = lambda prm, yy, tt: N.mlp_forward(prm, yy, tt, None)
    for i in range(3):
        t = S.clamp_time(sp, g["u_t"][i])
        y = S.vp_perturb(sp, t, g["x"][i], g["eps"][i])
        v = S.rademacher_from_uniform(g["u_v"][i])
        loss, per, grads = L.ssm_mean_and_grads(sp, score, p, t, y, v)
        assert float(loss) == pytest.approx(float(g["loss"][i]), rel=2e-5)
        for k in p:
            p[k], m[k], vv[k] = L.adam_step(p[k], grads[k], m[k], vv[k], i + 1, lr=1e-3)
    for k in p:
        close(p[k], g[f"final::a.{k}"], 2e-5)


def test_g12_timestep_embedding():
    g = load_golden("g12_embedding")
    close(N.sinusoidal_embedding(g["t"], 32), g["emb32"], 1e-7)
    close(N.sinusoidal_embedding(g["t"], 7), g["emb7"], 1e-7)


def test_g13_misc():
    g = load_golden("g13_misc")
    assert torch.equal(S.rademacher_from_uniform(g["rad_u"]), g["rad_v"])
    close(S.unit_sphere_from_normal(g["sph_z"]), g["sph_s"], 1e-7)
    r_T = torch.log(torch.linalg.norm(g["lat_xinit"], dim=1) + 1e-6)
    close(r_T, g["lat_rT"], 1e-7)
    close(S.msgm_latent(g["lat_rT"], g["lat_u"], g["lat_z"], log_map=True), g["lat_x0"], 1e-6)


@pytest.mark.parametrize("tag,kind,d,pre", [("sp", S.MSGM_SPARSE, 6, "NormalizeLogRadius"), ("dn", S.MSGM_DENSE, 4, None)])
@pytest.mark.parametrize("form", ["jvp", "double_backward"])
def test_g14_ssm_msgm(tag, kind, d, pre, form):
    g = load_golden("g14_ssm_msgm")
    sp = spec(kind, n=d, num_steps_forward=4, G=g.get("dn_G") if kind == S.MSGM_DENSE else None)
    p = {k[2:]: v for k, v in g.sub(tag + "::").items() if k.startswith("a.")}
    v = S.rademacher_from_uniform(g[tag + "_u_v"])
    score = lambda prm, yy, tt: N.mlp_forward(prm, yy, tt, pre)
    loss, per, grads = L.ssm_mean_and_grads(sp, score, p, g[tag + "_t"], g[tag + "_y"], v, form=form)
    close(per, g[tag + "_per"], 1e-5)
    for k, gr in grads.items():
        close(gr, g[f"{tag}_grad::a.{k}"], 5e-5)


# ---- g15: reporting metrics next to the hot path -----------------------------------------------
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g15_rbf_mmd(tag):
    from oracle import metrics_ref as M
    g = load_golden("g15_metrics")
    x, y = g[f"mmd_{tag}_x"], g[f"mmd_{tag}_y"]
    close(M.rbf_kernel(x, y), g[f"mmd_{tag}_Kxy"], 1e-6)
    assert abs(float(M.mmd(x, y)) - float(g[f"mmd_{tag}"])) <= 1e-6


def test_g15_log_latent_pdf_and_elbo():
    from oracle import metrics_ref as M
    g = load_golden("g15_metrics")
    close(M.sgm_log_latent_pdf(g["lp_y"]), g["lp"], 1e-6)
    sp = spec()
    p = {k[2:]: v for k, v in g.sub("elbo::").items() if k.startswith("a.")}
    score = lambda prm, yy, tt: N.mlp_forward(prm, yy, tt, None)
    elbo = M.elbo_sgm(sp, score, p, g["elbo_x"], g["elbo_draw0_rand"], g["elbo_draw1_randn_like"], g["elbo_draw2_rand"],
                      g["elbo_draw5_randn_like"])
    close(elbo, g["elbo"], 1e-5)


# ----------------------------------------------------------------------------- round 2 pins (g16)
def _unet2d16_score():
    cfg = N.UNet2DConfig(in_space=16)
    p = det_state_dict(unet2d_shapes(cfg, "core."))
    return lambda y, s: N.vorticity_unet_forward(p, y, s, cfg, None, "F")


@pytest.mark.parametrize("tag,steps", [("em", 8), ("rk4", 4), ("heun", 4)])
def test_g16_unet2d_reverse_sde_trajectories(tag, steps):
    """The reference integrators driving VorticityUNet 16x16 F-order (sde_scheme.py:43-269 x NNUnet.py:195-245)."""
    g = load_golden("g16_round2")
    fn = {"em": S.euler_maruyama, "rk4": S.rk4_stratonovich, "heun": S.heun}[tag]
    proc = S.ReverseProcess(spec(), _unet2d16_score())
    with torch.no_grad():
        tr = fn(proc, g["u2d_x0"], steps, g[f"u2d_{tag}_z"], keep_all=True, include_t0=True)
    e = rel_l2(tr, g[f"u2d_{tag}_traj"])
    print(f"oracle vs reference, U-Net {tag} trajectory: rel-L2 {e:.2e}")
    assert e <= 2e-5


def test_g16_mlp_em_256_steps():
    """Error growth over a long reverse-SDE run: 256 EM steps, every 32nd state pinned."""
    g = load_golden("g16_round2")
    proc = S.ReverseProcess(spec(), _mlp_score(g.sub("mlp::")))
    tr = S.euler_maruyama(proc, g["mlp_x0"], 256, g["mlp_em256_z"], keep_all=True, include_t0=True)
    e = rel_l2(tr[::32], g["mlp_em256_every32"])
    print(f"oracle vs reference, 256-step MLP EM: rel-L2 {e:.2e}")
    assert e <= 2e-5


@pytest.mark.parametrize("vt", ["gaussian", "uniform"])
def test_g16_ssm_gaussian_and_sphere_probes(vt):
    """SSM loss with the non-Rademacher probes (SDEs.py:517-536)."""
    g = load_golden("g16_round2")
    sp = spec()
    p = {k[2:]: v for k, v in g.sub("mlp::").items() if k.startswith("a.")}
    t = S.clamp_time(sp, g[f"ssm_{vt}_u_t"])
    y = S.vp_perturb(sp, t, g[f"ssm_{vt}_x"], g[f"ssm_{vt}_eps"])
    z = g[f"ssm_{vt}_zv"]
    v = z if vt == "gaussian" else z / torch.linalg.norm(z, dim=1, keepdim=True)
    score = lambda prm, yy, tt: N.mlp_forward(prm, yy, tt, None)
    for form in ("jvp", "double_backward"):
        loss, per, grads = L.ssm_mean_and_grads(sp, score, p, t, y, v, form=form)
        close(per, g[f"ssm_{vt}_per"], 5e-6)
        for k, gr in grads.items():
            close(gr, g[f"ssm_{vt}_grad::a.{k}"], 2e-5)


def test_g16_sample_at_times_below_t_epsilon():
    """SGMsde.sample(t, y0) uses t as given — no clamp at t_epsilon (SDEs.py:134-146)."""
    g = load_golden("g16_round2")
    y = S.vp_perturb(spec(), g["smallt_t"], g["smallt_x0"], g["smallt_eps"])
    close(y, g["smallt_y"], 1e-6)
