#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the ACTUAL reference on CPU.

Build-container only: imports /root/reference (read-only) with the
plotting / IO-only dependencies it cannot import here stubbed out
(seaborn, netCDF4, torchvision — SURVEY.md App. C; the stubs never touch
arithmetic).  Every random draw on the hot path (torch.rand / randn /
randn_like: SDEs.py:141,515,518,522,688; sde_scheme.py:84,144,227) is
recorded in call order and stored next to the inputs and the reference's
outputs, so the oracle and the HIP kernels can be fed identical noise.

The reference never travels to the GPU box: only the .npz vectors written
here (inputs + expected outputs, no source) are committed.

Usage:  python tools/make_golden.py            # rewrites tests/golden/
"""
import importlib.machinery as mach
import os
import sys
import types

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")


def _stub(name):
    m = types.ModuleType(name)
    m.__spec__ = mach.ModuleSpec(name, None)
    m.__path__ = []
    sys.modules[name] = m
    return m


_stub("seaborn")
_stub("netCDF4").Dataset = object
_tv = _stub("torchvision")
for _s in ("datasets", "transforms", "utils"):
    setattr(_tv, _s, _stub("torchvision." + _s))
sys.modules["torchvision.utils"].save_image = lambda *a, **k: None
sys.path.insert(0, "/root/reference")

from NN import MLP  # noqa: E402
from NNUnet import VorticityUNet, UNetModelWithLogNorm  # noqa: E402
from NNUnet1D import UNet1D  # noqa: E402
from SDEs import SGMsde, MSGMsde, PluginReverseSDE, forward_SDE, sample_rademacher, randu_on_sphere  # noqa: E402
from sde_scheme import EMstep, euler_maruyama_sampler, heun_sampler, rk4_stratonovich_sampler  # noqa: E402
from model.nn_utils import timestep_embedding  # noqa: E402

from oracle.det_params import load_det_, load_init_like_  # noqa: E402


class Recorder:
    """Patch torch.rand / randn / randn_like and log every draw in order."""

    def __init__(self):
        self.draws = []

    def __enter__(self):
        self._o = (torch.rand, torch.randn, torch.randn_like)
        rec = self.draws

        def wrap(fn, kind):
            def f(*a, **k):
                out = fn(*a, **k)
                rec.append((kind, out.detach().clone()))
                return out
            return f
        torch.rand, torch.randn, torch.randn_like = (wrap(self._o[0], "rand"), wrap(self._o[1], "randn"),
                                                     wrap(self._o[2], "randn_like"))
        return self

    def __exit__(self, *a):
        torch.rand, torch.randn, torch.randn_like = self._o


def npy(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrs.items()})
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def Tparam(T0=1.0):
    return torch.nn.Parameter(torch.FloatTensor([T0]), requires_grad=False)


def sgm(nsf=16):
    return SGMsde(beta_min=0.1, beta_max=20.0, t_epsilon=1e-3, T=Tparam(), num_steps_forward=nsf, device="cpu")


def msgm(x_init, dense, nsf=16, norm_map="log"):
    return MSGMsde(x_init, beta_min=0.1, beta_max=20.0, t_epsilon=1e-3, T=Tparam(), num_steps_forward=nsf,
                   device="cpu", estim_cst_norm_dens_r_T=False, norm_sampler="ecdf", norm_map=norm_map,
                   denseTensor=dense, plot_validate=False)


def sd_np(sd, skip=("T", "base_sde.T")):
    return {k: npy(v) for k, v in sd.items() if k not in skip}


# ---------------------------------------------------------------------------

def g01_schedule():
    s = sgm()
    t = torch.cat([torch.tensor([0.0, 1e-3, 1.0]), torch.linspace(0, 1, 254)]).reshape(-1, 1)
    save("g01_schedule", t=t, beta=s.beta(t), mean_weight=s.mean_weight(t), var=s.var(t))


def g02_sample_t():
    torch.manual_seed(2)
    s = sgm()
    rev = PluginReverseSDE(s, MLP(2), Tparam())
    x = torch.zeros(4096, 2)
    with Recorder() as r:
        t = rev.sample_t(x)
    u = r.draws[0][1].clone()
    # force a few raw draws onto / below the clamp edge
    u2 = u.clone()
    u2[:4, 0] = torch.tensor([0.0, 1e-3, 9.99e-4, 1.0001e-3])
    orig = torch.rand
    torch.rand = lambda *a, **k: u2.clone()
    t2 = rev.sample_t(x)
    torch.rand = orig
    save("g02_sample_t", u=u, t=t, u_edge=u2, t_edge=t2)


def g03_step_index():
    out = {}
    for nsf in (4, 16, 128):
        s = sgm(nsf)
        grid = torch.cat([torch.arange(0, nsf + 1, dtype=torch.float32) / nsf,
                          torch.tensor([5e-4, 0.06, 0.0625, 0.99, 1.0, 0.999999, 1e-3]),
                          torch.linspace(0, 1, 301)]).reshape(-1, 1)
        k = torch.trunc(s.num_steps_forward * grid / s.T).to(torch.int).to("cpu")      # SDEs.py:89-90
        for b in range(grid.shape[0]):
            if grid[b] >= s.T:
                k[b] = nsf
        out[f"t_{nsf}"] = grid
        out[f"k_{nsf}"] = k.reshape(-1).to(torch.int32)
    save("g03_step_index", **out)


def g04_vp_perturb():
    torch.manual_seed(4)
    s = sgm()
    out = {}
    for d, B in ((2, 512), (1024, 8)):
        x0 = torch.randn(B, d) * 2
        t = torch.rand(B, 1).clamp_min(1e-3)
        with Recorder() as r:
            y = s.sample(t, x0)
        out.update({f"x0_{d}": x0, f"t_{d}": t, f"eps_{d}": r.draws[0][1], f"y_{d}": y})
    save("g04_vp_perturb", **out)


def g05_drift_diffusion():
    torch.manual_seed(5)
    out = {}
    s = sgm()
    y = torch.randn(16, 3)
    t = torch.rand(16, 1)
    out.update(sgm_y=y, sgm_t=t, sgm_f=s.f(t, y), sgm_fs=s.f_strato(t, y), sgm_div=s.div_Sigma(t, y), sgm_g=s.g(t, y))
    xi = torch.randn(64, 4)
    with Recorder() as r:
        md = msgm(xi, dense=True)
    gen = torch.stack([d[1] for d in r.draws if d[0] == "randn" and tuple(d[1].shape) == (4, 4)][:4])
    y = torch.randn(16, 4)
    t = torch.rand(16, 1)
    out.update(dense_gen=gen, dense_G=md.G, dense_LG=md.L_G, dense_y=y, dense_t=t, dense_f=md.f(t, y),
               dense_fs=md.f_strato(t, y), dense_div=md.div_Sigma(t, y), dense_g=md.g(t, y))
    for n, B in ((6, 16), (1024, 2)):
        ms = msgm(torch.randn(32, n), dense=False)
        y = torch.randn(B, n)
        t = torch.rand(B, 1)
        I, J, K = ms.IJK()
        out.update({f"sp{n}_I": I, f"sp{n}_J": J, f"sp{n}_K": K, f"sp{n}_V": ms.G_V, f"sp{n}_y": y, f"sp{n}_t": t,
                    f"sp{n}_f": ms.f(t, y), f"sp{n}_div": ms.div_Sigma(t, y), f"sp{n}_g": ms.g(t, y, sparse=True)})
    save("g05_drift_diffusion", **out)


def g06_emstep():
    torch.manual_seed(6)
    B, n = 8, 6
    mu, dW = torch.randn(B, n), torch.randn(B, n)
    sd, sn, ss = torch.rand(B, n), torch.randn(B, n, n), torch.randn(B, 2 * n)
    ms = msgm(torch.randn(32, n), dense=False)
    I, J, K = ms.IJK()
    save("g06_emstep", mu=mu, dW=dW, delta=np.float64(0.0625), sig_diag=sd, sig_dense=sn, sig_sparse=ss,
         out_diag=EMstep(mu, 0.0625, sd, dW), out_dense=EMstep(mu, 0.0625, sn, dW),
         out_sparse=EMstep(mu, 0.0625, ss, dW, sparse=True, I=I, K=K))


def _run_sampler(fn, sde, x0, steps, **kw):
    with Recorder() as r:
        xs = fn(sde, x0, num_steps=steps, **kw)
    z = torch.stack([d[1] for d in r.draws if d[0] == "randn_like"])
    return xs, z


def g07_samplers():
    torch.manual_seed(7)
    out = {}
    # SGM + MLP d=2
    net = MLP(2)
    rev = PluginReverseSDE(sgm(), net, Tparam())
    out.update({"sgm::" + k: v for k, v in sd_np(rev.state_dict()).items()})
    x0 = torch.randn(32, 2)
    out["sgm_x0"] = x0
    for tag, fn, steps in (("em", euler_maruyama_sampler, 8), ("heun", heun_sampler, 4), ("rk4", rk4_stratonovich_sampler, 4)):
        xs, z = _run_sampler(fn, rev, x0, steps, keep_all_samples=True, include_t0=True)
        out[f"sgm_{tag}_traj"], out[f"sgm_{tag}_z"] = xs, z
    xs, z = _run_sampler(euler_maruyama_sampler, rev, x0, 8, keep_all_samples=True, include_t0=False, lmbd=0.5)
    out["sgm_em_l05_traj"], out["sgm_em_l05_z"] = xs, z
    xs, z = _run_sampler(euler_maruyama_sampler, rev, x0, 8, keep_all_samples=False)
    out["sgm_em_final"], out["sgm_em_final_z"] = xs, z
    # MSGM sparse + MLP d=6 (NormalizeLogRadius), norm_correction on/off
    net6 = MLP(6, premodule="NormalizeLogRadius")
    ms = msgm(torch.randn(64, 6) * 1.5, dense=False)
    rev6 = PluginReverseSDE(ms, net6, Tparam())
    out.update({"sp::" + k: v for k, v in sd_np(rev6.state_dict()).items()})
    x0 = torch.randn(16, 6)
    out["sp_x0"] = x0
    for nc in (False, True):
        for tag, fn, steps in (("em", euler_maruyama_sampler, 8), ("heun", heun_sampler, 4), ("rk4", rk4_stratonovich_sampler, 4)):
            xs, z = _run_sampler(fn, rev6, x0, steps, keep_all_samples=True, include_t0=True, norm_correction=nc)
            out[f"sp_{tag}_nc{int(nc)}_traj"], out[f"sp_{tag}_nc{int(nc)}_z"] = xs, z
    # MSGM dense + MLP d=4
    net4 = MLP(4)
    xi = torch.randn(64, 4)
    md = msgm(xi, dense=True)
    rev4 = PluginReverseSDE(md, net4, Tparam())
    out.update({"dn::" + k: v for k, v in sd_np(rev4.state_dict()).items()})
    out["dn_G"] = md.G
    x0 = torch.randn(16, 4)
    out["dn_x0"] = x0
    for tag, fn, steps in (("em", euler_maruyama_sampler, 8), ("rk4", rk4_stratonovich_sampler, 4)):
        xs, z = _run_sampler(fn, rev4, x0, steps, keep_all_samples=True, include_t0=True, norm_correction=True)
        out[f"dn_{tag}_traj"], out[f"dn_{tag}_z"] = xs, z
    # forward process (noising) with RK4, as the driver does (MSGM_higherDim.py:783)
    xs, z = _run_sampler(rk4_stratonovich_sampler, forward_SDE(ms, Tparam()), out["sp_x0"], 4,
                         keep_all_samples=True, include_t0=True, norm_correction=True, lmbd=0.)
    out["sp_fwd_rk4_traj"], out["sp_fwd_rk4_z"] = xs, z
    # samplesToKeep error behaviour is tested on the host side; store a valid masked run
    keep = torch.tensor([1, 2, 3, 4] * 8, dtype=torch.int32)
    xs, z = _run_sampler(euler_maruyama_sampler, rev, out["sgm_x0"], 8, keep_all_samples=False, samplesToKeep=keep)
    out["sgm_em_keep_idx"], out["sgm_em_keep"], out["sgm_em_keep_z"] = keep, xs, z
    save("g07_samplers", **out)


def g08_sample_scheme():
    torch.manual_seed(8)
    out = {}
    for tag, dense, n in (("sp", False, 6), ("dn", True, 4)):
        ms = msgm(torch.randn(64, n), dense=dense, nsf=4)
        B = 12
        x0 = torch.randn(B, n)
        t = torch.tensor([0.001, 0.1, 0.24, 0.25, 0.3, 0.5, 0.74, 0.75, 0.99, 1.0, 0.6, 0.2]).reshape(B, 1)
        with Recorder() as r:
            y = ms.sample(t, x0)
        z_main = torch.stack([d[1] for d in r.draws[:4]])
        z_short = torch.zeros(B, n)
        k = torch.trunc(4 * t / ms.T).to(torch.int).reshape(-1)
        k[t.reshape(-1) >= 1.0] = 4
        rest = [d[1] for d in r.draws[4:]]
        rows0 = [b for b in range(B) if int(k[b]) == 0]
        assert len(rest) == len(rows0), (len(rest), rows0)
        for b, d in zip(rows0, rest):
            z_short[b] = d[0]
        out.update({f"{tag}_x0": x0, f"{tag}_t": t, f"{tag}_y": y, f"{tag}_k": k.to(torch.int32),
                    f"{tag}_z_main": z_main, f"{tag}_z_short": z_short})
        if dense:
            out["dn_G"] = ms.G
    save("g08_sample_scheme", **out)


def g09_nets():
    torch.manual_seed(9)
    out = {}
    # MLP — stored state (small)
    for tag, d, pre in (("mlp2", 2, None), ("mlp2n", 2, "NormalizeLogRadius"), ("mlp6n", 6, "NormalizeLogRadius"), ("mlp16", 16, None)):
        net = MLP(d, premodule=pre)
        x, t = torch.randn(64, d) * 1.5, torch.rand(64)
        out.update({f"{tag}::" + k: npy(v) for k, v in net.state_dict().items()})
        out.update({f"{tag}_x": x, f"{tag}_t": t, f"{tag}_out": net(x, t)})
    save("g09_mlp", **out)

    out = {}
    # UNet1D — deterministic fill, only I/O stored
    for tag, L, pre in (("u1d", 1024, None), ("u1dn", 1024, "NormalizeLogRadius"), ("u1d_odd", 1001, None), ("u1d_small", 64, None)):
        net = UNet1D(input_dim=L, base_channels=32, channel_mults=(1, 2, 4), num_res_blocks=2, premodule=pre, emb_dim=128)
        load_det_(net)
        x, t = torch.randn(2, L), torch.rand(2)
        with torch.no_grad():
            y = net(x, t)
        out.update({f"{tag}_x": x, f"{tag}_t": t, f"{tag}_out": y})
    save("g09_unet1d", **out)

    out = {}
    for tag, S, pre, order in (("u2d16C", 16, None, "C"), ("u2d16F", 16, None, "F"), ("u2d16Fn", 16, "NormalizeLogRadius", "F"),
                               ("u2d32F", 32, None, "F")):
        net = VorticityUNet(base_channels=32, channel_mults=(1, 2, 4), num_res_blocks=2, premodule=pre, in_space=S,
                            attention_resolutions=(2, 4), flatten_order=order)
        load_det_(net)
        x, t = torch.randn(2, S * S) * 3, torch.rand(2)
        with torch.no_grad():
            y = net(x, t)
        out.update({f"{tag}_x": x, f"{tag}_t": t, f"{tag}_out": y})
    # 3-channel 64x64 core (config C4 shape) — the wrapper upstream is 1-channel only (NNUnet.py:177-179)
    core = UNetModelWithLogNorm(in_channels=3, model_channels=32, out_channels=3, in_space=64, num_res_blocks=2,
                                attention_resolutions=(2, 4), dropout=0.0, channel_mult=(1, 2, 4), conv_resample=True,
                                dims=2, num_classes=None, use_checkpoint=False, num_heads=1, use_scale_shift_norm=False,
                                learn_potential=False, use_log_norm=False)
    load_det_(core)
    x, t = torch.randn(1, 3, 64, 64), torch.rand(1)
    with torch.no_grad():
        y = core(x, timesteps=t)
    out.update(core64_x=x, core64_t=t, core64_out=y)
    save("g09_unet2d", **out)


def _grad_digest(named_grads):
    """Per-tensor L2 norm, sum, and the first 8 entries — enough to pin a
    4M-parameter gradient without committing 16 MB."""
    names = sorted(named_grads)
    norms = np.array([float(named_grads[k].double().norm()) for k in names])
    sums = np.array([float(named_grads[k].double().sum()) for k in names])
    heads = np.stack([np.pad(npy(named_grads[k].reshape(-1)[:8]), (0, max(0, 8 - named_grads[k].numel()))) for k in names])
    return dict(names=np.array(names), norms=norms, sums=sums, heads=heads)


def _ssm_case(rev, x, u_t, eps, u_v, full_grads):
    """Run the reference ssm() with the three draws forced."""
    seq = [u_t, eps, u_v] if eps is not None else None
    o = (torch.rand, torch.randn_like)
    it = iter(seq)
    torch.rand = lambda *a, **k: next(it).clone()
    torch.randn_like = lambda *a, **k: next(it).clone()
    try:
        rev.zero_grad()
        per = rev.ssm(x)
        per.mean().backward()
    finally:
        torch.rand, torch.randn_like = o
    grads = {k: p.grad.detach().clone() for k, p in rev.named_parameters() if p.grad is not None}
    res = dict(per=per.detach(), loss=per.mean().detach())
    if full_grads:
        res.update({"grad::" + k: v for k, v in grads.items()})
    else:
        res.update({"gd_" + k: v for k, v in _grad_digest(grads).items()})
    return res


def g10_ssm():
    torch.manual_seed(10)
    out = {}
    B = 64
    for tag, d, pre in (("mlp2", 2, None), ("mlp6n", 6, "NormalizeLogRadius")):
        net = MLP(d, premodule=pre)
        rev = PluginReverseSDE(sgm(), net, Tparam())
        x, u_t, eps, u_v = torch.randn(B, d) * 1.5, torch.rand(B, 1), torch.randn(B, d), torch.rand(B, d)
        res = _ssm_case(rev, x, u_t, eps, u_v, full_grads=True)
        out.update({f"{tag}::" + k: v for k, v in sd_np(rev.state_dict()).items()})
        out.update({f"{tag}_x": x, f"{tag}_u_t": u_t, f"{tag}_eps": eps, f"{tag}_u_v": u_v})
        out.update({f"{tag}_{k}": v for k, v in res.items()})
    save("g10_ssm_mlp", **out)

    out = {}
    net = UNet1D(input_dim=256, base_channels=32, channel_mults=(1, 2, 4), num_res_blocks=2, premodule=None, emb_dim=128)
    load_det_(net)
    rev = PluginReverseSDE(sgm(), net, Tparam())
    B, d = 4, 256
    x, u_t, eps, u_v = torch.randn(B, d), torch.rand(B, 1), torch.randn(B, d), torch.rand(B, d)
    res = _ssm_case(rev, x, u_t, eps, u_v, full_grads=False)
    out.update(u1d_x=x, u1d_u_t=u_t, u1d_eps=eps, u1d_u_v=u_v)
    out.update({f"u1d_{k}": v for k, v in res.items()})
    net = VorticityUNet(base_channels=32, channel_mults=(1, 2, 4), num_res_blocks=2, premodule=None, in_space=16,
                        attention_resolutions=(2, 4), flatten_order="F")
    load_det_(net)
    rev = PluginReverseSDE(sgm(), net, Tparam())
    B, d = 2, 256
    x, u_t, eps, u_v = torch.randn(B, d) * 3, torch.rand(B, 1), torch.randn(B, d), torch.rand(B, d)
    res = _ssm_case(rev, x, u_t, eps, u_v, full_grads=False)
    out.update(u2d_x=x, u2d_u_t=u_t, u2d_eps=eps, u2d_u_v=u_v)
    out.update({f"u2d_{k}": v for k, v in res.items()})
    save("g10_ssm_unets", **out)


def g14_ssm_msgm():
    """SSM loss of the multiplicative SDE (sparse + dense tensor) with an MLP: ssm_loss(t, x, y) with the
    probe draw forced (SDEs.py:616-646)."""
    torch.manual_seed(14)
    out = {}
    B = 48
    for tag, d, dense, pre in (("sp", 6, False, "NormalizeLogRadius"), ("dn", 4, True, None)):
        net = MLP(d, premodule=pre)
        base = msgm(torch.randn(64, d) * 1.5, dense=dense, nsf=4)
        rev = PluginReverseSDE(base, net, Tparam())
        t_ = torch.rand(B, 1).clamp_min(1e-3)
        y = torch.randn(B, d) * 1.3
        u_v = torch.rand(B, d)
        o = torch.rand
        torch.rand = lambda *a, **k: u_v.clone()
        try:
            rev.zero_grad()
            yy = y.clone().requires_grad_(True)
            per = rev.ssm_loss(t_, y, yy)
            per.mean().backward()
        finally:
            torch.rand = o
        out.update({f"{tag}::" + k: v for k, v in sd_np(rev.state_dict()).items()})
        out.update({f"{tag}_t": t_, f"{tag}_y": y, f"{tag}_u_v": u_v, f"{tag}_per": per.detach()})
        out.update({f"{tag}_grad::" + k: p.grad.detach().clone() for k, p in rev.named_parameters() if p.grad is not None})
        if dense:
            out["dn_G"] = base.G
    save("g14_ssm_msgm", **out)


def g15_metrics():
    """Reporting metrics next to the hot path: RBF-kernel MMD (quantitative_comparison.py:22-46),
    Gaussian latent log-density (SDEs.py:209-215) and the ELBO slice estimate (SDEs.py:708-721) with every draw
    recorded in call order."""
    from quantitative_comparison import compute_kernel, compute_mmd
    torch.manual_seed(15)
    out = {}
    for tag, nx, ny, d in (("a", 37, 53, 2), ("b", 64, 40, 7), ("c", 20, 20, 300)):
        x, y = torch.randn(nx, d) * 1.3, torch.randn(ny, d) * 0.9 + 0.4
        out.update({f"mmd_{tag}_x": x, f"mmd_{tag}_y": y, f"mmd_{tag}_Kxy": compute_kernel(x, y),
                    f"mmd_{tag}": compute_mmd(x, y).reshape(1)})
    yT = torch.randn(19, 5) * 1.7
    out.update(lp_y=yT, lp=sgm().log_latent_pdf(yT))
    B, d = 96, 2
    net = MLP(d)
    rev = PluginReverseSDE(sgm(), net, Tparam())
    x = torch.randn(B, d) * 1.5
    with Recorder() as r:
        elbo = rev.elbo_random_t_slice(x)
    out.update({"elbo::" + k: v for k, v in sd_np(rev.state_dict()).items()})
    out.update(elbo_x=x, elbo=elbo.detach())
    for i, (kind, val) in enumerate(r.draws):
        out[f"elbo_draw{i}_{kind}"] = val
    save("g15_metrics", **out)


def g11_train3():
    torch.manual_seed(11)
    B, d, steps = 128, 2, 3
    net = MLP(d)
    rev = PluginReverseSDE(sgm(), net, Tparam())
    opt = torch.optim.Adam(rev.parameters(), lr=1e-3)
    out = {"init::" + k: npy(v).copy() for k, v in rev.state_dict().items() if k not in ("T", "base_sde.T")}
    xs, uts, epss, uvs, losses = [], [], [], [], []
    for i in range(steps):
        x, u_t, eps, u_v = torch.randn(B, d) * 1.5, torch.rand(B, 1), torch.randn(B, d), torch.rand(B, d)
        seq = iter([u_t, eps, u_v])
        o = (torch.rand, torch.randn_like)
        torch.rand = lambda *a, **k: next(seq).clone()
        torch.randn_like = lambda *a, **k: next(seq).clone()
        try:
            opt.zero_grad()
            loss = rev.ssm(x).mean()
            loss.backward()
            opt.step()
        finally:
            torch.rand, torch.randn_like = o
        xs.append(x); uts.append(u_t); epss.append(eps); uvs.append(u_v); losses.append(loss.detach())
    out.update(x=torch.stack(xs), u_t=torch.stack(uts), eps=torch.stack(epss), u_v=torch.stack(uvs), loss=torch.stack(losses))
    out.update({"final::" + k: npy(v) for k, v in rev.state_dict().items() if k not in ("T", "base_sde.T")})
    save("g11_train3", **out)


def g12_embedding():
    t = torch.cat([torch.linspace(0, 1, 33), torch.tensor([-13.8, -2.5, -0.3, 0.7, 3.1])])
    save("g12_embedding", t=t, emb32=timestep_embedding(t, 32), emb7=timestep_embedding(t, 7))


def g13_misc():
    torch.manual_seed(13)
    with Recorder() as r:
        v = sample_rademacher((32, 5), "cpu")
    u = r.draws[0][1]
    with Recorder() as r:
        s = randu_on_sphere((32, 5), "cpu")
    z = r.draws[0][1]
    # MSGM latent sample (ecdf + log map)                                   SDEs.py:438-471
    xi = torch.randn(500, 6) * 2
    ms = msgm(xi, dense=False)
    with Recorder() as r:
        x0 = ms.latent_sample(40, 6)
    save("g13_misc", rad_u=u, rad_v=v, sph_z=z, sph_s=s, lat_xinit=xi, lat_rT=ms.r_T, lat_u=r.draws[0][1],
         lat_z=r.draws[1][1], lat_x0=x0)


def g16_round2():
    """Round-2 pins: (a) the reference integrators driving the 2-D U-Net (sde_scheme.py:43-99,174-269 x
    NNUnet.py:195-245) with recorded noise; (b) a 256-step MLP Euler-Maruyama run (error growth over a long
    reverse-SDE trajectory); (c) the SSM loss with the Gaussian and the sphere probe (SDEs.py:517-536)."""
    torch.manual_seed(16)
    out = {}
    net = VorticityUNet(base_channels=32, channel_mults=(1, 2, 4), num_res_blocks=2, premodule=None, in_space=16,
                        attention_resolutions=(2, 4), flatten_order="F")
    load_det_(net)
    rev = PluginReverseSDE(sgm(), net, Tparam())
    x0 = torch.randn(4, 256)
    out["u2d_x0"] = x0
    with torch.no_grad():
        xs, z = _run_sampler(euler_maruyama_sampler, rev, x0, 8, keep_all_samples=True, include_t0=True)
        out["u2d_em_traj"], out["u2d_em_z"] = xs, z
        xs, z = _run_sampler(rk4_stratonovich_sampler, rev, x0, 4, keep_all_samples=True, include_t0=True)
        out["u2d_rk4_traj"], out["u2d_rk4_z"] = xs, z
        xs, z = _run_sampler(heun_sampler, rev, x0, 4, keep_all_samples=True, include_t0=True)
        out["u2d_heun_traj"], out["u2d_heun_z"] = xs, z
    # (b) long MLP trajectory: every 32nd state of 256 EM steps
    netm = MLP(2)
    revm = PluginReverseSDE(sgm(), netm, Tparam())
    out.update({"mlp::" + k: v for k, v in sd_np(revm.state_dict()).items()})
    x0 = torch.randn(16, 2)
    with torch.no_grad():
        xs, z = _run_sampler(euler_maruyama_sampler, revm, x0, 256, keep_all_samples=True, include_t0=True)
    out["mlp_x0"], out["mlp_em256_z"], out["mlp_em256_every32"] = x0, z, xs[::32]
    # (c) SSM with the two non-Rademacher probes
    B, d = 48, 2
    for vt in ("gaussian", "uniform"):
        revv = PluginReverseSDE(sgm(), netm, Tparam(), vtype=vt)
        x = torch.randn(B, d) * 1.5
        with Recorder() as r:
            revv.zero_grad()
            per = revv.ssm(x)
            per.mean().backward()
        kinds = [k for k, _ in r.draws]
        assert kinds == ["rand", "randn_like", "randn"], kinds
        u_t, eps, zv = (t for _, t in r.draws)
        out.update({f"ssm_{vt}_x": x, f"ssm_{vt}_u_t": u_t, f"ssm_{vt}_eps": eps, f"ssm_{vt}_zv": zv, f"ssm_{vt}_per": per.detach()})
        out.update({f"ssm_{vt}_grad::" + k: p.grad.detach().clone() for k, p in revv.named_parameters() if p.grad is not None})
    # (d) SGMsde.sample(t, y0) with times BELOW t_epsilon: upstream uses t as given, no clamp (SDEs.py:134-146,196-199)
    s0 = sgm()
    x0 = torch.randn(8, 5)
    t = torch.tensor([1e-5, 1e-4, 5e-4, 9.99e-4, 1e-3, 2e-3, 0.5, 1.0]).reshape(8, 1)
    with Recorder() as r:
        y = s0.sample(t, x0)
    out.update(smallt_x0=x0, smallt_t=t, smallt_eps=r.draws[0][1], smallt_y=y)
    save("g16_round2", **out)


def g17_ssm_wellconditioned():
    """VERDICT r2 #5a: SSM per-sample loss + gradient digests of the reference's VorticityUNet at 32x32 and at 64x64
    (attention at T = 1024 / 256, i.e. the C4 network's attention shapes) on a WELL-CONDITIONED parameter set
    (oracle.det_params.load_init_like_: the statistics of the reference's default init, zero-init layers re-randomised
    small) — the fixture the HIP path is held to with an ABSOLUTE tolerance."""
    torch.manual_seed(17)
    out = {}
    for tag, S_, B in (("w32", 32, 2), ("w64", 64, 2)):
        net = VorticityUNet(base_channels=32, channel_mults=(1, 2, 4), num_res_blocks=2, premodule=None, in_space=S_,
                            attention_resolutions=(2, 4), flatten_order="F")
        load_init_like_(net)
        rev = PluginReverseSDE(sgm(), net, Tparam())
        d = S_ * S_
        x, u_t, eps, u_v = torch.randn(B, d) * 3, torch.rand(B, 1), torch.randn(B, d), torch.rand(B, d)
        res = _ssm_case(rev, x, u_t, eps, u_v, full_grads=False)
        out.update({f"{tag}_x": x, f"{tag}_u_t": u_t, f"{tag}_eps": eps, f"{tag}_u_v": u_v})
        out.update({f"{tag}_{k}": v for k, v in res.items()})
        with torch.no_grad():
            t = torch.full((B,), 0.37)
            out[f"{tag}_fwd_t"], out[f"{tag}_fwd"] = t, net(x, t)
        print(tag, "loss", float(res["loss"]), "per", res["per"].tolist())
    save("g17_ssm_wellconditioned", **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    only = sys.argv[1:]
    for name, fn in list(globals().items()):
        if name.startswith("g") and name[1:3].isdigit() and callable(fn):
            if only and not any(name.startswith(o) for o in only):
                continue
            fn()
