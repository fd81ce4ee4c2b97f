"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle and the golden
vectors.  Tolerances: elementwise kernels rel-L2 <= 1e-5, fused nets <= 1e-4
(north_star: fp32 tolerance, sampler within 1e-4 rel-L2), integer index
tensors bit-exact."""
import math

import pytest
import torch

from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu

from oracle import sde_ref as S
from oracle import nets_ref as N
from oracle import ssm_ref as LR

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from sdeflow_light_amd import ops as _ops
    _ops.lib()
    return _ops


@pytest.fixture(scope="module")
def L():
    from sdeflow_light_amd import _lib
    return _lib


def dev(t):
    return t.to(DEV).contiguous()


def sgm_struct(L, **kw):
    return L.sde_struct(L.SDE_SGM, 0.1, 20.0, 1.0, 1e-3)


# ------------------------------------------------------------------ K1
@pytest.mark.parametrize("d", [2, 1024])
def test_perturb_vp_golden(ops, L, d):
    g = load_golden("g04_vp_perturb")
    x0, t, eps = g[f"x0_{d}"], g[f"t_{d}"], g[f"eps_{d}"]
    y, t_out = ops.perturb_vp(dev(x0), sgm_struct(L), u=dev(t.reshape(-1)), eps=dev(eps))   # T=1 so u == t
    assert rel_l2(y.cpu(), g[f"y_{d}"]) <= 1e-5
    assert torch.equal(t_out.cpu(), t.reshape(-1))


def test_perturb_clamp_golden(ops, L):
    g = load_golden("g02_sample_t")
    u = g["u_edge"]
    x0 = torch.zeros(u.shape[0], 2)
    y, t = ops.perturb_vp(dev(x0), sgm_struct(L), u=dev(u.reshape(-1)), eps=dev(torch.zeros_like(x0)))
    assert torch.equal(t.cpu(), g["t_edge"].reshape(-1))      # mask arithmetic is exact


def test_perturb_vp_philox_consistent(ops, L):
    """In-kernel draws equal the numbers msgm_fill_* reports for the same
    stream, and feeding those to the oracle reproduces y."""
    B, d = 4099, 6
    rng = L.PhiloxState(1234, DEV)
    x0 = torch.randn(B, d)
    y, t, eps = ops.perturb_vp(dev(x0), sgm_struct(L), rng=rng, return_eps=True)
    u = ops.fill_uniform(torch.empty(B, device=DEV), rng, L.RNG_STREAM_T)
    e2 = ops.fill_normal(torch.empty(B * d, device=DEV), rng, L.RNG_STREAM_EPS).reshape(B, d)
    assert torch.equal(eps, e2)
    sp = S.SdeSpec()
    t_ref = S.clamp_time(sp, u.cpu().reshape(B, 1))
    assert torch.equal(t.cpu(), t_ref.reshape(-1))
    assert rel_l2(y.cpu(), S.vp_perturb(sp, t_ref, x0, eps.cpu())) <= 1e-5
    # moments of the generators
    z = ops.fill_normal(torch.empty(1 << 20, device=DEV), rng, 17)
    uu = ops.fill_uniform(torch.empty(1 << 20, device=DEV), rng, 18)
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.std()) - 1) < 5e-3
    assert abs(float((z ** 4).mean()) - 3) < 0.05
    assert abs(float(uu.mean()) - 0.5) < 2e-3 and float(uu.min()) >= 0 and float(uu.max()) < 1
    rng.advance(1)
    z2 = ops.fill_normal(torch.empty(1 << 20, device=DEV), rng, 17)
    assert not torch.equal(z, z2) and abs(float((z * z2).mean())) < 5e-3


@pytest.mark.parametrize("nsf", [4, 16, 128])
def test_step_index_bit_exact(ops, nsf):
    g = load_golden("g03_step_index")
    k = ops.forward_step_index(dev(g[f"t_{nsf}"]), nsf, 1.0)
    assert k.dtype == torch.int32 and torch.equal(k.cpu(), g[f"k_{nsf}"])


def test_rademacher_golden(ops):
    g = load_golden("g13_misc")
    v = ops.rademacher(g["rad_u"].shape, DEV, u=dev(g["rad_u"]))
    assert torch.equal(v.cpu(), g["rad_v"])


# ------------------------------------------------------------------ K2/K3/K4
def _spec_and_struct(L, kind, n, G=None):
    if kind == "sgm":
        return S.SdeSpec(), L.sde_struct(L.SDE_SGM, 0.1, 20.0, 1.0, 1e-3), ()
    if kind == "sparse":
        return S.SdeSpec(kind=S.MSGM_SPARSE, n=n), L.sde_struct(L.SDE_MSGM_SPARSE, 0.1, 20.0, 1.0, 1e-3), ()
    sp = S.SdeSpec(kind=S.MSGM_DENSE, n=n, G=G)
    Gd, LGd = dev(G), dev(sp.L_G)
    return sp, L.sde_struct(L.SDE_MSGM_DENSE, 0.1, 20.0, 1.0, 1e-3, Gd, LGd), (Gd, LGd)


@pytest.mark.parametrize("kind,n,B", [("sgm", 2, 1000), ("sgm", 1024, 7), ("sgm", 5, 33), ("sparse", 6, 50),
                                      ("sparse", 1024, 5), ("sparse", 3, 129), ("dense", 4, 40), ("dense", 16, 9)])
@pytest.mark.parametrize("proc", ["reverse", "forward"])
@pytest.mark.parametrize("strato", [False, True])
def test_sde_stage_vs_oracle(ops, L, kind, n, B, proc, strato):
    torch.manual_seed(n * 131 + B)
    G = S.make_dense_G(n, torch.randn(n, n, n)) if kind == "dense" else None
    sp, st, keep = _spec_and_struct(L, kind, n, G)
    x, a, z = torch.randn(B, n), torch.randn(B, n), torch.randn(B, n)
    t, delta, lmbd = 0.37, 1.0 / 16, (0.25 if proc == "reverse" else 0.0)
    score = lambda y, s: a
    pr = S.ReverseProcess(sp, score, lmbd) if proc == "reverse" else S.ForwardProcess(sp)
    tt = torch.full((B, 1), t)
    mu = pr.drift_strato(tt, x) if strato else pr.drift(tt, x)
    inc = S.em_increment(sp, mu, delta, pr.sigma(tt, x), delta ** 0.5 * z)
    norm0 = torch.norm(x, dim=1)
    for nc in (False, True):
        ref = x + inc
        if nc:
            ref = ref * (norm0 / torch.norm(ref, dim=1))[:, None]
        out = torch.empty(B, n, device=DEV)
        ops.sde_stage(out, dev(x), 1.0, dev(x), dev(a) if proc == "reverse" else None, st,
                      L.PROC_REVERSE if proc == "reverse" else L.PROC_FORWARD, strato, t, delta, lmbd, z=dev(z),
                      norm0=dev(norm0) if nc else None)
        assert rel_l2(out.cpu(), ref) <= 1e-5, (nc, rel_l2(out.cpu(), ref))
    # pure increment with a pre-scaled dW and c_out
    out = torch.empty(B, n, device=DEV)
    ops.sde_stage(out, None, 0.5, dev(x), dev(a) if proc == "reverse" else None, st,
                  L.PROC_REVERSE if proc == "reverse" else L.PROC_FORWARD, strato, t, delta, lmbd, dW=dev(delta ** 0.5 * z))
    assert rel_l2(out.cpu(), 0.5 * inc) <= 1e-5


def test_sde_stage_golden_emstep(ops, L):
    """EMstep golden vectors (g06): drift-free forward process isolates sigma.dW."""
    g = load_golden("g06_emstep")
    # sparse layout: sigma given explicitly upstream; here sigma is recomputed from x, so
    # check the identity through the oracle instead (covered above) and pin the diag case:
    sp = S.SdeSpec()
    B, n = g["mu"].shape
    x = g["mu"]
    st = L.sde_struct(L.SDE_SGM, 0.1, 20.0, 1.0, 1e-3)
    out = torch.empty(B, n, device=DEV)
    ops.sde_stage(out, None, 1.0, dev(x), None, st, L.PROC_FORWARD, True, 0.5, float(g["delta"]), dW=dev(g["dW"]))
    tt = torch.full((B, 1), 0.5)
    ref = S.em_increment(sp, S.drift_f_strato(sp, tt, x), float(g["delta"]), S.diffusion_g(sp, tt, x), g["dW"])
    assert rel_l2(out.cpu(), ref) <= 1e-6


def test_sde_stage_rejects_bad_args(ops, L):
    x = torch.zeros(4, 6, device=DEV)
    st = L.sde_struct(L.SDE_MSGM_SPARSE, 0.1, 20.0, 1.0, 1e-3)
    with pytest.raises(L.MsgmError):                       # stencil may not run in place
        ops.sde_stage(x, x, 1.0, x, x.clone(), st, L.PROC_REVERSE, False, 0.1, 0.1, z=x.clone())
    with pytest.raises(L.MsgmError):                       # reverse needs the score
        ops.sde_stage(x.clone(), x, 1.0, x, None, st, L.PROC_REVERSE, False, 0.1, 0.1, z=x.clone())
    with pytest.raises(L.MsgmError):                       # shape mismatch is caught on the host
        ops.sde_stage(x.clone(), x, 1.0, x, x.clone(), st, L.PROC_REVERSE, False, 0.1, 0.1, z=torch.zeros(4, 5, device=DEV))


def test_rk4_combine_rownorm_keep(ops):
    torch.manual_seed(3)
    B, n = 37, 10
    x, k1, k2, k3, k4 = (torch.randn(B, n) for _ in range(5))
    norm0 = torch.norm(x, dim=1)
    ref = x + (k1 + 2 * k2 + 2 * k3 + k4) / 6
    out = ops.rk4_combine(torch.empty(B, n, device=DEV), dev(x), dev(k1), dev(k2), dev(k3), dev(k4))
    assert rel_l2(out.cpu(), ref) <= 1e-6
    out = ops.rk4_combine(torch.empty(B, n, device=DEV), dev(x), dev(k1), dev(k2), dev(k3), dev(k4), norm0=dev(norm0))
    assert rel_l2(out.cpu(), ref * (norm0 / torch.norm(ref, dim=1))[:, None]) <= 1e-6
    assert rel_l2(ops.row_norm(dev(x)).cpu(), norm0) <= 1e-6
    stop = torch.randint(0, 3, (B,), dtype=torch.int32)
    kept = torch.zeros(B, n, device=DEV)
    ops.keep_rows(kept, dev(x), dev(stop), 1)
    ref = torch.zeros(B, n)
    ref[stop == 1] = x[stop == 1]
    assert torch.equal(kept.cpu(), ref)


# ------------------------------------------------------------------ K13
def test_adam_vs_torch_optimizer(ops):
    torch.manual_seed(5)
    n = 33794
    p0, grads = torch.randn(n) * 0.1, [torch.randn(n) * 0.01 for _ in range(5)]
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pt], lr=1e-3)
    p, m, v = dev(p0.clone()), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    step_dev = torch.zeros(1, dtype=torch.int64, device=DEV)
    for i, g in enumerate(grads):
        pt.grad = g.clone()
        opt.step()
        ops.counter_inc(step_dev)
        ops.adam_step(p, dev(g), m, v, step=0, lr=1e-3, step_dev=step_dev)
    assert rel_l2(p.cpu(), pt.detach()) <= 1e-6
    # oracle formula agrees too
    po, mo, vo = p0.clone(), torch.zeros(n), torch.zeros(n)
    for i, g in enumerate(grads):
        po, mo, vo = LR.adam_step(po, g, mo, vo, i + 1)
    assert rel_l2(p.cpu(), po) <= 1e-6


# ------------------------------------------------------------------ K5 fused MLP
def _mlp_P(ops, p, pre):
    keep = [dev(p[f"main.{i}.{w}"]) for i in (0, 2, 4, 6) for w in ("weight", "bias")]
    return ops.mlp_params(*keep, premodule=pre is not None), keep


@pytest.mark.parametrize("tag,pre", [("mlp2", None), ("mlp2n", "NormalizeLogRadius"), ("mlp6n", "NormalizeLogRadius"), ("mlp16", None)])
def test_mlp_forward_golden(ops, tag, pre):
    g = load_golden("g09_mlp")
    P, keep = _mlp_P(ops, g.sub(tag + "::"), pre)
    a = ops.mlp_forward(P, dev(g[tag + "_x"]), dev(g[tag + "_t"]))
    assert rel_l2(a.cpu(), g[tag + "_out"]) <= 1e-5, rel_l2(a.cpu(), g[tag + "_out"])


@pytest.mark.parametrize("B", [1, 31, 32, 33, 1000, 8192 + 5])
@pytest.mark.parametrize("d,pre", [(2, None), (3, "NormalizeLogRadius"), (15, None), (16, None), (30, "NormalizeLogRadius")])
def test_mlp_forward_ragged(ops, B, d, pre):
    from tests_util import mlp_shapes
    torch.manual_seed(B * 7 + d)
    from oracle.det_params import det_state_dict
    p = det_state_dict(mlp_shapes(d, pre))
    x, t = torch.randn(B, d) * 1.5, torch.rand(B)
    P, keep = _mlp_P(ops, p, pre)
    a = ops.mlp_forward(P, dev(x), dev(t))
    ref = N.mlp_forward(p, x, t, pre)
    assert rel_l2(a.cpu(), ref) <= 1e-5, rel_l2(a.cpu(), ref)


def _flat_grads(grads, pre=None):
    return torch.cat([grads[f"main.{i}.{w}"].reshape(-1) for i in (0, 2, 4, 6) for w in ("weight", "bias")])


@pytest.mark.parametrize("tag,pre", [("mlp2", None), ("mlp6n", "NormalizeLogRadius")])
def test_mlp_ssm_grad_golden(ops, L, tag, pre):
    g = load_golden("g10_ssm_mlp")
    sp = S.SdeSpec()
    p = {k[2:]: v for k, v in g.sub(tag + "::").items() if k.startswith("a.")}
    t = S.clamp_time(sp, g[tag + "_u_t"])
    y = S.vp_perturb(sp, t, g[tag + "_x"], g[tag + "_eps"])
    v = S.rademacher_from_uniform(g[tag + "_u_v"])
    B, d = y.shape
    P, keep = _mlp_P(ops, p, pre)
    n = ops.mlp_num_params(d, pre is not None)
    grads = torch.empty(n, device=DEV)
    per = torch.empty(B, device=DEV)
    lsum = torch.empty(1, device=DEV)
    ws = ops.mlp_ssm_workspace(d, pre is not None, DEV)
    ops.mlp_ssm_grad(P, dev(y), dev(t.reshape(-1)), dev(v), sgm_struct(L), 1.0 / B, grads, ws, per, lsum)
    assert rel_l2(per.cpu(), g[tag + "_per"]) <= 1e-5, rel_l2(per.cpu(), g[tag + "_per"])
    assert float(lsum) == pytest.approx(float(g[tag + "_loss"]), rel=1e-5)
    ref = _flat_grads({k[len(tag) + 9:]: v for k, v in g.items() if k.startswith(f"{tag}_grad::a.")})
    assert rel_l2(grads.cpu(), ref) <= 1e-4, rel_l2(grads.cpu(), ref)
    # per-tensor check so a wrong small tensor cannot hide behind a big one
    off, worst = 0, 0.0
    for i in (0, 2, 4, 6):
        for wn in ("weight", "bias"):
            r = g[f"{tag}_grad::a.main.{i}.{wn}"].reshape(-1)
            worst = max(worst, rel_l2(grads[off:off + r.numel()].cpu(), r))
            off += r.numel()
    from conftest import within
    within(worst, 5e-7, f"fused MLP SSM kernel ({tag}): worst per-tensor gradient rel-L2 vs the reference")


@pytest.mark.parametrize("B,d,pre", [(1, 2, None), (15, 2, None), (17, 5, "NormalizeLogRadius"), (4097, 2, None),
                                     (300, 16, None), (260, 30, "NormalizeLogRadius"), (5000, 14, None)])
def test_mlp_ssm_grad_vs_oracle(ops, L, B, d, pre):
    from tests_util import mlp_shapes
    from oracle.det_params import det_state_dict
    torch.manual_seed(B + d)
    sp = S.SdeSpec()
    p = det_state_dict(mlp_shapes(d, pre))
    t = S.clamp_time(sp, torch.rand(B, 1))
    y = S.vp_perturb(sp, t, torch.randn(B, d) * 1.5, torch.randn(B, d))
    v = S.rademacher_from_uniform(torch.rand(B, d))
    score = lambda prm, yy, tt: N.mlp_forward(prm, yy, tt, pre)
    loss, per_ref, gref = LR.ssm_mean_and_grads(sp, score, p, t, y, v)
    P, keep = _mlp_P(ops, p, pre)
    n = ops.mlp_num_params(d, pre is not None)
    grads, per, lsum = torch.empty(n, device=DEV), torch.empty(B, device=DEV), torch.empty(1, device=DEV)
    ws = ops.mlp_ssm_workspace(d, pre is not None, DEV)
    ops.mlp_ssm_grad(P, dev(y), dev(t.reshape(-1)), dev(v), sgm_struct(L), 1.0 / B, grads, ws, per, lsum)
    assert rel_l2(per.cpu(), per_ref) <= 2e-5, rel_l2(per.cpu(), per_ref)
    assert float(lsum) == pytest.approx(float(loss), rel=2e-5)
    assert rel_l2(grads.cpu(), _flat_grads(gref)) <= 1e-4, rel_l2(grads.cpu(), _flat_grads(gref))


def test_mlp_em_sampler_golden(ops, L):
    """8 fused EM steps (net + update in one kernel) against the reference's
    trajectory with the recorded noise; sampler tolerance 1e-4 rel-L2."""
    g = load_golden("g07_samplers")
    p = {k[2:]: v for k, v in g.sub("sgm::").items() if k.startswith("a.")}
    P, keep = _mlp_P(ops, p, None)
    x = dev(g["sgm_x0"].clone())
    z = g["sgm_em_z"]
    steps = z.shape[0]
    ts = torch.linspace(0, 1, steps + 1) * 1.0
    delta = 1.0 / steps
    st = sgm_struct(L)
    for i in range(steps):
        ops.mlp_em_step(P, x, st, float(ts[i]), delta, 0.0, z=dev(z[i]))
        assert rel_l2(x.cpu(), g["sgm_em_traj"][i + 1]) <= 1e-4, (i, rel_l2(x.cpu(), g["sgm_em_traj"][i + 1]))
    # unfused path (mlp_forward + sde_stage) gives the same trajectory
    x2 = dev(g["sgm_x0"].clone())
    for i in range(steps):
        s = torch.full((x2.shape[0],), 1.0 - float(ts[i]), device=DEV)
        a = ops.mlp_forward(P, x2, s)
        ops.sde_stage(x2, x2, 1.0, x2, a, st, L.PROC_REVERSE, False, float(ts[i]), delta, 0.0, z=dev(z[i]))
    assert rel_l2(x2.cpu(), g["sgm_em_traj"][steps]) <= 1e-4
    assert rel_l2(x2.cpu(), x.cpu()) <= 1e-5


# ---- BASELINE full size (C2: B = 65 536): size-independent properties ---------------------------------------------
def test_c2_full_size_shard_equivalence_and_row_independence(ops, L):
    """At the full C2 batch the oracle is too slow, so check what must hold at any size:
    (i) data-parallel equivalence — the mean gradient of the full batch equals the average of the two half-batch
        gradients and the per-sample losses are the concatenation (rows never interact);
    (ii) the sampler's rows are independent and its noise is addressed by (step, element): changing the second half
        of the state leaves the first half of a 5-step Euler-Maruyama run bit-identical."""
    from tests_util import mlp_shapes
    from oracle.det_params import det_state_dict
    torch.manual_seed(7)
    B, d = 65536, 2
    sp = S.SdeSpec()
    p = det_state_dict(mlp_shapes(d, None))
    P, keep = _mlp_P(ops, p, None)
    t = S.clamp_time(sp, torch.rand(B, 1)).reshape(-1)
    y = torch.randn(B, d) * 1.5
    v = S.rademacher_from_uniform(torch.rand(B, d))
    n = ops.mlp_num_params(d, False)
    ws = ops.mlp_ssm_workspace(d, False, DEV)
    st = sgm_struct(L)

    def run(sl):
        yy, tt, vv = dev(y[sl].contiguous()), dev(t[sl].contiguous()), dev(v[sl].contiguous())
        grads, per, lsum = torch.empty(n, device=DEV), torch.empty(yy.shape[0], device=DEV), torch.empty(1, device=DEV)
        ops.mlp_ssm_grad(P, yy, tt, vv, st, 1.0 / yy.shape[0], grads, ws, per, lsum)
        return grads.cpu(), per.cpu(), float(lsum)

    gf, pf, lf = run(slice(0, B))
    g0, p0, l0 = run(slice(0, B // 2))
    g1, p1, l1 = run(slice(B // 2, B))
    assert torch.isfinite(gf).all() and torch.isfinite(pf).all()
    assert torch.equal(pf, torch.cat([p0, p1]))                       # same per-row arithmetic wherever the row sits
    assert rel_l2(0.5 * (g0 + g1), gf) <= 2e-5
    assert 0.5 * (l0 + l1) == pytest.approx(lf, rel=1e-5)

    ts = (torch.linspace(0, 1, 6) * 1.0).to(DEV)
    rng = L.PhiloxState(99, torch.device(DEV))
    xa = dev(torch.randn(B, d))
    xb = xa.clone()
    xb[B // 2:] = 3.0 * torch.randn(B // 2, d, device=DEV)
    ops.mlp_em_loop(P, xa, st, ts, 0.2, 0.0, rng, 0)
    ops.mlp_em_loop(P, xb, st, ts, 0.2, 0.0, rng, 0)
    assert torch.isfinite(xa).all() and torch.isfinite(xb).all()
    assert torch.equal(xa[: B // 2], xb[: B // 2]) and not torch.equal(xa[B // 2:], xb[B // 2:])
