import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_host_gpu import make_gen
from test_unet2d_gpu import _vunet
from oracle.shapes import unet2d_shapes
from oracle import nets_ref as N, sde_ref as S, ssm_ref as LR
from oracle.det_params import det_state_dict
S_ = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = _vunet(S_, "F")
gen = make_gen("sgm", net)
torch.manual_seed(0)
B, d = 2, S_ * S_
x, u, eps, uv = torch.randn(B, d) * 3, torch.rand(B), torch.randn(B, d), torch.rand(B, d)
cfg = N.UNet2DConfig(in_space=S_)
p = det_state_dict(unet2d_shapes(cfg, "core."))
score = lambda prm, yy, tt: N.vorticity_unet_forward(prm, yy, tt, cfg, None, "F")
gen.zero_grad()
per = gen.ssm(x.cuda(), u=u.cuda(), eps=eps.cuda(), u_v=uv.cuda())
per.mean().backward()
g = {k: pp.grad.detach().cpu().double() for k, pp in gen.a.named_parameters()}
sp = S.SdeSpec()
t = S.clamp_time(sp, u.reshape(B, 1)); y = S.vp_perturb(sp, t, x, eps); v = S.rademacher_from_uniform(uv)
dt = torch.float64
_, per64, g64 = LR.ssm_mean_and_grads(sp, score, {k: w.to(dt) for k, w in p.items()}, t.to(dt), y.to(dt), v.to(dt))
top = max(float(g64[k].norm()) for k in g64)
rows = sorted(((float((g[k] - g64[k]).norm()), k, float(g64[k].norm())) for k in g64), reverse=True)
tot = sum(r[0] ** 2 for r in rows) ** 0.5
print("flat abs err", tot, "flat norm", sum(float(g64[k].norm()) ** 2 for k in g64) ** 0.5)
for e, k, n in rows[:14]:
    print(f"{k:50s} abs err {e:.3e}  norm {n:.3e}  rel {e / max(n, 1e-30):.2e}")
# ---- variant: the embedding projections replaced by their correctly rounded values (float64 product, rounded once)
from sdeflow_light_amd import ops as _ops
_orig = _ops.EmbBank.forward
def exact_fwd(self, semb, rows, n_bias):
    outs = _orig(self, semb, rows, n_bias)
    s64 = semb.view(rows, self.K).double()
    for (w, b, _), o in zip(self.items, outs):
        ref = s64 @ w.double().t()
        ref[:n_bias] += b.double()
        o.copy_(ref.float().reshape(-1))
    return outs
_ops.EmbBank.forward = exact_fwd
gen.zero_grad()
per = gen.ssm(x.cuda(), u=u.cuda(), eps=eps.cuda(), u_v=uv.cuda())
per.mean().backward()
g2 = {k: pp.grad.detach().cpu().double() for k, pp in gen.a.named_parameters()}
rows2 = sorted(((float((g2[k] - g64[k]).norm()), k, float(g64[k].norm())) for k in g64), reverse=True)
print("EXACT eo: flat abs err", sum(r[0] ** 2 for r in rows2) ** 0.5)
for e, k, n in rows2[:5]:
    print(f"{k:50s} abs err {e:.3e}  norm {n:.3e}  rel {e / max(n, 1e-30):.2e}")
