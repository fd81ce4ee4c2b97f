import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_host_gpu import make_gen
from sdeflow_light_amd.NNUnet1D import UNet1D
from sdeflow_light_amd.train import UNetScoreTrainer
DEV = "cuda"
torch.manual_seed(0)
x = torch.randn(8, 128, device=DEV)
torch.manual_seed(1)
gen = make_gen("sgm", UNet1D(input_dim=128, base_channels=16, channel_mults=(1, 2), emb_dim=32))
opt = UNetScoreTrainer(gen, 8, 128, lr=0.0, seed=4, use_graph=True)
opt.set_data(x)
def step():
    opt.rng.state[1] = 7          # same noise every step; lr = 0: same parameters
    l = float(opt.step())
    return l, opt.gbuf.clone()
l0, g0 = step()                   # eager (capture)
l1, g1 = step()                   # replay back to back
l2, g2 = step()
torch.cuda.synchronize(); time.sleep(0.5)
l3, g3 = step()                   # replay after idle
l4, g4 = step()
names, off = [], 0
for k, p in gen.a.named_parameters():
    names.append((k, off, off + p.numel())); off += p.numel()
def rep(tag, a, b):
    d = (a - b).abs()
    bad = (d > 1e-4 * (b.abs() + 1e-3)).nonzero().reshape(-1)
    print(tag, "max abs diff", float(d.max()), "n bad", bad.numel(), "finite", bool(torch.isfinite(a).all()))
    seen = set()
    for i in bad.tolist()[:2000]:
        for k, s, e in names:
            if s <= i < e and k not in seen:
                seen.add(k); print("    ", k, i - s, float(a[i]), float(b[i]))
rep("replay1 vs eager", g1, g0)
rep("replay2 vs eager", g2, g0)
rep("after-idle vs eager", g3, g0)
rep("next vs eager", g4, g0)
print(l0, l1, l2, l3, l4)
