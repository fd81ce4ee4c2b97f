import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_host_gpu import make_gen
from sdeflow_light_amd.NNUnet1D import UNet1D
from sdeflow_light_amd.train import UNetScoreTrainer
DEV = "cuda"
torch.manual_seed(0)
x = torch.randn(8, 128, device=DEV)
def make(seed, use_graph=True):
    torch.manual_seed(seed)
    gen = make_gen("sgm", UNet1D(input_dim=128, base_channels=16, channel_mults=(1, 2), emb_dim=32))
    opt = UNetScoreTrainer(gen, 8, 128, lr=1e-3, seed=4, use_graph=use_graph)
    opt.set_data(x)
    return gen, opt
for variant in sys.argv[1:]:
    gen, opt = make(1, use_graph=(variant[0] != "E"))
    v = variant.lstrip("E")
    out = []
    for i in range(5):
        out.append(round(float(opt.step()), 4))
        if i == 1:
            if v == "optsd":
                sd = opt.state_dict()
            elif v == "gensd":
                sd = gen.state_dict()
            elif v == "alloc":
                junk = [torch.randn(3000, device=DEV) for _ in range(50)]
            elif v == "save":
                torch.save({"m": gen.state_dict()}, "/tmp/x.pt")
            elif v == "item":
                _ = float(opt.step_dev.item())
            elif v == "clone":
                c = opt.m.clone()
            elif v == "sleep":
                import time; torch.cuda.synchronize(); time.sleep(0.5)
            elif v == "d2hflat":
                c = opt.flat.cpu()
            elif v == "d2hx":
                c = x.cpu()
            elif v == "d2hparam":
                c = next(gen.a.parameters()).detach().cpu()
            elif v == "storagecpu":
                c = next(gen.a.parameters()).detach().untyped_storage().cpu()
            elif v == "savecpu":
                torch.save({k: t.cpu() for k, t in gen.state_dict().items()}, "/tmp/x.pt")
            elif v == "saveT":
                torch.save({"T": gen.state_dict()["T"]}, "/tmp/x.pt")
            elif v == "save1":
                torch.save({"w": next(gen.a.parameters()).detach()}, "/tmp/x.pt")
    print(variant, out, bool(torch.isfinite(opt.v).all()), flush=True)
