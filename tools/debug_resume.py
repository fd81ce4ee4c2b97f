import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_host_gpu import make_gen
from sdeflow_light_amd.NNUnet1D import UNet1D
from sdeflow_light_amd.NN import save_checkpoint, load_checkpoint
from sdeflow_light_amd.train import UNetScoreTrainer
DEV = "cuda"
torch.manual_seed(0)
x = torch.randn(8, 128, device=DEV)
def make(seed):
    torch.manual_seed(seed)
    gen = make_gen("sgm", UNet1D(input_dim=128, base_channels=16, channel_mults=(1, 2), emb_dim=32))
    opt = UNetScoreTrainer(gen, 8, 128, lr=1e-3, seed=4)
    opt.set_data(x)
    return gen, opt
def fin(tag, gen, opt):
    flat, _ = gen.a.flat_parameters()
    print(tag, "params", bool(torch.isfinite(flat).all()), "m", bool(torch.isfinite(opt.m).all()), "v", bool(torch.isfinite(opt.v).all()), float(opt.v.min()),
          "step", int(opt.step_dev), "rng", opt.rng.state.tolist(), flush=True)
gen, opt = make(1)
for i in range(2): print("loss", float(opt.step()))
fin("after2", gen, opt)
save_checkpoint("/tmp/ck.pt", gen, opt, 2)
for i in range(2): print("loss", float(opt.step()))
fin("after4", gen, opt)
gen2, opt2 = make(2)
fin("fresh", gen2, opt2)
load_checkpoint("/tmp/ck.pt", gen2, opt2, DEV)
fin("loaded", gen2, opt2)
f1, f2 = gen.a.flat_parameters()[0], gen2.a.flat_parameters()[0]
for i in range(2):
    print("loss", float(opt2.step()))
    fin(f"resumed{i}", gen2, opt2)
