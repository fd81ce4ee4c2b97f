import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_host_gpu import make_gen
from sdeflow_light_amd.NNUnet1D import UNet1D
from sdeflow_light_amd.train import UNetScoreTrainer
DEV = "cuda"
for use_graph in (False, True):
    torch.manual_seed(1)
    gen = make_gen("sgm", UNet1D(input_dim=128, base_channels=16, channel_mults=(1, 2), emb_dim=32))
    tr = UNetScoreTrainer(gen, 8, 128, lr=1e-3, seed=4, use_graph=use_graph)
    torch.manual_seed(0)
    tr.set_data(torch.randn(8, 128, device=DEV))
    for i in range(6):
        l = float(tr.step())
        flat, g = gen.a.flat_parameters()
        print(use_graph, i, l, bool(torch.isfinite(flat).all()), bool(torch.isfinite(tr.gbuf).all()), float(tr.gbuf[:-1].abs().max()), flush=True)
