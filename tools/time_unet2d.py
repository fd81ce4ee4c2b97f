#!/usr/bin/env python3
"""GPU box: time the 2-D U-Net (C4: 64x64x3) train step and one EM sampler step."""
import sys, time, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdeflow_light_amd.NNUnet import VorticityUNet
from sdeflow_light_amd.SDEs import SGMsde, PluginReverseSDE
from sdeflow_light_amd.train import UNetScoreTrainer
from sdeflow_light_amd.data import random_images
from sdeflow_light_amd import ops, _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
S_, Cc = 64, 3
d = Cc * S_ * S_
dev = torch.device("cuda")
torch.manual_seed(0)
net = VorticityUNet(base_channels=32, channel_mults=(1, 2, 4), num_res_blocks=2, in_space=S_, attention_resolutions=(2, 4),
                    flatten_order="F", channels=Cc).to(dev)
with torch.no_grad():        # non-zero "zero-init" layers (model/nn_utils.py:151-157) so every kernel does real work
    for prm in net.parameters():
        if float(prm.abs().sum()) == 0.0:
            prm.normal_(0.0, 0.02)
T = torch.nn.Parameter(torch.FloatTensor([1.0]), requires_grad=False)
gen = PluginReverseSDE(SGMsde(T=T, num_steps_forward=16, device=dev), net, T, deviceReverseSDE=dev).to(dev)
if not os.environ.get("EM_ONLY"):
    tr = UNetScoreTrainer(gen, B, d, lr=1e-4, use_graph=not os.environ.get("NO_GRAPH"))
    tr.set_data(random_images(B, Cc, S_, S_, device=dev))
    for _ in range(2):
        l = tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        l = tr.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"B={B} train step {dt*1e3:.1f} ms  loss {float(l):.4f} -> {6*5.974e9*B/dt/1e12:.1f} TFLOP/s (algorithmic) "
          f"mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
if os.environ.get("TRAIN_ONLY"):
    sys.exit(0)
x = torch.randn(B, d, device=dev)
st = gen.base_sde.struct()
rng = gen.base_sde.philox(dev)
def em():
    s = torch.full((B,), 0.5, device=dev)
    a = net(x, s)
    ops.sde_stage(x, x, 1.0, x, a, st, L.PROC_REVERSE, False, 0.5, 1e-3, 0.0, rng=rng, rng_step=0)
for _ in range(2):
    em()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    em()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"B={B} EM step {dt*1e3:.1f} ms -> {5.974e9*B/dt/1e12:.1f} TFLOP/s")
