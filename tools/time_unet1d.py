#!/usr/bin/env python3
"""GPU box: time one UNet1D (C3) train step and EM step at a given batch."""
import sys, time, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdeflow_light_amd.NNUnet1D import UNet1D
from sdeflow_light_amd.SDEs import SGMsde, PluginReverseSDE
from sdeflow_light_amd.train import UNetScoreTrainer
from sdeflow_light_amd.data import signals_1d
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda")
torch.manual_seed(0)
net = UNet1D(1024).to(dev)
T = torch.nn.Parameter(torch.FloatTensor([1.0]), requires_grad=False)
gen = PluginReverseSDE(SGMsde(T=T, num_steps_forward=16, device=dev), net, T, deviceReverseSDE=dev).to(dev)
tr = UNetScoreTrainer(gen, B, 1024, lr=1e-4)
tr.set_data(signals_1d(B, device=dev))
for _ in range(2):
    l = tr.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    l = tr.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
flop = 6 * 0.4554e9 * B
print(f"B={B} train step {dt*1e3:.1f} ms  loss {float(l):.4f}  -> {flop/dt/1e12:.1f} TFLOP/s (algorithmic, as-written FLOPs) "
      f"mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
x = torch.randn(B, 1024, device=dev)
s = torch.full((B,), 0.5, device=dev)
for _ in range(2):
    a = net(x, s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    a = net(x, s)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"B={B} forward {dt*1e3:.1f} ms -> {0.4554e9*B/dt/1e12:.1f} TFLOP/s")
