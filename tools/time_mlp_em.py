#!/usr/bin/env python3
"""GPU box: k_mlp<EM> / k_mlp<FWD> time vs batch (fixed cost of the per-workgroup prologue vs per-row cost)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdeflow_light_amd.NN import MLP
from sdeflow_light_amd.SDEs import SGMsde
from sdeflow_light_amd import ops, _lib as L
dev = torch.device("cuda")
net = MLP(input_dim=2).to(dev)
P = net.kernel_params()
T = torch.nn.Parameter(torch.FloatTensor([1.0]), requires_grad=False)
sde = SGMsde(T=T, device=dev)
st = sde.struct()
rng = sde.philox(dev)
for B in (4096, 16384, 65536, 131072, 262144, 1048576):
    x = torch.randn(B, 2, device=dev)
    f = lambda i: ops.check(ops.lib().msgm_mlp_em_step(P, x.data_ptr(), B, st, 0.5, 1e-3, 0.0, None, rng.ptr(), i, ops.stream()), "em")
    for i in range(5):
        f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(100):
        f(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    print(f"B={B:8d}: {us:8.1f} us  {66816 * B / us / 1e6:6.1f} TFLOP/s")
