#!/usr/bin/env python3
"""GPU box: run ONE conv shape repeatedly (for PMC passes).  usage: one_conv.py N H Ci Co [reps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdeflow_light_amd import ops
N, H, Ci, Co = (int(a) for a in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = "cuda"
x = torch.randn(N * H * H * Ci, device=dev)
Wp = torch.randn(9 * ops.pad16(Co) * ops.pad16(Ci), device=dev) * 0.05
out = torch.empty(N * H * H * Co, device=dev)
geom = ops.conv_geom(N, H, H, H, H, 3, 3, 1, 1)
for _ in range(reps):
    ops.conv_forward(geom, x, Ci, Wp, Co, out, n_bias=N // 2)
torch.cuda.synchronize()
print("done")
