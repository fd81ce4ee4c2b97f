#!/usr/bin/env python3
"""GPU box: the conv / wgrad probes of bench.py (3x3 64 -> 64 @ 32x32, dual batch 512) launched a few times, for the PMC
passes (`rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` / `MfmaUtil`, one counter per pass, folded by tools/pmc_fold.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdeflow_light_amd import ops
dev = "cuda"
Bp, H, Ci, Co = 256, 32, 64, 64
N = 2 * Bp
x = torch.randn(N * H * H * Ci, device=dev)
gy = torch.randn(N * H * H * Co, device=dev)
Wp = torch.randn(9 * ops.pad16(Co) * ops.pad16(Ci), device=dev) * 0.05
dWp = torch.zeros(9 * ops.pad16(Co) * ops.pad16(Ci), device=dev)
out = torch.empty(N * H * H * Co, device=dev)
geom = ops.conv_geom(N, H, H, H, H, 3, 3, 1, 1)
for _ in range(6):
    ops.conv_forward(geom, x, Ci, Wp, Co, out, n_bias=Bp)
    ops.conv_wgrad(geom, gy, x, Ci, 0, dWp, Co, ops.pad16(Co), ops.pad16(Ci))
torch.cuda.synchronize()
print("algorithmic bytes: conv", 4 * (x.numel() + out.numel() + Wp.numel()), " wgrad", 4 * (x.numel() + gy.numel() + dWp.numel()))
