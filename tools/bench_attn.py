#!/usr/bin/env python3
"""GPU box: fused attention forward (sampler path) at the U-Net's two shapes."""
import os, sys, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdeflow_light_amd import ops
dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for T, C in ((1024, 64), (256, 128)):
    qkv = torch.randn(N * T * 3 * C, device=dev)
    out = torch.empty(N * T * C, device=dev)
    f = lambda: ops.attention_forward(qkv, out, N, T, C, 1.0 / math.sqrt(C))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"T={T} C={C} N={N}: {us:8.1f} us  {4 * T * T * C * N / us / 1e6:6.1f} TFLOP/s")
