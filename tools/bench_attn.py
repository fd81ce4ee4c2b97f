#!/usr/bin/env python3
"""GPU box: the training-path attention at the C4 block shapes — fused dual forward / backward
(msgm_attention_dual_*) in TFLOP/s of EXECUTED products (forward 6, backward 15 x 2 T^2 C per sample) and of the
products as written upstream (6 / 12), next to the composed bmm + dual-softmax chain it replaces."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdeflow_light_amd import ops  # noqa: E402

dev = "cuda"
Bp = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def timeit(fn, it=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / it * 1e-3


for (T, C) in ((1024, 64), (256, 64), (64, 32)):
    if not ops.attention_dual_supported(T, C):
        continue
    N = 2 * Bp
    qkv = torch.randn(N * T * 3 * C, device=dev)
    s2 = 1.0 / math.sqrt(C)
    att, stats = ops.attention_dual_forward(qkv, Bp, T, C, s2)
    datt = torch.randn(N * T * C, device=dev)
    ops.attention_dual_backward(qkv, att, datt, stats, Bp, T, C, s2)
    prod = 2.0 * T * T * C * Bp
    tf = timeit(lambda: ops.attention_dual_forward(qkv, Bp, T, C, s2))
    tb = timeit(lambda: ops.attention_dual_backward(qkv, att, datt, stats, Bp, T, C, s2))
    print(f"T={T} C={C} Bp={Bp}: dual fwd {tf * 1e3:7.3f} ms = {6 * prod / tf / 1e12:6.1f} TF/s | "
          f"dual bwd {tb * 1e3:7.3f} ms = {15 * prod / tb / 1e12:6.1f} TF/s executed ({12 * prod / tb / 1e12:6.1f} as written)")
