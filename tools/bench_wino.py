#!/usr/bin/env python3
"""GPU box: Winograd F(2x2,3x3) forward and the opt-in bf16-split kernel vs the direct halo-tile kernel at the C5 layer shapes
(N = 1024 rows; AFF=1: with the folded GroupNorm + SiLU input transform, as the sampler runs them)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdeflow_light_amd import ops
from sdeflow_light_amd.convnet import ConvOp
dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
def timeit(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / it * 1e-3
for (H, C0, C1, Co) in ((64, 32, 0, 32), (64, 64, 32, 32), (32, 64, 0, 64), (32, 128, 64, 64), (16, 128, 0, 128), (16, 128, 128, 128)):
    w = torch.nn.Parameter(torch.randn(Co, C0 + C1, 3, 3, device=dev) * 0.05)
    op = ConvOp(w, None, "conv", (3, 3), 1, 1, [C0, C1] if C1 else [C0])
    op.pack(); ops.PackTable(op.wino_jobs(), dev).run_wino(); op.pack_b6()
    srcs = [torch.randn(N * H * H * C0, device=dev)] + ([torch.randn(N * H * H * C1, device=dev)] if C1 else [])
    out = torch.empty(N * H * H * Co, device=dev)
    fl = 2 * 9 * (C0 + C1) * Co * N * H * H
    ctot = C0 + C1
    aff = (1 + 0.3 * torch.randn(N * ctot, device=dev), 0.2 * torch.randn(N * ctot, device=dev)) if os.environ.get("AFF") else None
    kw = dict(in_affine=aff, in_act=1 if aff is not None else 0)
    td = timeit(lambda: op.forward(srcs, N, H, H, N, out=out, **kw))
    tw = timeit(lambda: op.forward(srcs, N, H, H, N, out=out, wino=True, **kw))
    tb = timeit(lambda: op.forward(srcs, N, H, H, N, out=out, b6=True, **kw))
    print(f"{H}x{H} {C0}+{C1}->{Co}: direct {td*1e3:7.3f} ms = {fl/td/1e12:6.1f} TF/s | winograd {tw*1e3:7.3f} ms = {fl/tw/1e12:6.1f} TF/s as-written ({fl/2.25/tw/1e12:5.1f} executed)  x{td/tw:.2f}"
          f" | bf16 split {tb*1e3:7.3f} ms = {fl/tb/1e12:6.1f} TF/s fp32-equivalent  x{td/tb:.2f}")
