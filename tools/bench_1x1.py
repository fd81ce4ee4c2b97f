#!/usr/bin/env python3
"""GPU box: the 1x1 convolutions of the attention blocks (qkv / proj) as HBM streams: achieved GB/s of algorithmic bytes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdeflow_light_amd import ops
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
for (T, Ci, Co) in ((1024, 64, 64), (1024, 64, 192), (1024, 192, 64), (256, 128, 128), (256, 128, 384), (256, 384, 128), (256, 256, 128), (256, 192, 128),
                    (1024, 128, 64), (4096, 96, 32)):
    x = torch.randn(N * T * Ci, device=dev)
    Wp = torch.randn(ops.pad16(Co) * ops.pad16(Ci), device=dev) * 0.05
    out = torch.empty(N * T * Co, device=dev)
    geom = ops.conv_geom(N, 1, T, 1, T, 1, 1, 1, 0)
    t = timeit(lambda: ops.conv_forward(geom, x, Ci, Wp, Co, out, n_bias=N))
    by = 4.0 * N * T * (Ci + Co)
    print(f"1x1 T={T} {Ci}->{Co} N={N}: {t*1e3:7.3f} ms  {by/t/1e12:5.2f} TB/s of (in+out) bytes  {2*Ci*Co*N*T/t/1e12:6.1f} TF/s")
