#!/usr/bin/env python3
"""GPU box: the dual GroupNorm(+SiLU) passes of the training step as HBM streams (algorithmic bytes / time), at the
largest layers of the C4 step (B = 256: 64x64x32, 32x32x64, 16x16x128)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdeflow_light_amd import ops
dev = "cuda"
Bp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
def timeit(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
for (H, C) in ((64, 32), (32, 64), (16, 128), (32, 192)):
    P, G = H * H, 32
    n = 2 * Bp * P * C                                     # primal | tangent
    x = torch.randn(n, device=dev); g = torch.randn(n, device=dev)
    gam, bet = torch.randn(C, device=dev), torch.randn(C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    st = torch.empty(Bp * G * 4, device=dev)
    out = torch.empty(n, device=dev)
    tf = timeit(lambda: ops.groupnorm_dual_forward(x, gam, bet, Bp, P, C, G, True, True, stats=st, out=out))
    tb = timeit(lambda: ops.groupnorm_dual_backward(x, gam, bet, st, g, dg, db, Bp, P, C, G, True))
    by = 4.0 * n
    # forward: reduce reads x (1 tensor pair), apply reads x and writes out -> 3 passes; backward: reduce reads x, g; apply
    # reads x, g and writes gx -> 5 passes
    print(f"GN dual {H}x{H}x{C} Bp={Bp}: forward {tf*1e6:7.1f} us = {3*by/tf/1e12:.2f} TB/s over 3 passes;  "
          f"backward {tb*1e6:7.1f} us = {5*by/tb/1e12:.2f} TB/s over 5 passes   ({by/2**20:.0f} MiB per (primal|tangent) tensor)")
