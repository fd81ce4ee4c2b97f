#!/usr/bin/env python3
"""GPU box: the HBM-bound kernels of the path at the C5 state size (1024 rows x 12288) and at a large flat size,
as algorithmic GB/s against the 8 TB/s HBM3E peak (perturb 8 B/elt, EM stage 12 B/elt, Adam 28 B/param,
lincomb 12 B/elt, GELU/SiLU dual 16 B/elt)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdeflow_light_amd import ops, _lib as L
dev = torch.device("cuda")
st = L.sde_struct(0, 0.1, 20.0, 1.0, 1e-3)
rng = L.PhiloxState(7, dev)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for B, d in ((1024, 12288), (8192, 12288)):
    n = B * d
    x = torch.randn(B, d, device=dev); a = torch.randn(B, d, device=dev); out = torch.empty_like(x)
    rows = []
    rows.append(("perturb_vp (K1)", 8 * n, timeit(lambda: ops.perturb_vp(x, st, rng=rng))))
    rows.append(("EM stage diag (K2)", 12 * n, timeit(lambda: ops.sde_stage(x, x, 1.0, x, a, st, L.PROC_REVERSE, False, 0.5, 1e-3, 0.0, rng=rng, rng_step=1))))
    rows.append(("lincomb x+y", 12 * n, timeit(lambda: ops.lincomb(out, x, 1.0, a, 1.0))))
    z = torch.randn(2 * n, device=dev); h = torch.empty_like(z)
    rows.append(("SiLU dual fwd", 16 * n, timeit(lambda: ops.act_dual_forward(1, z, h, True))))
    p, g, m, v = (torch.randn(n, device=dev) for _ in range(4)); v.abs_()
    rows.append(("Adam (K13)", 28 * n, timeit(lambda: ops.adam_step(p, g, m, v, step=3, lr=1e-4))))
    print(f"state {B} x {d} ({4 * n / 2**20:.0f} MiB per tensor)")
    for name, byts, t in rows:
        print(f"  {name:22s} {t * 1e6:9.1f} us  {byts / t / 1e9:8.1f} GB/s  ({byts / t / 8e12:.2f} of 8 TB/s)")
