#!/usr/bin/env python3
"""GPU box: graph-captured reverse-SDE step of the C5 net (VorticityUNet 64x64x3) at 1024 rows — Euler-Maruyama vs Heun
vs RK4 (what MSGM_higherDim.py:903 generates with).  RK4 = 4 score-net evaluations: the target is <= 1.03 x 4 EM steps."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sdeflow_light_amd.sde_scheme import GraphedStepSampler  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
gen, d = bench.build_unet("c5", dev)
rows, N = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 6
x = gen.latent_sample(rows, d)
res = {}
for m in ("em", "heun", "rk4"):
    gs = GraphedStepSampler(gen, rows, d, N, method=m)
    gs.run(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = gs.run(x)
    torch.cuda.synchronize()
    res[m] = (time.perf_counter() - t0) / N
    print(f"{m:5s}: {res[m] * 1e3:8.2f} ms per step at {rows} rows (finite={bool(torch.isfinite(out).all())})", flush=True)
    del gs
print(f"rk4 / (4 x em) = {res['rk4'] / (4 * res['em']):.3f}   heun / (2 x em) = {res['heun'] / (2 * res['em']):.3f}")
