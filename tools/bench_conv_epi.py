#!/usr/bin/env python3
"""GPU box: what the epilogue options of the 3x3 halo-tile conv cost (plain / residual / accumulate / stats), sampler shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdeflow_light_amd import ops
dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
def timeit(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
for (H, Ci, Co) in ((32, 64, 64), (64, 32, 32), (16, 128, 128)):
    x = torch.randn(N * H * H * Ci, device=dev)
    Wp = torch.randn(9 * ops.pad16(Co) * ops.pad16(Ci), device=dev) * 0.05
    out = torch.zeros(N * H * H * Co, device=dev)
    res = torch.randn(N * H * H * Co, device=dev)
    sb = torch.randn(N * Co, device=dev)
    geom = ops.conv_geom(N, H, H, H, H, 3, 3, 1, 1)
    S = ops.conv_chanstats_slots(geom, Ci, 0, Co, ops.pad16(Co))
    cs = torch.empty(N * S * 2 * Co, device=dev)
    fl = 2 * 9 * Ci * Co * N * H * H
    r = {}
    r["plain"] = timeit(lambda: ops.conv_forward(geom, x, Ci, Wp, Co, out, n_bias=N))
    r["samp_bias"] = timeit(lambda: ops.conv_forward(geom, x, Ci, Wp, Co, out, n_bias=N, samp_bias=sb))
    r["residual"] = timeit(lambda: ops.conv_forward(geom, x, Ci, Wp, Co, out, n_bias=N, residual=res))
    r["accumulate"] = timeit(lambda: ops.conv_forward(geom, x, Ci, Wp, Co, out, n_bias=N, accumulate=True))
    r["residual+stats"] = timeit(lambda: ops.conv_forward(geom, x, Ci, Wp, Co, out, n_bias=N, residual=res, chanstats=cs))
    print(f"3x3 {Ci}->{Co} @{H}x{H} N={N}: " + "  ".join(f"{k} {v*1e3:.3f} ms ({fl/v/1e12:.0f} TF/s)" for k, v in r.items()))
print("--- input transform (GroupNorm affine + SiLU applied while staging) ---")
for (H, Ci, Co) in ((32, 64, 64), (64, 32, 32), (16, 128, 128), (64, 96, 32)):
    x = torch.randn(N * H * H * Ci, device=dev)
    Wp = torch.randn(9 * ops.pad16(Co) * ops.pad16(Ci), device=dev) * 0.05
    out = torch.zeros(N * H * H * Co, device=dev)
    sc, sh = torch.rand(N * Ci, device=dev) + 0.5, torch.randn(N * Ci, device=dev)
    geom = ops.conv_geom(N, H, H, H, H, 3, 3, 1, 1)
    fl = 2 * 9 * Ci * Co * N * H * H
    r = {}
    r["plain"] = timeit(lambda: ops.conv_forward(geom, x, Ci, Wp, Co, out, n_bias=N))
    r["affine"] = timeit(lambda: ops.conv_forward(geom, x, Ci, Wp, Co, out, n_bias=N, in_scale=sc, in_shift=sh, in_act=0))
    r["affine+SiLU"] = timeit(lambda: ops.conv_forward(geom, x, Ci, Wp, Co, out, n_bias=N, in_scale=sc, in_shift=sh, in_act=1))
    print(f"3x3 {Ci}->{Co} @{H}x{H} N={N}: " + "  ".join(f"{k} {v*1e3:.3f} ms ({fl/v/1e12:.0f} TF/s)" for k, v in r.items()))
