"""1x1 weight-gradient kernels on the C4 shapes (GPU box): the pixel-streaming kernel (k_wgrad1x1, default) against the
32 x 32-block tile kernel it replaces (MSGM_NO_WGRAD1X1=1 in a second process).  Usage: python tools/bench_wgrad1x1.py [B]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdeflow_light_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = 2 * B
# (C, Cout, H, W) — attention qkv / proj (1-D token grids) and the ResBlock skip convolutions of the C4 U-Net
shapes = [(64, 192, 1, 1024), (64, 64, 1, 1024), (128, 384, 1, 256), (128, 128, 1, 256),
          (32, 64, 32, 32), (64, 128, 16, 16), (128, 128, 16, 16), (64, 128, 16, 16), (128, 64, 32, 32), (64, 64, 32, 32),
          (32, 64, 32, 32), (64, 32, 64, 64), (32, 32, 64, 64)]
dev = "cuda"
tot = 0.0
for C, Cout, H, W in shapes:
    P = H * W
    gy = torch.randn(N * P, Cout, device=dev)
    x = torch.randn(N * P, C, device=dev)
    geom = ops.conv_geom(N, H, W, H, W, 1, 1, 1, 0, 0, 0)
    dWp = torch.zeros(Cout * C, device=dev)
    db = torch.zeros(Cout, device=dev)
    f = lambda: ops.conv_wgrad(geom, gy, x, C, 0, dWp, Cout, Cout, C, dbias=db, n_bias=B)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * N * P * C * Cout
    by = 4.0 * N * P * (C + Cout)
    tot += us
    print(f"wgrad1x1 C={C:4d} Cout={Cout:4d} M={N*P:8d}: {us:8.1f} us  {fl/us/1e6:6.1f} TFLOP/s  {by/us/1e3:7.1f} GB/s (incl. slot reduce)", flush=True)
print(f"total {tot:.0f} us")
