#!/usr/bin/env python3
"""GPU box: A/B of the 256-pixel-tile variant of the halo-tile convolution (3-tap kernels) (MSGM_NO_CONV_WIDE)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdeflow_light_amd import ops
dev = "cuda"


def timeit(fn, it=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3


def ab(tag, geom, x, Ci, Wp, Co, out, N, fl):
    r = []
    for off in (True, False):
        if off:
            os.environ["MSGM_NO_CONV_WIDE"] = "1"
        else:
            os.environ.pop("MSGM_NO_CONV_WIDE", None)
        t = timeit(lambda: ops.conv_forward(geom, x, Ci, Wp, Co, out, n_bias=N // 2))
        r.append((fl / t / 1e12, out.clone()))
    same = torch.equal(r[0][1], r[1][1])
    print(f"  {tag}: 128-px {r[0][0]:5.1f} TF/s | 256-px {r[1][0]:5.1f} TF/s | identical {same}")


print("2-D 3x3 / 1x1, N=512:")
for (H, Ci, Co, K) in ((64, 32, 32, 3), (64, 96, 32, 3), (32, 32, 32, 3), (32, 64, 32, 3), (28, 32, 32, 3), (28, 64, 32, 3),
                       (32, 64, 64, 3), (32, 192, 64, 3), (16, 128, 128, 3), (16, 256, 128, 3)):
    N = 512
    x = torch.randn(N * H * H * Ci, device=dev)
    Wp = torch.randn(K * K * ops.pad16(Co) * ops.pad16(Ci), device=dev) * 0.05
    out = torch.empty(N * H * H * Co, device=dev)
    geom = ops.conv_geom(N, H, H, H, H, K, K, 1, (K - 1) // 2)
    ab(f"{H}x{H} k{K} {Ci:3d}->{Co}", geom, x, Ci, Wp, Co, out, N, 2 * K * K * Ci * Co * N * H * H)
print("1-D k3, N=8192:")
for (L, Ci, Co) in ((1024, 32, 32), (1024, 64, 32), (512, 64, 64), (512, 128, 64), (256, 128, 128)):
    N = 8192
    x = torch.randn(N * L * Ci, device=dev)
    Wp = torch.randn(3 * ops.pad16(Co) * ops.pad16(Ci), device=dev) * 0.05
    out = torch.empty(N * L * Co, device=dev)
    geom = ops.conv_geom(N, 1, L, 1, L, 1, 3, 1, 1)
    ab(f"L={L} {Ci:3d}->{Co}", geom, x, Ci, Wp, Co, out, N, 2 * 3 * Ci * Co * N * L)

print("XCD-aware grid (MSGM_CONV_NO_XCD = old order), N=512:")
for (H, Ci, Co, K) in ((32, 64, 192, 1), (32, 64, 64, 1), (16, 128, 384, 1), (16, 256, 128, 1), (16, 256, 128, 3), (32, 192, 64, 3), (16, 128, 128, 3)):
    N = 512
    x = torch.randn(N * H * H * Ci, device=dev)
    Wp = torch.randn(K * K * ops.pad16(Co) * ops.pad16(Ci), device=dev) * 0.05
    out = torch.empty(N * H * H * Co, device=dev)
    geom = ops.conv_geom(N, H, H, H, H, K, K, 1, (K - 1) // 2)
    fl = 2 * K * K * Ci * Co * N * H * H
    r = []
    for off in (True, False):
        if off:
            os.environ["MSGM_CONV_NO_XCD"] = "1"
        else:
            os.environ.pop("MSGM_CONV_NO_XCD", None)
        t = timeit(lambda: ops.conv_forward(geom, x, Ci, Wp, Co, out, n_bias=N // 2))
        r.append((t, out.clone()))
    gb = (x.numel() + out.numel()) * 4 / 1e9
    print(f"  {H}x{H} k{K} {Ci:3d}->{Co:3d}: old {fl / r[0][0] / 1e12:5.1f} TF/s {gb / r[0][0]:6.0f} GB/s | xcd {fl / r[1][0] / 1e12:5.1f} TF/s {gb / r[1][0]:6.0f} GB/s"
          f" | identical {torch.equal(r[0][1], r[1][1])}", flush=True)
