#!/usr/bin/env python3
"""GPU box: time individual conv / wgrad / bmm shapes of the C4 U-Net (dual batch 512) — TFLOP/s per kernel."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdeflow_light_amd import ops
dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512


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


print("conv forward (3x3 s1):")
for (H, Ci, Co) in ((64, 32, 32), (64, 96, 32), (32, 64, 64), (32, 192, 64), (16, 128, 128), (16, 256, 128)):
    x = torch.randn(N * H * H * Ci, device=dev)
    Wp = torch.randn(9 * ops.pad16(Co) * ops.pad16(Ci), device=dev) * 0.05
    out = torch.empty(N * H * H * Co, device=dev)
    geom = ops.conv_geom(N, H, H, H, H, 3, 3, 1, 1)
    fl = 2 * 9 * Ci * Co * N * H * H
    res = []
    for env in ("1", None):
        if env:
            os.environ["MSGM_NO_CONV_TILE"] = env
        else:
            os.environ.pop("MSGM_NO_CONV_TILE", None)
        t = timeit(lambda: ops.conv_forward(geom, x, Ci, Wp, Co, out, n_bias=N // 2))
        res.append(fl / t / 1e12)
    gy = torch.randn(N * H * H * Co, device=dev)
    dWp = torch.zeros(9 * ops.pad16(Co) * ops.pad16(Ci), device=dev)
    os.environ["MSGM_NO_WGRAD_TILE"] = "1"
    tw = timeit(lambda: ops.conv_wgrad(geom, gy, x, Ci, 0, dWp, Co, ops.pad16(Co), ops.pad16(Ci)))
    os.environ.pop("MSGM_NO_WGRAD_TILE")
    tw2 = timeit(lambda: ops.conv_wgrad(geom, gy, x, Ci, 0, dWp, Co, ops.pad16(Co), ops.pad16(Ci)))
    print(f"  {H}x{H} {Ci:3d}->{Co:3d}: gemm {res[0]:5.1f} TF/s | tile {res[1]:5.1f} TF/s | wgrad {fl / tw / 1e12:5.1f} -> tile {fl / tw2 / 1e12:5.1f} TF/s")
print("conv1d (k3 s1), dual batch 8192:")
for (L, Ci, Co) in ((1024, 32, 32), (1024, 64, 32), (512, 64, 64), (512, 128, 64), (256, 128, 128), (256, 256, 128)):
    Nn = 8192
    x = torch.randn(Nn * L * Ci, device=dev)
    Wp = torch.randn(3 * ops.pad16(Co) * ops.pad16(Ci), device=dev) * 0.05
    out = torch.empty(Nn * L * Co, device=dev)
    geom = ops.conv_geom(Nn, 1, L, 1, L, 1, 3, 1, 1)
    fl = 2 * 3 * Ci * Co * Nn * L
    res = []
    for env in ("1", None):
        if env:
            os.environ["MSGM_NO_CONV_TILE"] = env
        else:
            os.environ.pop("MSGM_NO_CONV_TILE", None)
        res.append(fl / timeit(lambda: ops.conv_forward(geom, x, Ci, Wp, Co, out, n_bias=Nn // 2), 5) / 1e12)
    gy = torch.randn(Nn * L * Co, device=dev)
    dWp = torch.zeros(3 * ops.pad16(Co) * ops.pad16(Ci), device=dev)
    os.environ["MSGM_NO_WGRAD_TILE"] = "1"
    tw = timeit(lambda: ops.conv_wgrad(geom, gy, x, Ci, 0, dWp, Co, ops.pad16(Co), ops.pad16(Ci)), 5)
    os.environ.pop("MSGM_NO_WGRAD_TILE")
    tw2 = timeit(lambda: ops.conv_wgrad(geom, gy, x, Ci, 0, dWp, Co, ops.pad16(Co), ops.pad16(Ci)), 5)
    print(f"  L={L} {Ci:3d}->{Co:3d}: gemm {res[0]:5.1f} | tile {res[1]:5.1f} | wgrad {fl / tw / 1e12:5.1f} -> tile {fl / tw2 / 1e12:5.1f} TF/s")
print("attention products (per block, batch = N/2):")
for (T, C) in ((1024, 64), (256, 128)):
    Bp = N // 2
    ld = 3 * C
    qkv = torch.randn(N * T * ld, device=dev)
    S = torch.empty(Bp * T * T, device=dev)
    att = torch.empty(N * T * C, device=dev)
    fl = 2 * T * T * C * Bp
    t1 = timeit(lambda: ops.bmm(qkv, 0, qkv, C, S, 0, T, T, C, Bp, (T * ld, ld, 1), (T * ld, 1, ld), (T * T, T, 1)))
    t2 = timeit(lambda: ops.bmm(S, 0, qkv, 2 * C, att, 0, T, C, T, Bp, (T * T, T, 1), (T * ld, ld, 1), (T * C, C, 1)))
    t3 = timeit(lambda: ops.bmm(S, 0, att, 0, qkv, 2 * C, T, C, T, Bp, (T * T, 1, T), (T * C, C, 1), (T * ld, ld, 1)))
    print(f"  T={T} C={C}: QK^T (KVEC) {fl / t1 / 1e12:5.1f} TF/s | P.V (B j-contig) {fl / t2 / 1e12:5.1f} TF/s | P^T.a (A i-contig) {fl / t3 / 1e12:5.1f} TF/s")
