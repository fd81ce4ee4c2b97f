#!/usr/bin/env python3
"""Diagnostic (GPU box): build variants of the library with -D flags into /tmp and time one halo-tile conv shape
for each (what bounds k_conv_tile: weight-fragment loads, LDS activation reads, or neither).
usage: exp_conv.py N H Ci Co ["-DCT_EXP_NOW" "-DCT_EXP_NOLDS" ...]"""
import ctypes as C, glob, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdeflow_light_amd import _lib
N, H, Ci, Co = (int(a) for a in sys.argv[1:5])
variants = [""] + sys.argv[5:]
src = sorted(glob.glob(os.path.join(ROOT, "sdeflow_light_amd", "csrc", "*.hip")))
dev = "cuda"
pad16 = lambda c: (c + 15) // 16 * 16
x = torch.randn(N * H * H * Ci, device=dev)
Wp = torch.randn(9 * pad16(Co) * pad16(Ci), device=dev) * 0.05
out = torch.empty(N * H * H * Co, device=dev)
geom = _lib.ConvGeomT(N, H, H, H, H, 3, 3, 1, 1, 1, 1, 0, 0)
fl = 2 * 9 * Ci * Co * N * H * H
for vi, flags in enumerate(variants):
    so = f"/tmp/libmsgm_exp{vi}.so"
    subprocess.check_call(["hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-shared"] +
                          flags.split() + src + ["-o", so])
    L = C.CDLL(so)
    fn = L.msgm_conv_forward
    fn.restype, fn.argtypes = _lib.SIGNATURES["msgm_conv_forward"]
    run = lambda: fn(geom, x.data_ptr(), Ci, None, 0, Wp.data_ptr(), Co, pad16(Co), pad16(Ci), None, None, N // 2, N // 2,
                     out.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"variant[{flags or 'shipped'}]: {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s")
