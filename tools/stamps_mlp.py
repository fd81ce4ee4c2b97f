#!/usr/bin/env python3
"""Diagnostic (GPU box): build mlp_kernels.hip with -DMLP_STAMPS into /tmp and
print where a k_mlp<TRAIN> tile spends its cycles (workgroup 0, wave 0)."""
import ctypes as C, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "sdeflow_light_amd", "csrc")
so = "/tmp/libmsgm_stamps.so"
subprocess.check_call(["hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-DMLP_STAMPS",
                       "-shared", os.path.join(src, "mlp_kernels.hip"), os.path.join(src, "sde_kernels.hip"), "-o", so])
from sdeflow_light_amd import _lib
_lib.LIB_PATH = so
from sdeflow_light_amd import ops
L = ops.lib()
L.msgm_debug_stamps.restype = C.c_int
L.msgm_debug_stamps.argtypes = [C.c_void_p]
dev = "cuda"
B, d = 65536, 2
W = [torch.randn(128, 3, device=dev) * .5, torch.zeros(128, device=dev), torch.randn(128, 128, device=dev) * .09, torch.zeros(128, device=dev),
     torch.randn(128, 128, device=dev) * .09, torch.zeros(128, device=dev), torch.randn(2, 128, device=dev) * .09, torch.zeros(2, device=dev)]
P = ops.mlp_params(*W, premodule=False)
st = _lib.sde_struct(0, 0.1, 20.0, 1.0, 1e-3)
y, t, v = torch.randn(B, d, device=dev), torch.rand(B, device=dev), torch.randn(B, d, device=dev).sign()
g = torch.empty(ops.mlp_num_params(d, False), device=dev)
ws = ops.mlp_ssm_workspace(d, False, dev)
for _ in range(3):
    ops.mlp_ssm_grad(P, y, t, v, st, 1.0 / B, g, ws)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 12)()
assert L.msgm_debug_stamps(buf) == 0
names = ["p0 h0-build", "p1 layer1", "p2 layer2 gemm", "p3 layer3 gemm+L4 partial", "p4 loss (16 thr)", "p5 L4 bwd+dW4", "p6 dgrad3+dW3",
         "p7 dgrad2+dW2", "p8 dW1", "prologue->epilogue", "epilogue"]
tot = sum(buf[:9])
tiles = (B // 16 + 255) // 256
print(f"tiles per WG: {tiles}; cycles per tile: {tot / tiles:.0f} (ideal MFMA-only 824*32 = 26368)")
for n, c in zip(names, buf):
    print(f"  {n:28s} {c / tiles:9.0f} cyc/tile  {100.0 * c / max(tot, 1):5.1f} %")
