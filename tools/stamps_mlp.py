#!/usr/bin/env python3
"""Diagnostic (GPU box): build variants of mlp_kernels.hip (-D flags) into /tmp, time k_mlp<TRAIN> at C2 for
each, and print the per-phase cycle breakdown (workgroup 0, wave 0) of the -DMLP_STAMPS builds.
Usage: stamps_mlp.py ["-DPF=1 -DMLP_HINTS=0" ...]   (each argument = one variant; default = shipped flags)"""
import ctypes as C, glob, os, subprocess, sys, importlib
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = sorted(glob.glob(os.path.join(ROOT, "sdeflow_light_amd", "csrc", "*.hip")))
variants = sys.argv[1:] or [""]
dev = "cuda"
B, d = 65536, 2
torch.manual_seed(0)
W = [torch.randn(128, 3, device=dev) * .5, torch.zeros(128, device=dev), torch.randn(128, 128, device=dev) * .09, torch.zeros(128, device=dev),
     torch.randn(128, 128, device=dev) * .09, torch.zeros(128, device=dev), torch.randn(2, 128, device=dev) * .09, torch.zeros(2, device=dev)]
y, t, v = torch.randn(B, d, device=dev), torch.rand(B, device=dev), torch.randn(B, d, device=dev).sign()
names = ["p0 h0-build", "p1 layer1", "p2 layer2 gemm", "p3 layer3 gemm+L4 partial", "p4 loss (16 thr)", "p5 L4 bwd+dW4", "p6 dgrad3+dW3",
         "p7 dgrad2+dW2", "p8 dW1", "prologue->epilogue", "epilogue"]
from sdeflow_light_amd import _lib
for vi, flags in enumerate(variants):
    for stamps in (False, True):
        so = f"/tmp/libmsgm_var{vi}_{int(stamps)}.so"
        cmd = ["hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-shared"] + flags.split() + \
              (["-DMLP_STAMPS"] if stamps else []) + src + ["-o", so]
        subprocess.check_call(cmd)
        L = C.CDLL(so)
        for name, (res, args) in _lib.SIGNATURES.items():
            fn = getattr(L, name); fn.restype, fn.argtypes = res, args
        P = _lib.MlpParamsT(*[w.data_ptr() for w in W], d, 0)
        st = _lib.sde_struct(0, 0.1, 20.0, 1.0, 1e-3)
        ws = torch.empty(int(L.msgm_mlp_ssm_workspace(d, 0)) // 4, device=dev)
        nsl = C.c_int32(0)
        run = lambda: L.msgm_mlp_ssm_partial(P, y.data_ptr(), t.data_ptr(), v.data_ptr(), None, None, B, st, 1.0 / B, None, ws.data_ptr(),
                                             ws.numel() * 4, C.byref(nsl), torch.cuda.current_stream().cuda_stream)
        for _ in range(5):
            assert run() == 0
        torch.cuda.synchronize()
        if not stamps:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                run()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 50 * 1e3
            print(f"variant[{flags or 'default'}]: k_mlp<TRAIN> {us:.1f} us  -> {400896 * B / us / 1e6:.1f} TFLOP/s")
            x = y.clone()
            rng = torch.tensor([1234, 0], dtype=torch.int64, device=dev)
            em = lambda i: L.msgm_mlp_em_step(P, x.data_ptr(), B, st, 0.5, 1e-3, 0.0, None, rng.data_ptr(), i,
                                              torch.cuda.current_stream().cuda_stream)
            for i in range(5):
                assert em(i) == 0
            torch.cuda.synchronize()
            e0.record()
            for i in range(100):
                em(i)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 100 * 1e3
            print(f"variant[{flags or 'default'}]: k_mlp<EM> {us:.1f} us  -> {66816 * B / us / 1e6:.1f} TFLOP/s")
        else:
            L.msgm_debug_stamps.restype = C.c_int; L.msgm_debug_stamps.argtypes = [C.c_void_p]
            buf = (C.c_ulonglong * 12)()
            assert L.msgm_debug_stamps(buf) == 0
            tot = sum(buf[:9]); tiles = (B // 16 + 255) // 256
            print(f"  cycles/tile {tot / tiles:.0f} (MFMA-only 26368): " + " | ".join(f"{n.split()[0]} {c / tiles:.0f}" for n, c in zip(names[:9], buf)))
