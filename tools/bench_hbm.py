#!/usr/bin/env python3
"""GPU box: what HBM streams reach from plain kernels — fill (write only), copy (read + write), sum (read only) on 1 GiB."""
import torch
n = 256 * 1024 * 1024
a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
by = 4.0 * n
print(f"fill  (write only): {by / t(lambda: a.fill_(1.0)) / 1e12:.2f} TB/s")
print(f"copy  (read+write): {2 * by / t(lambda: b.copy_(a)) / 1e12:.2f} TB/s of traffic")
print(f"sum   (read only) : {by / t(lambda: a.sum()) / 1e12:.2f} TB/s")
print(f"axpy  (2 reads + 1 write): {3 * by / t(lambda: torch.add(a, b, alpha=2.0, out=b)) / 1e12:.2f} TB/s of traffic")
