"""Synthetic inputs of the BASELINE configs.  The reference has
no Gaussian-mixture sampler (its synthetic samplers are SwissRoll / Gaussian /
Cauchy, data.py:702-802); the benchmark mixture is build-defined: 8 equal
components on the circle r=2, isotropic sigma=0.15."""
from __future__ import annotations

import math

import torch


def gaussian_mixture_2d(n: int, seed: int = 1234, device="cpu") -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    k = torch.randint(0, 8, (n,), generator=g)
    ang = 2 * math.pi * k.float() / 8
    mean = torch.stack([2 * torch.cos(ang), 2 * torch.sin(ang)], 1)
    return (mean + 0.15 * torch.randn(n, 2, generator=g)).to(device)


def signals_1d(n: int, L: int = 1024, seed: int = 1234, device="cpu") -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    j = torch.arange(L).float()[None, None, :]
    A = 0.5 + 0.5 * torch.rand(n, 3, 1, generator=g)
    f = torch.randint(1, 17, (n, 3, 1), generator=g).float()
    ph = 2 * math.pi * torch.rand(n, 3, 1, generator=g)
    x = (A * torch.sin(2 * math.pi * f * j / L + ph)).sum(1) + 0.05 * torch.randn(n, L, generator=g)
    return x.to(device)


def random_images(n: int, C: int = 3, H: int = 64, W: int = 64, seed: int = 1234, device="cpu") -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, C * H * W, generator=g).to(device)
