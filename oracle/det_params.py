"""Deterministic, closed-form parameter fill shared by the golden generator
and the tests (TEST INFRASTRUCTURE ONLY).

The U-Nets have 0.8 M / 4 M parameters — too large to commit as fixtures —
so both sides rebuild them from a formula instead of a stored state_dict.
The fill also overwrites the reference's zero-initialised layers
(model/nn_utils.py:151-157), which would otherwise make a fresh U-Net
output exactly 0 and hide bugs (SURVEY.md §7 "hard parts").
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Mapping, Sequence

import torch


def det_tensor(name: str, shape: Sequence[int], gain: float = 1.4) -> torch.Tensor:
    """value[i] = scale * sin(phi * i + phase(name)); quasi-uniform in
    [-scale, scale].  >=2-D tensors are fan-in scaled, 1-D '.weight'
    (norm gains) sit around 1, biases are small."""
    n = 1
    for s in shape:
        n *= int(s)
    idx = torch.arange(n, dtype=torch.float64)
    phase = (zlib.crc32(name.encode()) % 10007) * 0.001
    base = torch.sin(idx * 0.6180339887498949 * 7.0 + phase)
    if len(shape) >= 2:
        fan_in = n // int(shape[0])
        val = base * (gain * math.sqrt(2.0 / fan_in))
    elif name.endswith("weight"):
        val = 1.0 + 0.1 * base
    else:
        val = 0.1 * base
    return val.to(torch.float32).reshape(tuple(shape))


def det_state_dict(shapes: Mapping[str, Sequence[int]], skip=("T", "base_sde.T")) -> Dict[str, torch.Tensor]:
    """Fill every tensor named in ``shapes`` except the scalar horizons."""
    return {k: det_tensor(k, s) for k, s in shapes.items() if k not in skip}


def load_det_(module: torch.nn.Module, prefix_skip=("T", "base_sde.T")) -> Dict[str, torch.Tensor]:
    """In-place deterministic fill of an nn.Module; returns the new state."""
    sd = module.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            if k in prefix_skip or not v.dtype.is_floating_point:
                continue
            v.copy_(det_tensor(k, v.shape))
    return module.state_dict()
