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


# ---------------------------------------------------------------------------------------------------------------------
# A WELL-CONDITIONED closed-form fill: the statistics of the reference's own default initialisation (PyTorch's
# kaiming-uniform conv / linear weights and biases, bound 1/sqrt(fan_in); GroupNorm gains ~1, shifts ~0) with the layers the
# reference zero-initialises (model/nn_utils.py:151-157: every ResBlock's second conv, every attention proj_out, the output
# conv) re-randomised small (std 0.02) so that a fresh U-Net is not identically zero.  Values come from a counter-based
# integer hash, not from a sinusoid: no structure, so the net amplifies rounding like a freshly initialised net does
# (det_tensor's sinusoidal fill makes the 2-D U-Net amplify one ulp by ~1e3, which is what the float64-yardstick tests
# exist for).  Used by the absolute-tolerance parity fixtures (tests/golden/g17_*.npz).
_ZERO_INIT_MARKS = (".out_layers.3.", ".proj_out.", "core.out.2.", "out.2.")


def _hash_uniform(n: int, seed: int) -> torch.Tensor:
    """n float64 values in [0, 1): murmur3's 32-bit finaliser of (index * golden + seed); int64 arithmetic, masked."""
    m = 0xFFFFFFFF
    x = (torch.arange(n, dtype=torch.int64) * 0x9E3779B1 + (seed & m)) & m
    x = x ^ (x >> 16)
    x = (x * 0x85EBCA6B) & m
    x = x ^ (x >> 13)
    x = (x * 0xC2B2AE35) & m
    x = x ^ (x >> 16)
    return x.to(torch.float64) / 4294967296.0


def init_like_tensor(name: str, shape: Sequence[int], shapes: Mapping[str, Sequence[int]]) -> torch.Tensor:
    n = 1
    for s in shape:
        n *= int(s)
    u = 2.0 * _hash_uniform(n, zlib.crc32(name.encode())) - 1.0              # (-1, 1)
    zero_init = any(mk in "." + name for mk in _ZERO_INIT_MARKS) or name.startswith("out.2.")
    if len(shape) >= 2:
        fan_in = n // int(shape[0])
        bound = 0.02 * math.sqrt(3.0) if zero_init else 1.0 / math.sqrt(fan_in)
        val = u * bound
    else:
        sib = name[: -len("bias")] + "weight" if name.endswith("bias") else None
        wshape = shapes.get(sib) if sib is not None else None
        if name.endswith("weight"):                                            # GroupNorm gain
            val = 1.0 + 0.05 * u
        elif wshape is not None and len(wshape) >= 2:                          # conv / linear bias
            fi = 1
            for s in wshape[1:]:
                fi *= int(s)
            val = u * (0.02 * math.sqrt(3.0) if zero_init else 1.0 / math.sqrt(fi))
        else:                                                                  # GroupNorm shift
            val = 0.05 * u
    return val.to(torch.float32).reshape(tuple(shape))


def init_like_state_dict(shapes: Mapping[str, Sequence[int]], skip=("T", "base_sde.T")) -> Dict[str, torch.Tensor]:
    return {k: init_like_tensor(k, s, shapes) for k, s in shapes.items() if k not in skip}


def load_init_like_(module: torch.nn.Module, prefix_skip=("T", "base_sde.T")) -> Dict[str, torch.Tensor]:
    """In-place well-conditioned fill of an nn.Module (see above); returns the new state."""
    sd = module.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    with torch.no_grad():
        for k, v in sd.items():
            if k in prefix_skip or not v.dtype.is_floating_point:
                continue
            v.copy_(init_like_tensor(k, v.shape, shapes))
    return module.state_dict()
