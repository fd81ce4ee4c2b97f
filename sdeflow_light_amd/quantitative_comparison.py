"""RBF-kernel MMD on the device — mirror of the reference's quantitative_comparison.py:22-46
(`compute_kernel`, `compute_mmd`; same names, arguments and return values).  The reference expands both
sample sets to an (Nx, Ny, d) tensor on the host; here one HIP kernel walks 64 x 64 tiles of pairs and, for the
MMD, only three scalars leave the chip.  No CPU fallback: inputs are moved to the HIP device."""
from __future__ import annotations

import torch

from . import ops
from ._lib import MsgmError


def _dev(x: torch.Tensor, like: torch.Tensor = None) -> torch.Tensor:
    if not x.is_cuda:
        if not torch.cuda.is_available():
            raise MsgmError("compute_mmd needs a HIP device; there is no CPU fallback")
        x = x.to(like.device if (like is not None and like.is_cuda) else "cuda")
    return x.detach().float().contiguous()


@torch.no_grad()
def compute_kernel(x, y):
    """(x_size, y_size) matrix exp(-mean((x_i - y_j)^2) / dim) — quantitative_comparison.py:22-36."""
    x = _dev(x, y)
    y = _dev(y, x)
    K, _ = ops.rbf_kernel(x, y, want_matrix=True, want_sum=False)
    return K


@torch.no_grad()
def compute_mmd(x, y):
    """mean(Kxx) + mean(Kyy) - 2 mean(Kxy) (quantitative_comparison.py:38-46) as a 0-dim tensor; the three kernel
    matrices are never materialised.  Means are formed in float64 from the float64 sums."""
    x = _dev(x, y)
    y = _dev(y, x)
    nx, ny = x.shape[0], y.shape[0]
    sxx = ops.rbf_kernel(x, x)[1]
    syy = ops.rbf_kernel(y, y)[1]
    sxy = ops.rbf_kernel(x, y)[1]
    return (sxx / (nx * nx) + syy / (ny * ny) - 2.0 * sxy / (nx * ny)).reshape(()).float()
