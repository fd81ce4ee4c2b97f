"""MLP score network on the fused HIP kernels — host mirror of the
reference's ``NN.py`` (same class names, constructor arguments and
``state_dict`` keys: ``main.{0,2,4,6}.{weight,bias}``, NN.py:73-106).

``forward(x, t)`` launches ONE kernel (msgm_mlp_forward: four layers chained
through MFMA accumulators).  The module is inference-only under autograd:
training goes through ``PluginReverseSDE.ssm`` which calls the fused
forward-mode SSM kernel (msgm_mlp_ssm_grad) and fills ``.grad`` directly.
"""
from __future__ import annotations

import random

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import MsgmError


class Swish(nn.Module):
    """sigmoid(x)*x (NN.py:48-53).  Kept for constructor compatibility; the
    fused kernel applies it (and its first two derivatives) in registers."""

    def forward(self, x):
        raise MsgmError("Swish is fused into the MLP kernel; it is not a standalone op in this build")


class NormalizeLogRadius(nn.Module):
    """x -> (x/(|x|+eps), log(|x|+eps)) (NN.py:56-70); fused into the first
    layer of the score-net kernels (eps is fixed to 1e-6 there)."""

    def __init__(self, eps: float = 1e-6):
        super().__init__()
        if eps != 1e-6:
            raise MsgmError("the fused kernels hard-code eps = 1e-6 (the reference default)")
        self.eps = eps


class FlatParamMixin:
    """Re-homes all parameters of a module as views into ONE flat fp32 bucket
    (and a matching flat gradient bucket): the fused Adam kernel and the RCCL
    gradient all-reduce then touch a single contiguous tensor.  state_dict
    keys and shapes are unchanged."""

    def _flatten_parameters(self):
        params = [p for p in self.parameters()]
        n = sum(p.numel() for p in params)
        dev = params[0].device
        flat = torch.empty(n, dtype=torch.float32, device=dev)
        # the gradient bucket carries ONE spare slot (the mean loss): [grads | loss] is then the buffer the RCCL
        # all-reduce works on in place — no staging copy of the 16 MB bucket per step
        self._gbucket = torch.zeros(n + 1, dtype=torch.float32, device=dev)
        gflat = self._gbucket[:n]
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                flat[off:off + k].copy_(p.data.reshape(-1))
                p.data = flat[off:off + k].view(p.shape)
                p.grad = gflat[off:off + k].view(p.shape)
                off += k
        self._flat, self._gflat = flat, gflat
        return flat, gflat

    def flat_parameters(self):
        """(params, grads) flat buckets, (re)built if the module moved device."""
        first = next(self.parameters())
        if getattr(self, "_flat", None) is None or self._flat.device != first.device or \
                first.data_ptr() != self._flat.data_ptr():
            self._flatten_parameters()
        return self._flat, self._gflat

    def grad_bucket(self):
        """[flat gradients | one spare slot] — the collective buffer of the data-parallel trainers."""
        self.flat_parameters()
        return self._gbucket


class MLP(nn.Module, FlatParamMixin):
    """Same signature as the reference (NN.py:73-80).  ``hidden_dim`` must be
    128 and ``act`` Swish — the only configuration the driver ever builds
    (MSGM_higherDim.py:702) and the one the kernel is specialised for."""

    def __init__(self, input_dim=2, index_dim=1, hidden_dim=128, act=None, premodule=None):
        super().__init__()
        if hidden_dim != 128 or index_dim != 1:
            raise MsgmError("fused MLP kernel supports hidden_dim=128, index_dim=1")
        if act is not None and not isinstance(act, Swish):
            raise MsgmError("fused MLP kernel implements Swish only")
        assert premodule is None or premodule in ["NormalizeLogRadius"]
        if input_dim > 30:
            raise MsgmError("fused MLP kernel supports input_dim <= 30")
        self.input_dim, self.index_dim, self.hidden_dim = input_dim, index_dim, hidden_dim
        self.output_dim = input_dim
        self.premodule = premodule
        self.pre = NormalizeLogRadius() if premodule == "NormalizeLogRadius" else None
        self.learnable_network_input_dim = input_dim + (1 if self.pre is not None else 0)
        act = Swish()
        self.act = act
        # identical container layout -> identical state_dict keys (main.0 / .2 / .4 / .6)
        self.main = nn.Sequential(
            nn.Linear(self.learnable_network_input_dim + index_dim, hidden_dim), act,
            nn.Linear(hidden_dim, hidden_dim), act,
            nn.Linear(hidden_dim, hidden_dim), act,
            nn.Linear(hidden_dim, self.output_dim),
        )
        self._flat = None

    def kernel_params(self):
        """msgm_mlp_params_t for the current parameter storage."""
        self.flat_parameters()
        m = self.main
        return ops.mlp_params(m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[4].weight, m[4].bias,
                              m[6].weight, m[6].bias, premodule=self.pre is not None)

    @torch.no_grad()
    def forward(self, input, t):
        sz = input.size()
        x = input.reshape(-1, self.input_dim).contiguous().float()
        t = t.reshape(-1).contiguous().float()
        if t.numel() == 1 and x.shape[0] != 1:
            t = t.expand(x.shape[0]).contiguous()
        return ops.mlp_forward(self.kernel_params(), x, t).view(*sz)


def _philox_owners(gen_sde, optim):
    """Every object on the training path that owns a device Philox stream: the base SDE (perturbation / probe / sampler
    draws of ``ssm`` and the integrators) and, for the graph-replayed trainers, the trainer itself."""
    owners = {}
    base = getattr(gen_sde, "base_sde", None)
    if base is not None and getattr(base, "rng", None) is not None:
        owners["base_sde"] = base.rng
    if getattr(optim, "rng", None) is not None:
        owners["trainer"] = optim.rng
    return owners


def save_checkpoint(path, gen_sde, optim, iteration):
    """Same dictionary layout as the reference (NN.py:13-22) plus ONE extra key the reference's loader ignores:
    ``msgm_hip`` = the device Philox state {seed, offset, row_base, elem_base} of every stream owner.  All training
    noise of this build comes from those streams (the torch / numpy / python generator states upstream saves are kept
    for wire compatibility), so a resumed run continues the same noise sequence instead of replaying iteration 0.
    ``optim`` is a ``torch.optim.Adam`` / ``FusedAdam`` or one of the graph-replayed trainers
    (``train.MLPScoreTrainer`` / ``UNetScoreTrainer``: Adam state in the same layout, device step counter included)."""
    torch.save({"iteration": iteration, "model": gen_sde.state_dict(), "optimizer": optim.state_dict(),
                "torch_rng": torch.get_rng_state().cpu(), "numpy_rng": np.random.get_state(),
                "python_rng": random.getstate(),
                "msgm_hip": {"philox": {k: r.state_dict() for k, r in _philox_owners(gen_sde, optim).items()}}}, path)


def load_checkpoint(path, gen_sde, optim, device):
    """Counterpart of NN.py:24-42.  Only load checkpoints you wrote yourself:
    like upstream this unpickles optimizer / RNG objects.  Checkpoints written by the reference (no ``msgm_hip``
    key) load too; the Philox streams then keep their current state."""
    ck = torch.load(path, map_location=device, weights_only=False)
    gen_sde.load_state_dict(ck["model"])
    optim.load_state_dict(ck["optimizer"])
    rng = ck["torch_rng"]
    torch.set_rng_state((rng if rng.dtype == torch.uint8 else rng.to(torch.uint8)).cpu())
    np.random.set_state(ck["numpy_rng"])
    random.setstate(ck["python_rng"])
    saved = ck.get("msgm_hip", {}).get("philox", {})
    if "base_sde" in saved:
        gen_sde.base_sde.philox(device).load_state_dict(saved["base_sde"])
    if "trainer" in saved and getattr(optim, "rng", None) is not None:
        optim.rng.load_state_dict(saved["trainer"])
    return ck["iteration"]


def evaluate(gen_sde, x_test):
    """Mean and standard error of the ELBO slice estimate on a test batch (NN.py:123-129)."""
    gen_sde.eval()
    num_samples_ = x_test.size(0)
    test_elbo = gen_sde.elbo_random_t_slice(x_test)
    gen_sde.train()
    return test_elbo.mean(), test_elbo.std() / num_samples_ ** 0.5
