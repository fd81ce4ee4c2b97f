"""pytest configuration: the ``gpu`` marker and shared fixture helpers."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


class Golden(dict):
    """npz -> dict of torch tensors; ``sub('p::')`` strips a key prefix."""

    def sub(self, prefix):
        return {k[len(prefix):]: v for k, v in self.items() if k.startswith(prefix)}


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    g = Golden()
    for k in z.files:
        a = z[k]
        g[k] = torch.from_numpy(a) if a.dtype.kind in "fiub" else a
    return g


def rel_l2(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def check_digest(g, tag, grads, prefix, tol):
    """Per-tensor gradient check against a golden digest (per-tensor L2 norm + first 8 entries, recorded from the
    reference).  Tensors whose true gradient is ~0 (conv / embedding biases feeding a one-channel-per-group GroupNorm)
    hold only rounding noise, so the slack is floored at 1e-3 of the largest tensor norm.  Prints and returns the
    worst measured deviation in units of the slack-free scale max(norm_i, floor)."""
    names = [str(s) for s in g[f"{tag}_gd_names"]]
    norms, heads = g[f"{tag}_gd_norms"], g[f"{tag}_gd_heads"]
    assert set(names) == {prefix + k for k in grads}
    floor = 1e-3 * float(norms.max())
    worst, worst_name = 0.0, ""
    for i, nm in enumerate(names):
        gr = grads[nm[len(prefix):]]
        scale = max(float(norms[i]), floor)
        h = gr.reshape(-1)[:8]
        ref = heads[i][: h.numel()]
        dev = max(abs(float(gr.double().norm()) - float(norms[i])), float((h.double() - ref.double()).abs().max())) / scale
        if dev > worst:
            worst, worst_name = dev, nm
    print(f"gradient digest {tag}: worst per-tensor deviation {worst:.2e} ({worst_name}), tolerance {tol:.1e}")
    assert worst <= tol, (worst_name, worst)
    return worst


def parity_vs_fp64(hip, oracle, what, floor_frac=1e-3, slack=2.0, slack_worst=4.0, abs_floor=1e-5):
    """The yardstick for 'within fp32 tolerance' on a deep net.  ``hip()`` -> (per_sample, {name: grad}) from the HIP path;
    ``oracle(dtype)`` -> the same from the CPU oracle evaluated in that dtype.  The float64 oracle is the truth, the
    float32 oracle is the reference's own arithmetic (same formulas, ATen fp32).  The HIP result must be no further from
    the truth than ``slack`` x the reference's own float32 evaluation — for the per-sample loss, the flat gradient and
    the worst single parameter tensor (per-tensor errors relative to max(|g_k|, floor_frac * max_k |g_k|): tensors under
    that floor have an analytically zero gradient and hold rounding noise only).  The worst-tensor bound is
    ``slack_worst`` x the reference's worst tensor: it is a maximum over ~280 tensors of two independent fp32 roundings
    of ill-conditioned sums, which differ by factors of a few between any two correct implementations (measured in
    round 2: 0.5x at 64x64x3, 0.7x for the 1-D U-Net, 2.5x at 32x32 — the same 2.5-3x with the composed round-1
    attention, with __expf and with expf).  Prints everything it measured."""
    per, grads = hip()
    per32, g32 = oracle(torch.float32)
    per64, g64 = oracle(torch.float64)
    assert set(grads) == set(g64), set(grads) ^ set(g64)
    keys = list(g64)
    cat = lambda g: torch.cat([g[k].reshape(-1).double().cpu() for k in keys])
    e_per, r_per = rel_l2(per.cpu(), per64), rel_l2(per32, per64)
    e_flat, r_flat = rel_l2(cat(grads), cat(g64)), rel_l2(cat(g32), cat(g64))
    top = max(float(g64[k].norm()) for k in keys)
    pt = lambda g, k: float((g[k].double().cpu() - g64[k]).norm()) / max(float(g64[k].norm()), floor_frac * top)
    e_t = {k: pt(grads, k) for k in keys}
    r_t = {k: pt(g32, k) for k in keys}
    kw, kr = max(e_t, key=e_t.get), max(r_t, key=r_t.get)
    print(f"{what} (errors vs the float64 oracle; HIP | reference arithmetic = float32 oracle):\n"
          f"  per-sample SSM loss rel-L2   {e_per:.2e} | {r_per:.2e}   (HIP vs fp32 oracle directly: {rel_l2(per.cpu(), per32):.2e})\n"
          f"  all gradients, flat rel-L2   {e_flat:.2e} | {r_flat:.2e}\n"
          f"  worst parameter tensor       {e_t[kw]:.2e} ({kw}) | {r_t[kr]:.2e} ({kr})")
    assert e_per <= max(slack * r_per, abs_floor)
    assert e_flat <= max(slack * r_flat, abs_floor)
    assert e_t[kw] <= max(slack_worst * r_t[kr], abs_floor), kw
    return e_per, e_flat


def within(measured, tol, what):
    """assert measured <= tol, printing the measured value (tolerances are set from these prints: <= 2x measured)."""
    print(f"{what}: measured {measured:.2e} (tolerance {tol:.1e})")
    assert measured <= tol, (what, measured, tol)
    return measured
