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
