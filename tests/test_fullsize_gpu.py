"""BASELINE full sizes of the U-Net configurations (C3, C4 shard, C5 chunk) on the GPU: size-independent PROPERTIES the
path has.  (The oracle comparisons at these network shapes — 64x64x3 with T = 1024 attention, UNet1D
L = 1024 — are in tests/test_round2_gpu.py at batch 2, where the float64 oracle takes seconds; here the batch is the
config's own, which only properties can cover.)
Rows never interact (GroupNorm is per sample, attention is per sample), so
  (i)  the per-sample losses of a batch are those of its two halves, and the mean gradient of the batch is the average
       of the two half-batch gradients — exactly the data-parallel equivalence the multi-GPU path relies on;
  (ii) the score of a row does not depend on which other rows share the launch.
Per-row values are compared BIT FOR BIT: no kernel on the path lets the batch size choose the order of a row's
arithmetic (the GroupNorm statistics are reduced over per-sample chunks of a fixed size and combined in double, in a
fixed order).  Gradients: 1e-4 rel-L2 (the slot-ordered sums group the batch differently for different batch sizes).
Small-size parity against the oracle and the golden vectors of the same code paths: test_host_gpu.py (1-D),
test_unet2d_gpu.py (2-D)."""
import pytest
import torch

from conftest import rel_l2
from test_host_gpu import make_gen, _unet1d
from test_unet2d_gpu import _vunet

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ssm(gen, x, u, eps, uv, sl):
    gen.zero_grad()
    per = gen.ssm(x[sl].contiguous().to(DEV), u=u[sl].contiguous().to(DEV), eps=eps[sl].contiguous().to(DEV),
                  u_v=uv[sl].contiguous().to(DEV))
    per.mean().backward()
    g = torch.cat([p.grad.reshape(-1) for _, p in gen.a.named_parameters()]).cpu()
    return per.detach().cpu(), g


def _shard_equivalence(gen, B, d, scale):
    torch.manual_seed(11)
    x, u, eps, uv = torch.randn(B, d) * scale, torch.rand(B), torch.randn(B, d), torch.rand(B, d)
    pf, gf = _ssm(gen, x, u, eps, uv, slice(0, B))
    p0, g0 = _ssm(gen, x, u, eps, uv, slice(0, B // 2))
    p1, g1 = _ssm(gen, x, u, eps, uv, slice(B // 2, B))
    assert torch.isfinite(pf).all() and torch.isfinite(gf).all() and float(gf.abs().max()) > 0
    assert torch.equal(torch.cat([p0, p1]), pf), rel_l2(torch.cat([p0, p1]), pf)
    assert rel_l2(0.5 * (g0 + g1), gf) <= 1e-4, rel_l2(0.5 * (g0 + g1), gf)
    # and the halves really are different problems
    assert rel_l2(g0, g1) > 1e-3


def test_c3_full_size_shard_equivalence():
    """C3: UNet1D L = 1024, B = 4096 (the per-GPU batch of BASELINE configs[2])."""
    gen = make_gen("sgm", _unet1d(1024))
    _shard_equivalence(gen, 4096, 1024, 1.0)


def test_c4_shard_size_equivalence():
    """C4: VorticityUNet 64x64x3 (d = 12 288) at the per-GPU shard of the global batch 256 over 8 GPUs (B = 32)."""
    gen = make_gen("sgm", _vunet(64, "F", channels=3))
    _shard_equivalence(gen, 32, 3 * 64 * 64, 1.0)


def test_c5_chunk_rows_are_independent():
    """C5: one score evaluation of the sampler at its per-GPU set (1024 rows of d = 12 288): a row's score is the same
    in the full launch, in a half-size launch, and when every other row of the launch is replaced."""
    net = _vunet(64, "F", channels=3)
    torch.manual_seed(5)
    B, d = 1024, 3 * 64 * 64
    x = torch.randn(B, d, device=DEV)
    s = torch.rand(B, device=DEV) * 0.9 + 0.05
    full = net(x, s)
    assert full.shape == (B, d) and torch.isfinite(full).all()
    h0, h1 = net(x[: B // 2].contiguous(), s[: B // 2].contiguous()), net(x[B // 2:].contiguous(), s[B // 2:].contiguous())
    assert torch.equal(torch.cat([h0, h1]), full)
    x2 = x.clone()
    x2[1:] = torch.randn(B - 1, d, device=DEV) * 2.0
    other = net(x2, s)
    assert torch.equal(other[:1], full[:1])
    assert rel_l2(other[1:].cpu(), full[1:].cpu()) > 1e-2


def test_c4_full_batch_256_matches_its_32_row_shards():
    """VERDICT r2 #5d: the N = 1 bench shape itself — C4 at B = 256 in ONE launch — against its eight 32-row shards (the
    8-GPU partition): per-sample losses bit for bit, the mean gradient = the average of the shard gradients (1e-4)."""
    gen = make_gen("sgm", _vunet(64, "F", channels=3))
    B, d = 256, 3 * 64 * 64
    torch.manual_seed(12)
    x, u, eps, uv = torch.randn(B, d), torch.rand(B), torch.randn(B, d), torch.rand(B, d)
    pf, gf = _ssm(gen, x, u, eps, uv, slice(0, B))
    assert torch.isfinite(pf).all() and torch.isfinite(gf).all() and float(gf.abs().max()) > 0
    ps, gs = [], torch.zeros_like(gf)
    for r in range(8):
        p, g = _ssm(gen, x, u, eps, uv, slice(32 * r, 32 * (r + 1)))
        ps.append(p)
        gs += g / 8
    assert torch.equal(torch.cat(ps), pf)
    e = rel_l2(gs, gf)
    print(f"C4 B=256: mean of the eight 32-row shard gradients vs the full-batch gradient: rel-L2 {e:.2e}")
    assert e <= 1e-4


def test_c5_4096_row_chunk_rows_are_independent():
    """VERDICT r2 #5d: the sampler's N = 1 chunk size (4096 rows of d = 12 288 per launch, bench.py CHUNK): a row's score
    is the same bits in the 4096-row launch and in the 1024-row launch an 8-GPU rank runs."""
    net = _vunet(64, "F", channels=3)
    torch.manual_seed(6)
    B, d = 4096, 3 * 64 * 64
    x = torch.randn(B, d, device=DEV)
    s = torch.rand(B, device=DEV) * 0.9 + 0.05
    full = net(x, s)
    assert full.shape == (B, d) and torch.isfinite(full).all()
    for c0 in (0, 3072):
        part = net(x[c0:c0 + 1024].contiguous(), s[c0:c0 + 1024].contiguous())
        assert torch.equal(part, full[c0:c0 + 1024])
    del full
    torch.cuda.empty_cache()
