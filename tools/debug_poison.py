#!/usr/bin/env python3
"""GPU box: does any kernel of the U-Net training pass READ workspace it did not write?  One SSM forward + backward of the 2-D
U-Net (32x32, B = 2 by default), then every cached workspace of the product (ops._SCRATCH buffers, the DeferredReduces slab
arena) is filled with NaN and the pass is repeated: the gradients must be the same bits.  (Found while chasing a
test-order-dependent 3e-6 in the Adam-loop test: stale finite values in a recycled workspace look like rounding noise.)
usage: python tools/debug_poison.py [S=32] [B=2]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from sdeflow_light_amd import ops
from sdeflow_light_amd.NNUnet import VorticityUNet
from sdeflow_light_amd.SDEs import SGMsde, PluginReverseSDE

S_ = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda")
torch.manual_seed(0)
net = VorticityUNet(base_channels=32, channel_mults=(1, 2, 4), num_res_blocks=2, in_space=S_, attention_resolutions=(2, 4),
                    flatten_order="F").to(dev)
with torch.no_grad():
    for prm in net.parameters():
        if float(prm.abs().sum()) == 0.0:
            prm.normal_(0.0, 0.02)
T = torch.nn.Parameter(torch.FloatTensor([1.0]), requires_grad=False)
gen = PluginReverseSDE(SGMsde(T=T, num_steps_forward=16, device=dev), net, T, deviceReverseSDE=dev).to(dev)
d = S_ * S_
x, u, eps, uv = torch.randn(B, d) * 3, torch.rand(B), torch.randn(B, d), torch.rand(B, d)


def run():
    gen.zero_grad()
    loss = gen.ssm(x.to(dev), u=u.to(dev), eps=eps.to(dev), u_v=uv.to(dev)).mean()
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {k: p.grad.detach().clone() for k, p in net.named_parameters()}


def poison(val):
    n = 0
    for ws in list(ops._SCRATCH.values()) + list(getattr(ops, "_SCRATCH_KEEP", [])):
        ws.fill_(val); n += 1
    for dr in ops.DeferredReduces._cache.values():
        for c in dr.chunks:
            c.fill_(val); n += 1
    torch.cuda.synchronize()
    return n


l0, g0 = run()
for val in (float("nan"), 1.0e3):
    n = poison(val)
    l1, g1 = run()
    bad = [k for k in g0 if not torch.equal(g0[k], g1[k])]
    print(f"poison {val}: {n} workspaces filled; loss {l0!r} -> {l1!r}; {len(bad)} of {len(g0)} gradient tensors changed")
    for k in bad[:12]:
        a, b = g0[k], g1[k]
        print(f"   {k:60s} {tuple(a.shape)}  nan {int(torch.isnan(b).sum())}  max|diff| {float((a - b).abs().nan_to_num(0).max()):.3e}")
