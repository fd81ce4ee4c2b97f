"""Diagnostic: the C5 leg of bench.py under 2 ranks sharing one GPU (gloo), with finiteness checks at every stage."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sdeflow_light_amd import parallel
rank, local, world = parallel.init_distributed()
dev = parallel.local_device(local)
torch.cuda.set_device(dev)
gen, d = bench.build_unet("c5", dev)
flat, _ = gen.a.flat_parameters()
parallel.broadcast_(flat, 0)
print(rank, "params finite", bool(torch.isfinite(flat).all()), flush=True)
from sdeflow_light_amd.sde_scheme import GraphedStepSampler
rows, chunk, N = 2048, 1024, 2
if os.environ.get("DBG_SHARD", "1") == "1":
    gen.base_sde.set_shard(rank * rows, d)
print(rank, "rng", gen.base_sde.philox(dev).state.tolist(), flush=True)
gs = GraphedStepSampler(gen, chunk, d, N)
x = gen.latent_sample(rows, d)
print(rank, "latent finite", bool(torch.isfinite(x).all()), float(x.abs().max()), float(x.std()), flush=True)
for c0 in range(0, rows, chunk):
    out = gs.run(x[c0:c0 + chunk])
    print(rank, c0, "chunk finite", bool(torch.isfinite(out).all()), float(out.abs().max()), flush=True)
parallel.barrier()
