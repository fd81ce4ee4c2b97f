#!/usr/bin/env python3
"""GPU box: the collectives of the data-parallel path (sdeflow_light_amd/parallel.py) on a ONE-rank RCCL communicator —
checks that backend "nccl" (= RCCL) initialises with device_id on this image / GPU and that the calls the trainers and
bench.py make (float32 bucket all-reduce, float64 MAX, barrier, all_gather, broadcast) run; no second GPU is needed."""
import os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
t0 = time.time()
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
b = torch.arange(4_000_000, dtype=torch.float32, device="cuda")
ref = b.clone()
dist.all_reduce(b, op=dist.ReduceOp.SUM)
m = torch.tensor([3.5], dtype=torch.float64, device="cuda")
dist.all_reduce(m, op=dist.ReduceOp.MAX)
dist.barrier()
outs = [torch.empty(5, 3, device="cuda")]
dist.all_gather(outs, torch.ones(5, 3, device="cuda"))
dist.broadcast(b, src=0)
torch.cuda.synchronize()
assert torch.equal(b, ref) and float(m) == 3.5 and float(outs[0].sum()) == 15.0
# the captured-step pattern: graph replay, then the collective on the same stream, then Adam-like use of the bucket
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    b.mul_(1.0)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    b.mul_(2.0)
for _ in range(3):
    g.replay()
    dist.all_reduce(b, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
assert torch.equal(b, ref * 8)
dist.destroy_process_group()
print(f"RCCL single-rank collectives ok ({time.time() - t0:.1f} s)")
