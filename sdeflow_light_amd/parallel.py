"""Data parallelism for the hot path: one process per GPU, batches sharded by
rank, ONE flat fp32 gradient bucket all-reduced (sum) over RCCL/xGMI per train
step (backend "nccl" IS RCCL on ROCm); the 1/world scale is folded into the
fused Adam kernel.  The sampler shards rows across ranks with no collective in
the loop and one all_gather at the end.  The reference has no distributed code
 — this file adds the single collective the path needs.

The helpers are backend-agnostic so the N>1 logic is covered by world_size-2
gloo tests on CPU (tests/test_parallel_gloo.py).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def local_device(local: int) -> torch.device:
    """cuda:<local rank>, wrapped onto the visible devices (several ranks may share a GPU in rehearsals)."""
    return torch.device("cuda", local % max(torch.cuda.device_count(), 1))


def forced() -> bool:
    """MSGM_FORCE_DIST=1: take the multi-rank code path (process group, graph that ends before the collective, RCCL
    all-reduce of the gradient bucket) even with ONE rank — a rehearsal of the N > 1 path with the real backend on a
    one-GPU box.  The arithmetic is the one-rank run's (sum over one rank, x 1/1)."""
    return bool(os.environ.get("MSGM_FORCE_DIST"))


def multi(world: int) -> bool:
    """Does a run of this world size go through the collectives?"""
    return world > 1 or forced()


def _active() -> bool:
    return dist.is_initialized() and (dist.get_world_size() > 1 or forced())


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    rank, local, world = env_world()
    if multi(world) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # MSGM_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N>1 path on a 1-GPU box)
            backend = os.environ.get("MSGM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if torch.cuda.is_available():
            dev = local % torch.cuda.device_count()
            torch.cuda.set_device(dev)
            if backend == "nccl":
                # bind the communicator to this rank's GPU up front (eager RCCL init on the right device, no lazy
                # device guess at the first collective)
                kw["device_id"] = torch.device("cuda", dev)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def shard_rows(n_global: int, rank: int, world: int) -> Tuple[int, int]:
    """[begin, end) of this rank's rows; the remainder goes to the low ranks."""
    base, rem = divmod(n_global, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def allreduce_sum_(bucket: torch.Tensor, async_op: bool = False):
    """In-place sum all-reduce of the flat gradient bucket (0.135 / 3.3 / 16 MB:
    latency-bound over xGMI, so ONE collective per step, never per tensor)."""
    if _active():
        return dist.all_reduce(bucket, op=dist.ReduceOp.SUM, async_op=async_op)
    return None


def broadcast_(bucket: torch.Tensor, src: int = 0):
    if _active():
        dist.broadcast(bucket, src=src)


def gather_rows(local: torch.Tensor, n_global: int) -> torch.Tensor:
    """Concatenate per-rank row shards (sizes from ``shard_rows``) on every rank."""
    if not _active():
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_rows(n_global, r, world) for r in range(world)]
    mx = max(e - b for b, e in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]].copy_(local)
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[: e - b] for o, (b, e) in zip(outs, sizes)], dim=0)


def max_over_ranks(value: float, device) -> float:
    if not _active():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if _active():
        dist.barrier()
