"""world_size-2 gloo tests (CPU) of the N>1 logic: sharding + one flat-bucket
sum all-reduce reproduces the full-batch gradient; row gather reassembles the
sampler output.  Per-rank gradients come from the CPU oracle (the HIP kernel
itself is checked against the same oracle in the -m gpu tests)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sde_ref as S, nets_ref as N, ssm_ref as LR
from oracle.det_params import det_state_dict
from tests_util import mlp_shapes


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _flat(g):
    return torch.cat([g[k].reshape(-1) for k in sorted(g)])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from sdeflow_light_amd import parallel
    torch.set_num_threads(2)
    r, l, w = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    B, d = 257, 2                                       # ragged global batch
    sp = S.SdeSpec()
    p = det_state_dict(mlp_shapes(d))
    x, u, eps, uv = torch.randn(B, d), torch.rand(B, 1), torch.randn(B, d), torch.rand(B, d)
    t = S.clamp_time(sp, u); y = S.vp_perturb(sp, t, x, eps); v = S.rademacher_from_uniform(uv)
    score = lambda prm, yy, tt: N.mlp_forward(prm, yy, tt)
    b, e = parallel.shard_rows(B, rank, world)
    # local gradient of sum_b loss_b / B_global  (what msgm_mlp_ssm_grad computes with inv_batch = 1/B_global)
    _, per, g = LR.ssm_mean_and_grads(sp, score, p, t[b:e], y[b:e], v[b:e])
    bucket = _flat(g) * ((e - b) / B)
    parallel.allreduce_sum_(bucket)
    _, per_full, gfull = LR.ssm_mean_and_grads(sp, score, p, t, y, v)
    err = float((bucket - _flat(gfull)).norm() / _flat(gfull).norm())
    gathered = parallel.gather_rows(per.reshape(-1, 1), B)
    err2 = float((gathered.reshape(-1) - per_full).abs().max())
    mx = parallel.max_over_ranks(float(rank + 1), "cpu")
    # params broadcast from rank 0
    flat = torch.full((10,), float(rank))
    parallel.broadcast_(flat, 0)
    parallel.barrier()
    q.put((rank, err, err2, mx, float(flat.sum())))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ddp_bucket_allreduce_equals_full_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, err2, mx, fsum in res:
        assert err <= 1e-5, err
        assert err2 <= 1e-5, err2
        assert mx == 2.0 and fsum == 0.0
