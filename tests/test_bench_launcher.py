"""CPU: `bench.py --gpus N` starts its own ranks through torch.distributed.run BEFORE any GPU call (the reference is
single-process, MSGM_higherDim.py:437-446: the launcher is the build's job).  Without a GPU every rank must stop with the
loud "needs a GPU" message — which proves the parent spawned N fresh children and relayed their exit status."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="the GPU-side launcher run is part of the -m gpu bench rehearsal")
@pytest.mark.timeout(300)
def test_bench_gpus2_spawns_ranks_and_fails_loudly_without_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MSGM_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU") >= 2, r.stderr[-2000:]      # one message per spawned rank


def test_bench_worker_refuses_mismatched_world(monkeypatch):
    """Under an existing torchrun environment the ranks run directly; a WORLD_SIZE that contradicts --gpus is an error."""
    sys.path.insert(0, ROOT)
    import bench
    assert callable(bench.launch_ranks) and callable(bench.worker)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "torch.distributed.run" in src and "launch_ranks(a))              # before ANY GPU call" in src
