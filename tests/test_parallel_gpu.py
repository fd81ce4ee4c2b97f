"""-m gpu: the product's N>1 path (MLPScoreTrainer / UNetScoreTrainer with world=2) rehearsed with two ranks sharing
the one GPU of the test box over gloo (RCCL needs one GPU per rank; the collective call site is the same
`parallel.allreduce_sum_`).  Checks what data parallelism must guarantee: after every step both ranks hold identical
parameters (same all-reduced bucket, same Adam state), the loss is the global mean, and training moves them."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, kind, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), MSGM_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from sdeflow_light_amd import parallel
    from sdeflow_light_amd.SDEs import SGMsde, PluginReverseSDE
    from sdeflow_light_amd.train import MLPScoreTrainer, UNetScoreTrainer
    r, local, w = parallel.init_distributed()
    dev = parallel.local_device(local)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)                                   # same initial parameters on both ranks
    T = torch.nn.Parameter(torch.FloatTensor([1.0]), requires_grad=False)
    if kind == "mlp":
        from sdeflow_light_amd.NN import MLP
        net, d, B = MLP(2).to(dev), 2, 1024
    else:
        from sdeflow_light_amd.NNUnet1D import UNet1D
        net, d, B = UNet1D(input_dim=128, base_channels=16, channel_mults=(1, 2), emb_dim=32).to(dev), 128, 8
    gen = PluginReverseSDE(SGMsde(T=T, num_steps_forward=16, device=dev), net, T, deviceReverseSDE=dev).to(dev)
    flat, _ = net.flat_parameters()
    parallel.broadcast_(flat, 0)
    p0 = flat.clone()
    tr = (MLPScoreTrainer(gen, B, lr=1e-3, world=w, seed=1 + r) if kind == "mlp"
          else UNetScoreTrainer(gen, B, d, lr=1e-3, world=w, seed=1 + r))
    torch.manual_seed(100 + r)                             # different data shard per rank
    tr.set_data(torch.randn(B, d, device=dev))
    losses = [float(tr.step()) for _ in range(3)]
    flat, _ = net.flat_parameters()
    q.put((r, losses, flat.detach().cpu().numpy().tobytes(), float((flat - p0).abs().max())))   # bytes: no shared-memory handle to outlive
    parallel.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("kind", ["mlp", "unet1d"])
def test_two_rank_training_keeps_ranks_identical(kind):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, kind, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=500) for _ in range(world)], key=lambda t: t[0])
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    (_, l0, f0, m0), (_, l1, f1, m1) = res
    assert all(abs(a - b) <= 1e-6 * max(1.0, abs(a)) for a, b in zip(l0, l1)), (l0, l1)     # the all-reduced global mean
    assert f0 == f1 and len(f0) > 0                                                          # bitwise identical replicas
    assert m0 > 1e-5 and all(map(lambda v: v == v, l0))


def _shard_worker(rank, world, port, q):
    """Rank r of a 2-rank run with shard placement (same seed everywhere + first global row of the shard): the ranks
    must draw / compute exactly what a single-rank run does for the same global rows."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), MSGM_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from sdeflow_light_amd import parallel, sde_scheme as SS
    from sdeflow_light_amd.NN import MLP
    from sdeflow_light_amd.SDEs import SGMsde, PluginReverseSDE
    from sdeflow_light_amd.train import MLPScoreTrainer
    r, local, w = parallel.init_distributed()
    dev = parallel.local_device(local)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    T = torch.nn.Parameter(torch.FloatTensor([1.0]), requires_grad=False)
    gen = PluginReverseSDE(SGMsde(T=T, num_steps_forward=16, device=dev), MLP(2).to(dev), T, deviceReverseSDE=dev).to(dev)
    rows, d, N = 256, 2, 6
    b, e = parallel.shard_rows(rows, r, w)
    gen.base_sde.set_shard(b, d)
    x0 = gen.latent_sample(e - b, d)                                      # this rank's rows of the global latent draw
    xs = SS.euler_maruyama_sampler(gen, x0, num_steps=N, keep_all_samples=False).to(dev)      # Philox dW, sharded
    full = parallel.gather_rows(xs, rows)
    # training with shard placement: same seed, row_base = first global row
    torch.manual_seed(5)
    data = torch.randn(rows, d, device=dev)
    tr = MLPScoreTrainer(gen, e - b, lr=1e-3, world=w, seed=9, row_base=b, use_graph=False)
    tr.set_data(data[b:e])
    losses = [float(tr.step()) for _ in range(3)]
    flat, _ = gen.a.flat_parameters()
    q.put((r, full.cpu().numpy().tobytes(), losses, flat.detach().cpu().numpy().tobytes()))
    parallel.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_run_draws_and_computes_what_the_single_gpu_run_does():
    """the base SDE's Philox stream had no rank / row term, so every rank of a sharded sampler drew
    the same latent and dW.  With ``set_shard`` / ``row_base`` the 2-rank run must EQUAL the 1-rank run row for row
    (sampler: bit for bit; trainer: to the re-association of the gradient sum across ranks)."""
    import numpy as np
    from sdeflow_light_amd import sde_scheme as SS
    from sdeflow_light_amd.NN import MLP
    from sdeflow_light_amd.SDEs import SGMsde, PluginReverseSDE
    from sdeflow_light_amd.train import MLPScoreTrainer
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=500) for _ in range(world)], key=lambda t: t[0])
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    dev = "cuda"
    torch.manual_seed(0)
    T = torch.nn.Parameter(torch.FloatTensor([1.0]), requires_grad=False)
    gen = PluginReverseSDE(SGMsde(T=T, num_steps_forward=16, device=dev), MLP(2).to(dev), T, deviceReverseSDE=dev).to(dev)
    rows, d, N = 256, 2, 6
    x0 = gen.latent_sample(rows, d)
    ref = SS.euler_maruyama_sampler(gen, x0, num_steps=N, keep_all_samples=False)
    for r, full, _, _ in res:
        got = torch.from_numpy(np.frombuffer(full, dtype=np.float32).reshape(rows, d).copy())
        assert torch.equal(got, ref), float((got - ref).abs().max())
    assert not torch.equal(ref[: rows // 2], ref[rows // 2:])                 # the two shards are different draws
    torch.manual_seed(5)
    data = torch.randn(rows, d, device=dev)
    tr = MLPScoreTrainer(gen, rows, lr=1e-3, world=1, seed=9, use_graph=False)
    tr.set_data(data)
    losses = [float(tr.step()) for _ in range(3)]
    flat = gen.a.flat_parameters()[0].detach().cpu()
    for r, _, l2, f2 in res:
        got = torch.from_numpy(np.frombuffer(f2, dtype=np.float32).copy())
        e = float((got - flat).norm() / flat.norm())
        print(f"rank {r}: 2-rank vs 1-rank parameters after 3 steps rel-L2 {e:.2e}; losses {l2} vs {losses}")
        assert e <= 1e-6
        assert all(abs(a - b) <= 1e-5 * max(1.0, abs(b)) for a, b in zip(l2, losses))


def _rccl_worker(port, forced, q):
    """One rank; with MSGM_FORCE_DIST the trainer goes through the multi-rank code path on a REAL RCCL communicator."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.pop("MSGM_DIST_BACKEND", None)
    if forced:
        os.environ["MSGM_FORCE_DIST"] = "1"
    import torch.distributed as dist
    from sdeflow_light_amd import parallel
    from sdeflow_light_amd.SDEs import SGMsde, PluginReverseSDE
    from sdeflow_light_amd.train import UNetScoreTrainer
    from sdeflow_light_amd.NNUnet import VorticityUNet
    from oracle.det_params import load_det_
    parallel.init_distributed()
    assert dist.is_initialized() == bool(forced)
    if forced:
        assert dist.get_backend() == "nccl"
    dev = torch.device("cuda", 0)
    net = VorticityUNet(base_channels=32, channel_mults=(1, 2), num_res_blocks=1, in_space=16, attention_resolutions=(2,),
                        flatten_order="F")
    load_det_(net)
    net = net.to(dev)
    T = torch.nn.Parameter(torch.FloatTensor([1.0]), requires_grad=False)
    gen = PluginReverseSDE(SGMsde(T=T, num_steps_forward=16, device=dev), net, T, deviceReverseSDE=dev).to(dev)
    tr = UNetScoreTrainer(gen, 8, 256, lr=1e-3, world=1, seed=3)
    torch.manual_seed(5)
    tr.set_data(torch.randn(8, 256, device=dev))
    losses = [float(tr.step()) for _ in range(4)]
    kinds = ops_kinds(tr)
    flat, _ = net.flat_parameters()
    q.put((losses, flat.detach().cpu().numpy().tobytes(), kinds, parallel.multi(1)))
    if forced:
        parallel.barrier()
        dist.destroy_process_group()


def ops_kinds(tr):
    from sdeflow_light_amd import ops
    return ops.graph_node_kinds(tr.graph)


@pytest.mark.timeout(600)
def test_multi_rank_code_path_on_a_real_rccl_communicator():
    """RCCL needs one GPU per rank, so the 2-rank tests above run over gloo.  Here ONE rank is sent through the same
    multi-rank code path (MSGM_FORCE_DIST=1: process group with device_id, hipGraph captured in thread_local mode while
    the backend's watchdog thread is alive, graph replay -> RCCL all-reduce of the [gradients | loss] bucket -> Adam
    outside the graph) with backend "nccl" = RCCL, and must reproduce the plain one-rank run (whole step inside the
    graph) bit for bit: a sum over one rank and x 1/1 change nothing."""
    ctx = mp.get_context("spawn")
    res = {}
    for forced in (False, True):
        q = ctx.Queue()
        p = ctx.Process(target=_rccl_worker, args=(_free_port(), forced, q))
        p.start()
        res[forced] = q.get(timeout=540)
        p.join(60)
        assert p.exitcode == 0
    (l0, b0, k0, m0), (l1, b1, k1, m1) = res[False], res[True]
    assert (m0, m1) == (False, True)
    assert l0 == l1 and b0 == b1
    # the forced run's graph ends before the collective: it lacks the Adam / Philox-advance nodes of the one-rank graph
    assert set(k0) == set(k1) == {"kernel"} and k1["kernel"] < k0["kernel"]
    print(f"one rank through RCCL == plain run bit for bit; graph nodes {k0['kernel']} (whole step) vs {k1['kernel']} (up to the collective)")


def _resume_worker(rank, world, port, path, q):
    """a data-parallel resume.  Both ranks train 2 steps with shard placement, rank 0 writes ONE
    checkpoint, BOTH ranks load it into fresh trainers (each with its own row_base) and train 2 more steps.  The result
    must equal the uninterrupted 4-step run bit for bit — which it cannot if the loader takes rank 0's shard base."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), MSGM_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from sdeflow_light_amd import parallel
    from sdeflow_light_amd.NN import MLP, save_checkpoint, load_checkpoint
    from sdeflow_light_amd.SDEs import SGMsde, PluginReverseSDE
    from sdeflow_light_amd.train import MLPScoreTrainer
    r, local, w = parallel.init_distributed()
    dev = parallel.local_device(local)
    torch.cuda.set_device(dev)
    rows, d = 512, 2
    b, e = parallel.shard_rows(rows, r, w)
    torch.manual_seed(5)
    data = torch.randn(rows, d, device=dev)

    def fresh():
        torch.manual_seed(0)
        T = torch.nn.Parameter(torch.FloatTensor([1.0]), requires_grad=False)
        gen = PluginReverseSDE(SGMsde(T=T, num_steps_forward=16, device=dev), MLP(2).to(dev), T, deviceReverseSDE=dev).to(dev)
        tr = MLPScoreTrainer(gen, e - b, lr=1e-3, world=w, seed=9, row_base=b, use_graph=False)
        tr.set_data(data[b:e])
        return gen, tr
    gen, tr = fresh()                                     # uninterrupted: 4 steps
    for _ in range(4):
        tr.step()
    want = gen.a.flat_parameters()[0].detach().cpu().numpy().tobytes()
    gen, tr = fresh()                                     # interrupted after 2
    for _ in range(2):
        tr.step()
    if r == 0:
        save_checkpoint(path, gen, tr, 2)
    parallel.barrier()
    gen2, tr2 = fresh()
    it = load_checkpoint(path, gen2, tr2, dev)
    base = [int(v) for v in tr2.rng.state.tolist()]
    for _ in range(2):
        tr2.step()
    got = gen2.a.flat_parameters()[0].detach().cpu().numpy().tobytes()
    q.put((r, it, base, got == want, b))
    parallel.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_resume_from_rank0_checkpoint_equals_the_uninterrupted_run(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    path = str(tmp_path / "dp_resume.pt")
    ps = [ctx.Process(target=_resume_worker, args=(r, world, port, path, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=500) for _ in range(world)], key=lambda t: t[0])
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    for r, it, base, same, b in res:
        assert it == 2
        assert base[2] == b and base[3] == b * 2, (r, base)          # the loader kept ITS shard base, not rank 0's
        assert same, f"rank {r}: resumed run differs from the uninterrupted one"
