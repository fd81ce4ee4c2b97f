#!/usr/bin/env python3
"""Headline benchmark of the MI355X hot path (driver contract: see task brief).

Default line (N = --gpus ranks, one per GPU): BASELINE.json configs[3], "C4" — the config the north-star target is
quoted on: VorticityUNet 64x64x3 (4.02 M parameters; MSGM_higherDim.py:708-716), SGM, GLOBAL batch 256 split evenly
over the ranks, fp32.
  step   = one score-matching training step (MSGM_higherDim.py:803-809): perturb (K1) + Rademacher probe + dual-number
           forward + hand-written backward of the 2-D U-Net + [RCCL all-reduce of the flat gradient bucket when N>1] +
           fused Adam; inputs resident in HBM; everything before the collective replayed as ONE hipGraph.
  value  = global-batch-256 training steps per second ("strong" scaling: total work fixed as N grows).
  timing = W warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + device synchronise, MAX over
           ranks; blocks repeat until >= 1 s has been timed and the MEDIAN block is reported.
On the same JSON line:
  sample_steps_per_s  C5: reverse-SDE Euler-Maruyama steps/s for 8192 samples of the same net (rows split over the
                      ranks, chunks of <= 4096 rows, ONE hipGraph-captured step replayed --sample-steps times per chunk);
  roofline            the dominant kernel of the C4 step, timed live with per-launch HIP events on the launch stream
                      (+ the other heavy kernels and the whole-step fraction);
  cpu_baseline        the CPU oracle on this host's cores at a reduced batch (rank 0, N=1 only);
  extra               C2 (MLP, B=65536) and C3 (UNet1D L=1024, B=4096) short legs (N=1 only).
--workload c2|c3 selects those configs as the main line instead (c5 = sampling only).

`--gpus N` with N > 1 and no torchrun environment: this process touches no GPU and starts N ranks itself
(`python -m torch.distributed.run ... bench.py --gpus N ...`), relaying their output; under the driver's own
torch.distributed.run launch the ranks run directly.  MSGM_DIST_BACKEND=gloo lets several ranks share one GPU;
MSGM_FORCE_DIST=1 sends a ONE-rank run through the multi-rank code path (process group, graph up to the collective, RCCL
all-reduce of the gradient bucket): a rehearsal of the N > 1 path with the real backend on a one-GPU box.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_C2 = 65536
MLP_FWD_FLOP = 66816                 # SURVEY.md App. A.3 (2*MACs, d=2)
PEAK_F32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md: v_mfma_f32_* = fp32 vector rate
UNET_FWD_FLOP = {"c3": 0.4554e9, "c4": 5.974e9 + 9.4e6, "c5": 5.974e9 + 9.4e6}    # per sample, SURVEY.md App. A
GLOBAL_BATCH = {"c2": B_C2, "c3": 4096, "c4": 256}
SAMPLE_ROWS = 8192
CHUNK = 4096        # rows per sampler launch: a rank's share of the 8192 rows if smaller (8 ranks: 1024); measured
                    # 92.6 / 94.2 / 95.2 algorithmic TFLOP/s per EM step at 1024 / 2048 / 4096 rows (tools/time_unet2d.py)


# ----------------------------------------------------------------------------------------------------- launcher
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(a) -> int:
    """Parent of an N-rank run: NO GPU call happens in this process; the ranks are fresh children."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC (RCCL across processes)
    return subprocess.run(cmd, env=env).returncode


# ----------------------------------------------------------------------------------------------------- helpers
def host_cores() -> int:
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    return min(avail, 16)             # a 1-GPU box is given a 16-CPU share; more threads only oversubscribe


def timed_blocks(step_fn, steps, warmup, dev, min_seconds=1.0, max_blocks=50):
    """W warm-up steps, then blocks of exactly `steps` steps (barrier + synchronise on both sides, MAX over ranks)
    until >= min_seconds have been timed; returns (median block seconds, all block seconds)."""
    import torch
    from sdeflow_light_amd import parallel
    for _ in range(warmup):
        step_fn()
    blocks, total = [], 0.0
    while total < min_seconds and len(blocks) < max_blocks:
        parallel.barrier(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        torch.cuda.synchronize(dev); parallel.barrier()
        dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
        blocks.append(dt)
        total += dt
    return statistics.median(blocks), blocks


def time_kernel_events(fn, iters, dev):
    """Average duration of ONE launch of fn: every launch is bracketed by its own pair of HIP events on the launch
    stream (start-to-end of the kernel, as rocprofv3's per-kernel duration counts it — the cross-check committed under
    profiles/).  A loop of back-to-back launches timed as a whole reads ~5 % lower (the next launch's workgroups start
    while the previous one drains) and an eager loop ~3 % higher (host launch gaps); neither is the kernel's duration."""
    import torch
    for _ in range(3):
        fn()
    torch.cuda.synchronize(dev)
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for e0, e1 in pairs:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize(dev)
    return sum(e0.elapsed_time(e1) for e0, e1 in pairs) / iters * 1e-3


def free_gpu():
    import gc
    import torch
    from sdeflow_light_amd import ops
    gc.collect()                       # drops the finished leg's trainers / graphs (they hold the arena's addresses)
    ops.DeferredReduces.release()      # then the slab arena of the deterministic weight gradients
    gc.collect()
    torch.cuda.empty_cache()


# ----------------------------------------------------------------------------------------------------- models
def build_mlp(dev):
    import torch
    from sdeflow_light_amd.NN import MLP
    from sdeflow_light_amd.SDEs import SGMsde, PluginReverseSDE
    torch.manual_seed(0)                                   # model init seed as upstream (MSGM_higherDim.py:41)
    net = MLP(input_dim=2, index_dim=1, hidden_dim=128).to(dev)
    T = torch.nn.Parameter(torch.FloatTensor([1.0]), requires_grad=False)
    sde = SGMsde(beta_min=0.1, beta_max=20.0, t_epsilon=1e-3, T=T, num_steps_forward=16, device=dev)
    return PluginReverseSDE(sde, net, T, vtype="rademacher", deviceReverseSDE=dev).to(dev)


def build_unet(workload, dev):
    import torch
    from sdeflow_light_amd.SDEs import SGMsde, PluginReverseSDE
    torch.manual_seed(0)
    if workload == "c3":
        from sdeflow_light_amd.NNUnet1D import UNet1D
        net, d = UNet1D(input_dim=1024, base_channels=32, channel_mults=(1, 2, 4), num_res_blocks=2, emb_dim=128).to(dev), 1024
    else:
        from sdeflow_light_amd.NNUnet import VorticityUNet
        net = VorticityUNet(base_channels=32, channel_mults=(1, 2, 4), num_res_blocks=2, in_space=64,
                            attention_resolutions=(2, 4), flatten_order="F", channels=3).to(dev)
        d = 3 * 64 * 64
        # upstream zero-initialises every ResBlock's 2nd conv, attention proj_out and the final conv
        # (model/nn_utils.py:151-157); random-init them so no kernel multiplies by zeros
        with torch.no_grad():
            for p in net.parameters():
                if p.dim() > 1 and float(p.abs().sum()) == 0.0:
                    p.normal_(0, 0.02)
    T = torch.nn.Parameter(torch.FloatTensor([1.0]), requires_grad=False)
    sde = SGMsde(beta_min=0.1, beta_max=20.0, t_epsilon=1e-3, T=T, num_steps_forward=16, device=dev)
    return PluginReverseSDE(sde, net, T, vtype="rademacher", deviceReverseSDE=dev).to(dev), d


# ----------------------------------------------------------------------------------------------------- CPU baselines
def cpu_baseline_mlp(budget_s=8.0):
    """The CPU oracle (restatement of the reference, pinned by golden vectors) on the C2 workload."""
    import torch
    from oracle import sde_ref as S, nets_ref as N, ssm_ref as LR
    from sdeflow_light_amd.data import gaussian_mixture_2d
    torch.manual_seed(0)
    torch.set_num_threads(host_cores())
    sp = S.SdeSpec()
    p = {"main.0.weight": torch.randn(128, 3) * 0.5, "main.0.bias": torch.zeros(128),
         "main.2.weight": torch.randn(128, 128) * 0.09, "main.2.bias": torch.zeros(128),
         "main.4.weight": torch.randn(128, 128) * 0.09, "main.4.bias": torch.zeros(128),
         "main.6.weight": torch.randn(2, 128) * 0.09, "main.6.bias": torch.zeros(2)}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    vv = {k: torch.zeros_like(v) for k, v in p.items()}
    x = gaussian_mixture_2d(B_C2)
    score = lambda prm, yy, tt: N.mlp_forward(prm, yy, tt, None)

    def one(step):
        t = S.clamp_time(sp, torch.rand(B_C2, 1))
        y = S.vp_perturb(sp, t, x, torch.randn(B_C2, 2))
        v = S.rademacher_from_uniform(torch.rand(B_C2, 2))
        loss, per, g = LR.ssm_mean_and_grads(sp, score, p, t, y, v, form="double_backward")   # as upstream
        for k in p:
            p[k], m[k], vv[k] = LR.adam_step(p[k], g[k], m[k], vv[k], step)

    one(1)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        n += 1
        one(n + 1)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "train_steps/s (B=65536)", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} full C2 train steps (B=65536, MLP d=2, double-backward SSM + Adam) on the CPU oracle in {dt:.1f} s"}


def cpu_baseline_unet(workload, budget_s=15.0):
    """CPU oracle at a reduced batch, scaled linearly to the config batch."""
    import torch
    from oracle import sde_ref as S, nets_ref as N, ssm_ref as LR
    from oracle.det_params import det_state_dict
    from oracle.shapes import unet1d_shapes, unet2d_shapes
    torch.set_num_threads(host_cores())
    sp = S.SdeSpec()
    if workload == "c3":
        b, d, full = 8, 1024, 4096
        p = det_state_dict(unet1d_shapes(1024, None))
        score = lambda prm, yy, tt: N.unet1d_forward(prm, yy, tt, None)
    else:
        b, d, full = 1, 3 * 64 * 64, 256
        cfg = N.UNet2DConfig(in_channels=3, out_channels=3, in_space=64)
        p = det_state_dict(unet2d_shapes(cfg))
        score = lambda prm, yy, tt: N.image_to_flat(N.unet2d_core_forward(prm, N.flat_to_image(yy, 64, 64, "F", 3), tt.reshape(-1), cfg), "F")
    x = torch.randn(b, d)

    def one():
        t = S.clamp_time(sp, torch.rand(b, 1)); y = S.vp_perturb(sp, t, x, torch.randn(b, d))
        v = S.rademacher_from_uniform(torch.rand(b, d))
        LR.ssm_mean_and_grads(sp, score, p, t, y, v, form="double_backward")
    one()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        one(); n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": 1.0 / (dt * full / b), "unit": f"train_steps/s (B={full})", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} oracle train steps at batch {b} ({dt:.2f} s each, double-backward SSM as upstream, no Adam), "
                      f"scaled linearly to batch {full}"}


# ----------------------------------------------------------------------------------------------------- legs
def leg_train_unet(workload, a, rank, world, dev):
    """C3 (weak: 4096 rows per GPU) / C4 (strong: global 256 split over the ranks)."""
    import torch
    from sdeflow_light_amd import parallel
    from sdeflow_light_amd.train import UNetScoreTrainer
    from sdeflow_light_amd.data import signals_1d, random_images
    gen, d = build_unet(workload, dev)
    flat, _ = gen.a.flat_parameters()
    parallel.broadcast_(flat, 0)
    B = 4096 if workload == "c3" else GLOBAL_BATCH["c4"] // world
    # same seed on every rank + the shard's first global row: the N-rank run draws what the 1-rank run would
    tr = UNetScoreTrainer(gen, B, d, lr=1e-4, world=world, seed=1, row_base=rank * B)
    tr.set_data(signals_1d(B, seed=1234 + rank, device=dev) if workload == "c3" else random_images(B, seed=1234 + rank, device=dev))
    med, blocks = timed_blocks(tr.step, a.steps, a.warmup, dev)
    per_step = med / a.steps
    gb = B * world
    units_per_step = world if workload == "c3" else 1          # c3 weak: B/GPU fixed; c4 strong: global 256
    flops = 6 * UNET_FWD_FLOP[workload] * gb
    tf = flops / per_step / 1e12 / world
    out = {"value": units_per_step / per_step, "ms_per_step": per_step * 1e3, "blocks_s": [round(b, 4) for b in blocks],
           "final_loss": float(tr.loss), "global_batch": gb, "graph": bool(tr.use_graph),
           "whole_step": {"algorithmic_tflops_per_gpu": tf, "frac_of_f32_mfma_peak": tf / PEAK_F32_MFMA_TFLOPS,
                          "flop_per_step": flops, "note": "as-written FLOPs = 6 x forward (SURVEY.md §8d), all kernels"}}
    del tr
    return out, gen, d


def leg_sample_unet(gen, d, a, rank, world, dev):
    """C5: EM sampling of 8192 rows split over the ranks, ONE captured step replayed per chunk of <= 4096 rows."""
    import torch
    from sdeflow_light_amd import parallel
    from sdeflow_light_amd.sde_scheme import GraphedStepSampler
    rows = SAMPLE_ROWS // world
    chunk = min(rows, CHUNK)                       # rows are independent: integrate them in chunks
    N = a.sample_steps
    gen.base_sde.set_shard(rank * rows, d)         # this rank's rows of the global sample set (latent + dW draws)
    gs = GraphedStepSampler(gen, chunk, d, N)      # ONE EM step captured as a hipGraph, device-side clock
    x = gen.latent_sample(rows, d)
    gs.run(x[:chunk])                              # warm-up replay
    parallel.barrier(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for c0 in range(0, rows, chunk):
        x[c0:c0 + chunk].copy_(gs.run(x[c0:c0 + chunk]))
    torch.cuda.synchronize(dev); parallel.barrier()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    flops = UNET_FWD_FLOP["c5"] * SAMPLE_ROWS
    tf = flops * N / dt / 1e12 / world
    out = {"sample_steps_per_s": N / dt, "ms_per_sample_step": dt / N * 1e3, "sample_steps": N, "rows_per_gpu": rows,
           "chunk": chunk, "sampler_finite": bool(torch.isfinite(x).all()),
           "algorithmic_tflops_per_gpu": tf, "frac_of_f32_mfma_peak": tf / PEAK_F32_MFMA_TFLOPS,
           "workload": f"C5: EM sampling, UNet2D 64x64x3, {SAMPLE_ROWS} samples over {world} GPU(s), hipGraph-captured step"}
    del gs, x
    return out


def leg_mlp(a, rank, world, dev, steps=200, sample_steps=200):
    """C2: MLP d=2, B=65536 per GPU (weak): graph-replayed train step + whole-loop EM sampler + k_mlp<TRAIN> roofline."""
    import ctypes as C
    import torch
    from sdeflow_light_amd import parallel, ops
    from sdeflow_light_amd.train import MLPScoreTrainer
    from sdeflow_light_amd.sde_scheme import GraphedEMSampler
    from sdeflow_light_amd.data import gaussian_mixture_2d
    gen = build_mlp(dev)
    flat, _ = gen.a.flat_parameters()
    parallel.broadcast_(flat, 0)
    tr = MLPScoreTrainer(gen, B_C2, lr=1e-3, world=world, use_graph=not parallel.multi(world), seed=1, row_base=rank * B_C2)
    tr.set_data(gaussian_mixture_2d(B_C2, seed=1234 + rank, device=dev))
    med, blocks = timed_blocks(tr.step, steps, 20, dev)
    per_step = med / steps
    gen.base_sde.set_shard(rank * B_C2, 2)
    smp = GraphedEMSampler(gen, B_C2, sample_steps)
    x0 = gen.latent_sample(B_C2, 2)
    smp.run(x0)
    meds, _ = timed_blocks(lambda: smp.run(x0), 3, 0, dev, min_seconds=0.2)
    out = {"workload": "C2: 2-D Gaussian mixture, MLP (d=2, hidden 128), SGM, batch 65536/GPU, SSM + Adam",
           "train_steps_per_s": world / per_step, "ms_per_step": per_step * 1e3, "final_loss": float(tr.loss.item()),
           "sample_steps_per_s": world * 3 * sample_steps / meds,
           "sample_config": f"EM, {sample_steps} steps, 65536 rows/GPU, whole loop = one launch replayed as a hipGraph"}
    if rank == 0:
        lib = ops.lib()
        nsl = C.c_int32(0)
        fn = lambda: ops.check(lib.msgm_mlp_ssm_partial(tr.P, tr.y.data_ptr(), tr.t.data_ptr(), tr.vp.data_ptr(), None, None, B_C2,
                                                        tr.st, tr.inv_batch, None, tr.ws.data_ptr(), tr.ws.numel() * 4,
                                                        C.byref(nsl), ops.stream()), "partial")
        tk = time_kernel_events(fn, 50, dev)
        flop = 6 * MLP_FWD_FLOP * B_C2
        out["roofline"] = {"kernel": "k_mlp<MODE_TRAIN> (msgm_mlp_ssm_partial)", "bound": "mfma", "achieved": flop / tk / 1e12,
                           "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": flop / tk / 1e12 / PEAK_F32_MFMA_TFLOPS,
                           "kernel_ms": tk * 1e3, "flop_per_launch": flop,
                           "traffic": 40.1e6, "traffic_source": "profiles/r01_pmc_traffic.json (round-1 PMC pass, not measured in this run)"}
    del tr, smp
    return out


# ----------------------------------------------------------------------------------------------------- kernel probes
def kernel_probes(dev, Bp):
    """The heavy kernels of the C4 step at their C4 shapes (dual batch N = 2*Bp), each launch bracketed by HIP events
    on the launch stream.  FLOPs per launch are the algorithmic ones (2*MACs of the product(s) as written upstream)."""
    import torch
    from sdeflow_light_amd import ops
    N = 2 * Bp
    probes = []

    def conv(H, Ci, Co):
        x = torch.randn(N * H * H * Ci, device=dev)
        Wp = torch.randn(9 * ops.pad16(Co) * ops.pad16(Ci), device=dev) * 0.05
        out = torch.empty(N * H * H * Co, device=dev)
        geom = ops.conv_geom(N, H, H, H, H, 3, 3, 1, 1)
        return (f"k_conv_tile 3x3 {Ci}->{Co} @ {H}x{H}, dual batch {N} (msgm_conv_forward)", 2 * 9 * Ci * Co * N * H * H,
                lambda: ops.conv_forward(geom, x, Ci, Wp, Co, out, n_bias=Bp), (x, Wp, out))

    def wgrad(H, Ci, Co):
        x = torch.randn(N * H * H * Ci, device=dev)
        gy = torch.randn(N * H * H * Co, device=dev)
        dWp = torch.zeros(9 * ops.pad16(Co) * ops.pad16(Ci), device=dev)
        geom = ops.conv_geom(N, H, H, H, H, 3, 3, 1, 1)
        return (f"k_wgrad_tile 3x3 {Ci}->{Co} @ {H}x{H}, dual batch {N} (msgm_conv_wgrad)", 2 * 9 * Ci * Co * N * H * H,
                lambda: ops.conv_wgrad(geom, gy, x, Ci, 0, dWp, Co, ops.pad16(Co), ops.pad16(Ci)), (x, gy, dWp))

    def attn(T, C):
        """Training attention at the C4 block shape.  Algorithmic products as written upstream (model/unet.py:236-250
        under the dual-number step): forward 6 x 2T^2C (S, 2 for Sdot, P v, Pdot v, P vdot), backward 12 x 2T^2C; the
        fused backward EXECUTES 15 (it recomputes S / Sdot instead of reading (B,T,T) tensors) — not counted."""
        qkv = torch.randn(N * T * 3 * C, device=dev)
        s2 = 1.0 / math.sqrt(C)
        att, stats = ops.attention_dual_forward(qkv, Bp, T, C, s2)
        datt = torch.randn(N * T * C, device=dev)
        ops.attention_dual_backward(qkv, att, datt, stats, Bp, T, C, s2)         # allocates the slab workspace
        prod = 2 * T * T * C * Bp
        return [(f"k_attn_dual_bwd<{C // 16}> T={T} C={C}, batch {Bp} (msgm_attention_dual_backward: delta + main + slab reduce)",
                 12 * prod, lambda: ops.attention_dual_backward(qkv, att, datt, stats, Bp, T, C, s2), (qkv, att, datt, stats)),
                (f"k_attn_dual_fwd<{C // 16}> T={T} C={C}, batch {Bp} (msgm_attention_dual_forward)", 6 * prod,
                 lambda: ops.attention_dual_forward(qkv, Bp, T, C, s2), (qkv,))]

    # shares of the C4 step in the committed rocprofv3 kernel stats (profiles/r02/c4_b256_train_kernel_stats_v6.csv, training
    # steps only): the fused attention backward CALL (main kernel 15.9 % + slab reduce 2.0 % + delta 0.2 %) and the 3x3
    # halo-tile conv in its 64-channel form (19.3 % over all its shapes, forward and dgrad; + 7.8 % in the 32-channel form)
    # are the two largest, then the tiled 3x3 wgrad (13.7 %) and the attention forward (6.6 %).  The attention backward is
    # reported as THE roofline entry: it is one shape, so its live timing can be checked against the rocprofv3 average.
    probes += attn(1024, 64)
    probes.append(conv(32, 64, 64))
    probes.append(wgrad(32, 64, 64))
    share = {"k_attn_dual_bwd": 0.180, "k_attn_dual_fwd": 0.066, "k_conv_tile": 0.193, "k_wgrad_tile": 0.137}
    pmc = {}
    pj = os.path.join(ROOT, "profiles", "r02", "pmc_attention_bp256.json")
    if os.path.exists(pj):
        try:
            pmc = json.load(open(pj))["kernels"]
        except Exception:
            pmc = {}
    pmc2 = {}
    pj2 = os.path.join(ROOT, "profiles", "r02", "pmc_conv_probes.json")     # the conv / wgrad probe shape (tools/probe_conv.py)
    if os.path.exists(pj2):
        try:
            pmc2 = json.load(open(pj2))["kernels"]
        except Exception:
            pmc2 = {}
    res = []
    for name, flop, fn, keep in probes:
        tk = time_kernel_events(fn, 20, dev)
        r = {"kernel": name, "bound": "mfma", "achieved": flop / tk / 1e12, "peak": PEAK_F32_MFMA_TFLOPS,
             "unit": "TFLOP/s", "frac": flop / tk / 1e12 / PEAK_F32_MFMA_TFLOPS, "kernel_ms": tk * 1e3,
             "flop_per_launch": flop, "traffic": None,
             "share_of_c4_step": next((v for k, v in share.items() if name.startswith(k)), None)}
        if name.startswith("k_attn_dual_bwd") and Bp == 256:
            r["executed_tflops"] = flop * 15 / 12 / tk / 1e12     # 15 products run (S / Sdot recomputed), 12 are algorithmic
            t = [v.get("hbm_bytes") for k, v in pmc.items() if k.startswith(("void k_attn_dual_bwd<4> grid=2097152", "k_attn_dq_reduce grid=4194304",
                                                                              "k_attn_dual_delta grid=4194304"))]
            if len(t) == 3 and all(t):
                r["traffic"] = float(sum(t))
                r["traffic_source"] = ("profiles/r02/pmc_attention_bp256.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                       "tools/bench_attn.py, 2 x FETCH + WRITE, the three launches of the call; not measured in this run); "
                                       "algorithmic bytes 0.81 GB (q, k, v, o, their tangents and cotangents once) — the rest is the 2.15 GB of "
                                       "query-gradient slabs written and read back")
        if name.startswith("k_attn_dual_fwd") and Bp == 256:
            t = [v.get("hbm_bytes") for k, v in pmc.items() if k.startswith("void k_attn_dual_fwd<4, 1, 32> grid=1048576")]
            if t and t[0]:
                r["traffic"] = float(t[0])
                r["traffic_source"] = "profiles/r02/pmc_attention_bp256.json (PMC pass, not measured in this run); algorithmic 0.54 GB"
        if name.startswith(("k_conv_tile", "k_wgrad_tile")) and Bp == 256:
            key = "void k_conv_tile<16, 16, 4, 3, 4, false> grid=524288" if name.startswith("k_conv_tile") else "void k_wgrad_tile<8, 16, 9, false> grid=262144"
            t = pmc2.get(key, {}).get("hbm_bytes")
            if t:
                r["traffic"] = float(t)
                r["traffic_source"] = ("profiles/r02/pmc_conv_probes.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/probe_conv.py, "
                                       "2 x FETCH + WRITE; not measured in this run); algorithmic 0.27 GB (input + output / gradient once; the "
                                       "wgrad reads each operand once per 32-channel block of the other: 2x at 64 channels)")
        res.append(r)
        del keep
    return res


# ----------------------------------------------------------------------------------------------------- worker
def worker(a):
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    from sdeflow_light_amd import parallel, ops
    rank, local, world = parallel.init_distributed()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    dev = parallel.local_device(local)
    torch.cuda.set_device(dev)
    ops.lib()
    w = a.workload
    out = {"n_gpus": world, "steps": a.steps, "warmup": a.warmup, "higher_is_better": True, "vs_baseline": None,
           "dtype": "f32", "data": "synthetic"}
    solo = rank == 0 and world == 1

    if w == "c2":
        r = leg_mlp(a, rank, world, dev, steps=a.steps)
        out.update(metric="score-matching train steps/sec (+ reverse-SDE sample steps/sec)", value=r["train_steps_per_s"],
                   unit="train_steps/s (B=65536 per step per GPU)", ms_per_step=r["ms_per_step"], scaling="weak",
                   config={"workload": r["workload"], "global_batch": B_C2 * world, "parallelism": f"dp{world}"},
                   sample_steps_per_s=r["sample_steps_per_s"], final_loss=r["final_loss"], roofline=r.get("roofline"),
                   cpu_baseline=cpu_baseline_mlp() if solo and not a.no_cpu_baseline else None)
    elif w == "c5":
        gen, d = build_unet("c5", dev)
        r = leg_sample_unet(gen, d, a, rank, world, dev)
        out.update(metric="reverse-SDE sample steps/sec", value=r["sample_steps_per_s"], unit="EM steps/s (8192 samples per step)",
                   steps=a.sample_steps, warmup=1, ms_per_step=r["ms_per_sample_step"], scaling="strong",
                   config={"workload": r["workload"], "parallelism": f"dp{world}", "rows_per_gpu": r["rows_per_gpu"]},
                   sampler_finite=r["sampler_finite"],
                   roofline={"kernel": "whole EM step (all kernels)", "bound": "mfma", "achieved": r["algorithmic_tflops_per_gpu"],
                             "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": r["frac_of_f32_mfma_peak"], "traffic": None},
                   cpu_baseline=None)
    else:
        r, gen, d = leg_train_unet(w, a, rank, world, dev)
        name = {"c3": "C3: UNet1D L=1024, batch 4096/GPU, SGM, SSM + Adam",
                "c4": "C4: VorticityUNet 2-D 64x64x3 (4.02 M params), GLOBAL batch 256, SGM, SSM loss + Adam"}[w]
        out.update(metric="score-matching train steps/sec (+ reverse-SDE sample steps/sec)", value=r["value"],
                   unit=f"train_steps/s (global batch {GLOBAL_BATCH[w]}{' per GPU' if w == 'c3' else ''} per step)",
                   ms_per_step=r["ms_per_step"], scaling="weak" if w == "c3" else "strong",
                   config={"workload": name, "global_batch": r["global_batch"], "parallelism": f"dp{world}", "graph": r["graph"]},
                   final_loss=r["final_loss"], timed_blocks_s=r["blocks_s"], whole_step=r["whole_step"])
        free_gpu()
        if w == "c4" and a.sample_steps > 0:
            s = leg_sample_unet(gen, d, a, rank, world, dev)
            out.update(sample_steps_per_s=s["sample_steps_per_s"], sample=s)
        del gen
        free_gpu()
        if solo:
            probes = kernel_probes(dev, GLOBAL_BATCH[w]) if w == "c4" else []
            if probes:
                # the dominant kernel of the C4 step (largest share in profiles/r02's rocprofv3 kernel stats) first
                out["roofline"] = dict(probes[0], whole_step_frac=r["whole_step"]["frac_of_f32_mfma_peak"])
                out["roofline"].setdefault("traffic_source", "not measured in this run; PMC passes are under profiles/")
                out["roofline_other_kernels"] = probes[1:]
            else:
                ws = r["whole_step"]
                out["roofline"] = {"kernel": "whole step (all kernels)", "bound": "mfma", "achieved": ws["algorithmic_tflops_per_gpu"],
                                   "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ws["frac_of_f32_mfma_peak"], "traffic": None}
            free_gpu()
            if w == "c4" and not a.no_extra:
                extra = {}
                extra["c2"] = leg_mlp(a, rank, world, dev)
                free_gpu()
                ac3 = argparse.Namespace(steps=5, warmup=2)
                r3, g3, _ = leg_train_unet("c3", ac3, rank, world, dev)
                del g3
                extra["c3"] = {"workload": "C3: UNet1D L=1024, batch 4096, SGM, SSM + Adam", "train_steps_per_s": r3["value"],
                               "ms_per_step": r3["ms_per_step"], "whole_step": r3["whole_step"], "final_loss": r3["final_loss"]}
                out["extra"] = extra
                free_gpu()
            out["cpu_baseline"] = None if a.no_cpu_baseline else cpu_baseline_unet(w)
        else:
            ws = r["whole_step"]
            out["roofline"] = {"kernel": "whole step (all kernels), per GPU", "bound": "mfma", "achieved": ws["algorithmic_tflops_per_gpu"],
                               "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ws["frac_of_f32_mfma_peak"], "traffic": None}
            out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    parallel.barrier()
    if parallel.multi(world):
        import torch.distributed as dist
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["c2", "c3", "c4", "c5"], default="c4",
                    help="c4 (default, the driver's line): 2-D U-Net 64x64x3 global batch 256 (+ C5 sampling leg, kernel "
                         "rooflines, C2/C3 extras, CPU baseline); c2 / c3 / c5: the other BASELINE.json configs as the main line")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sample-steps", type=int, default=20, help="EM steps of the C5 leg (0 = skip it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the C2 / C3 extra legs")
    a = ap.parse_args()
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and env_world == 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(a))              # before ANY GPU call in this process
    worker(a)


if __name__ == "__main__":
    main()
