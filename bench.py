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
  roofline            the kernel family with the LARGEST MEASURED SHARE of the C4 step: one eager step of the same kernels
                      with every launch bracketed by HIP events on the launch stream (StepProfiler), algorithmic FLOPs from
                      each call's own geometry; step_profile lists every family (share, TFLOP/s or GB/s, fraction of peak);
  roofline_hbm        the HBM-bound kernels (EM stage, perturbation, prep, Adam, GroupNorm) as GB/s of 8 TB/s and us/launch;
  cpu_baseline        the CPU oracle on this host's cores at batch 8 incl. Adam, scaled to 256 (rank 0, N=1 only);
  extra               C2 (MLP, B=65536), C3 (UNet1D L=1024, B=4096) and MSGM (multiplicative SDE: UNet1D d=1024 and the C4
                      net at d=12288, sparse tensor, nsf=16) short legs (N=1 only).
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
_T0 = time.perf_counter()


def log(msg: str) -> None:
    """Progress on stderr (rank 0 prints the ONE JSON line on stdout at the end)."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def host_cores() -> int:
    """CPU threads this process can actually use: the scheduler affinity, cut down to the cgroup CPU quota when there is one
    (a GPU box hands each GPU a 16-CPU share of a much larger host through the quota, not the affinity mask: 200+ threads
    on a 16-CPU quota thrash) and to 64 (more threads do not speed a batch-8 U-Net step up)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(math.ceil(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, int(math.ceil(q / per)))
        except Exception:
            quota = None
    n = min(avail, quota) if quota else min(avail, 16)       # no quota visible: the documented 16-CPU share of a 1-GPU box
    return max(1, min(n, 64))


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


def cpu_baseline_unet(workload, budget_s=20.0):
    """CPU oracle at the reduced batch SURVEY.md §8(d) / BASELINE.md §3 prescribe (C3: 64 rows -> here 8, C4: 8 samples),
    the whole training step as upstream runs it — double-backward SSM loss + Adam (MSGM_higherDim.py:803-809) — timed on
    this host's cores and scaled linearly to the config batch."""
    import torch
    from oracle import sde_ref as S, nets_ref as N, ssm_ref as LR
    from oracle.det_params import init_like_state_dict
    from oracle.shapes import unet1d_shapes, unet2d_shapes
    torch.set_num_threads(host_cores())
    sp = S.SdeSpec()
    if workload == "c3":
        b, d, full = 8, 1024, 4096
        p = init_like_state_dict(unet1d_shapes(1024, None))
        score = lambda prm, yy, tt: N.unet1d_forward(prm, yy, tt, None)
    else:
        b, d, full = 8, 3 * 64 * 64, 256
        cfg = N.UNet2DConfig(in_channels=3, out_channels=3, in_space=64)
        p = init_like_state_dict(unet2d_shapes(cfg))
        score = lambda prm, yy, tt: N.image_to_flat(N.unet2d_core_forward(prm, N.flat_to_image(yy, 64, 64, "F", 3), tt.reshape(-1), cfg), "F")
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    vv = {k: torch.zeros_like(v) for k, v in p.items()}
    x = torch.randn(b, d)

    def one(step):
        t = S.clamp_time(sp, torch.rand(b, 1)); y = S.vp_perturb(sp, t, x, torch.randn(b, d))
        v = S.rademacher_from_uniform(torch.rand(b, d))
        _, _, g = LR.ssm_mean_and_grads(sp, score, p, t, y, v, form="double_backward")
        for k in p:
            p[k], m[k], vv[k] = LR.adam_step(p[k], g[k], m[k], vv[k], step, lr=1e-4)
    one(1)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        n += 1
        one(n + 1)
    dt = (time.perf_counter() - t0) / n
    return {"value": 1.0 / (dt * full / b), "unit": f"train_steps/s (B={full})", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} oracle train steps at batch {b} ({dt:.2f} s each: double-backward SSM as upstream + Adam), "
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


# ----------------------------------------------------------------------------------------------------- step profile
PEAK_HBM_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E peak 8 TB/s (6.3 TB/s measured float4 copy)


class StepProfiler:
    """Per-launch timing of ONE eager training step, measured in this run: every C-ABI call of the step is bracketed by its
    own pair of HIP events on the launch stream, and calls are grouped into kernel families with their ALGORITHMIC work —
    FLOPs as written upstream (2*MACs from the call's own geometry arguments; attention: 6 / 12 products of 2 T^2 C per
    sample forward / backward) or bytes (GroupNorm: one read + one write of the dual tensor forward, two reads + one write
    backward).  `roofline` is the family with the largest measured share — nothing is typed in."""

    def __init__(self):
        self.recs = []

    def _family(self, name, args):
        from sdeflow_light_amd import _lib as L
        g = args[0] if args and isinstance(args[0], L.ConvGeomT) else None
        if name in ("msgm_conv_forward", "msgm_conv_forward_fused", "msgm_conv_forward_wino") and g is not None:
            cin = int(args[2]) + (int(args[4]) if args[3] else 0)
            flop = 2.0 * g.KH * g.KW * cin * int(args[6]) * g.N * g.Ho * g.Wo
            self._bytes = 4.0 * g.N * (g.Hi * g.Wi * cin + g.Ho * g.Wo * int(args[6]))      # input + output once
            if name == "msgm_conv_forward_wino":
                # AS-WRITTEN FLOPs of the 3x3 convolution (2 x 9 x Cin x Cout per pixel); the kernel executes 2.25x fewer
                return "3x3 stride-1 conv forward + dgrad, Winograd F(2x2,3x3) (k_conv_wino; msgm_conv_forward_wino)", "mfma", flop
            k3 = g.KH * g.KW == 9 or (g.KH == 1 and g.KW == 3)
            same = g.strideH == 1 and g.strideW == 1
            if k3 and same and cin >= 16 and int(args[6]) >= 16:
                return "3x3 stride-1 conv forward + dgrad (k_conv_tile; msgm_conv_forward_fused)", "mfma", flop
            if g.KH * g.KW == 1:
                return "1x1 conv / linear forward + dgrad (k_conv1x1)", "mfma", flop
            return "other conv forward + dgrad (strided / 3-channel in-out)", "mfma", flop
        if name in ("msgm_conv_wgrad_slabs", "msgm_conv_wgrad_det", "msgm_conv_wgrad") and g is not None:
            flop = 2.0 * g.KH * g.KW * int(args[3]) * int(args[6]) * g.N * g.Ho * g.Wo
            self._bytes = 4.0 * g.N * (g.Hi * g.Wi * int(args[3]) + g.Ho * g.Wo * int(args[6]))
            if g.KH * g.KW == 9 and g.strideH == 1:
                return "3x3 stride-1 conv wgrad (k_wgrad_tile; msgm_conv_wgrad_slabs)", "mfma", flop
            return "1x1 / strided conv wgrad", "mfma", flop
        if name == "msgm_attention_dual_forward":
            Bp, T, C = int(args[3]), int(args[4]), int(args[5])
            self._bytes = 4.0 * 2 * Bp * T * 4 * C                     # q, k, v in, o out (dual)
            return f"dual attention forward C={C} (k_attn_dual_fwd)", "mfma", 6 * 2.0 * T * T * C * Bp
        if name == "msgm_attention_dual_backward":
            Bp, T, C = int(args[5]), int(args[6]), int(args[7])
            self._bytes = 4.0 * 2 * Bp * T * 8 * C                     # q, k, v, o, obar in, qbar, kbar, vbar out (dual)
            return f"dual attention backward C={C} (delta + k_attn_dual_bwd + slab reduce)", "mfma", 12 * 2.0 * T * T * C * Bp
        if name in ("msgm_groupnorm_dual_forward", "msgm_groupnorm_dual_forward2"):
            if name.endswith("2"):
                Bp, P, C = int(args[8]), int(args[9]), int(args[1]) + int(args[3])
            else:
                Bp, P, C = int(args[5]), int(args[6]), int(args[7])
            return "GroupNorm(+SiLU) dual forward (2 launches)", "hbm", 2 * 2.0 * Bp * P * C * 4
        if name in ("msgm_groupnorm_dual_backward", "msgm_groupnorm_dual_backward2", "msgm_groupnorm_dual_backward_slots"):
            if name.endswith("_slots"):
                Bp, P, C = int(args[12]), int(args[13]), int(args[1]) + int(args[3])
            elif name.endswith("2"):
                Bp, P, C = int(args[12]), int(args[13]), int(args[1]) + int(args[3])
            else:
                Bp, P, C = int(args[8]), int(args[9]), int(args[10])
            return "GroupNorm(+SiLU) dual backward (2 launches)", "hbm", 3 * 2.0 * Bp * P * C * 4
        if name == "msgm_adam_step":
            return "Adam (k_adam)", "hbm", 28.0 * int(args[4])
        return "other (layout glue, embedding MLPs, reductions, loss)", None, 0.0

    def install(self):
        import torch
        from sdeflow_light_amd import _lib as L
        lib = L.lib()
        self._orig = {}
        for name in L.SIGNATURES:
            fn = getattr(lib, name)
            if fn.restype is not __import__("ctypes").c_int or name in ("msgm_version",):
                continue
            self._orig[name] = fn

            def wrap(*args, _fn=fn, _name=name):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = _fn(*args)
                e1.record()
                self._bytes = 0.0
                fam = self._family(_name, args)
                self.recs.append((_name, fam + (self._bytes,), e0, e1))
                return rc
            setattr(lib, name, wrap)
        self._lib = lib

    def remove(self):
        for name, fn in self._orig.items():
            setattr(self._lib, name, fn)

    def summary(self, dev):
        import torch
        torch.cuda.synchronize(dev)
        fam = {}
        total = 0.0
        for name, (family, bound, work, byts), e0, e1 in self.recs:
            ms = e0.elapsed_time(e1)
            total += ms
            f = fam.setdefault(family, {"family": family, "bound": bound, "calls": 0, "ms": 0.0, "work": 0.0, "bytes": 0.0})
            f["calls"] += 1; f["ms"] += ms; f["work"] += work; f["bytes"] += byts
        out = []
        for f in sorted(fam.values(), key=lambda v: -v["ms"]):
            r = {"family": f["family"], "calls": f["calls"], "ms": round(f["ms"], 3), "share_of_step": round(f["ms"] / total, 4)}
            if f["bound"] == "mfma":
                tf = f["work"] / (f["ms"] * 1e-3) / 1e12
                r.update(bound="mfma", achieved=tf, peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=tf / PEAK_F32_MFMA_TFLOPS,
                         flop=f["work"], algorithmic_bytes_per_launch=f["bytes"] / max(f["calls"], 1))
            elif f["bound"] == "hbm":
                gbs = f["work"] / (f["ms"] * 1e-3) / 1e9
                r.update(bound="hbm", achieved=gbs, peak=PEAK_HBM_GBS, unit="GB/s", frac=gbs / PEAK_HBM_GBS, bytes=f["work"])
            out.append(r)
        return out, total


def leg_step_profile(dev, Bp):
    """One EAGER C4 step (same kernels and shapes as the graph-replayed one) with every launch timed: the measured shares
    and the as-written rate of each kernel family.  Small launches include the host's launch gap (eager), so the sum is a
    few percent above the graph-replayed step; the MFMA families are >= 95 % of it and are not affected."""
    import torch
    from sdeflow_light_amd.train import UNetScoreTrainer
    from sdeflow_light_amd.data import random_images
    gen, d = build_unet("c4", dev)
    tr = UNetScoreTrainer(gen, Bp, d, lr=1e-4, world=1, seed=1, use_graph=False)
    tr.set_data(random_images(Bp, seed=1234, device=dev))
    for _ in range(2):
        tr.step()
    torch.cuda.synchronize(dev)
    sp = StepProfiler()
    sp.install()
    try:
        tr.step()
        fams, total = sp.summary(dev)
    finally:
        sp.remove()
    del tr, gen
    return fams, total


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of a kernel from the PMC passes committed under profiles/r03 (rocprofv3 --pmc FETCH_SIZE and
    WRITE_SIZE in separate runs, 2 x FETCH + WRITE as MI355X_MICROARCH.md prescribes for gfx950); None if absent."""
    pj = os.path.join(ROOT, "profiles", "r03", "pmc_c4_step.json")
    try:
        ks = json.load(open(pj))["kernels"]
    except Exception:
        return None, None
    hits = [v for k, v in ks.items() if kernel_prefix in k and v.get("hbm_bytes")]
    if not hits:
        return None, None
    n = sum(v.get("dispatches", 1) for v in hits)
    return (sum(v["hbm_bytes"] * v.get("dispatches", 1) for v in hits) / n,
            f"profiles/r03/pmc_c4_step.json: average HBM bytes per launch of {kernel_prefix}* over the C4 step's {n // 3} launches "
            "(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of tools/time_unet2d.py 256 1, 2 x FETCH + WRITE; "
            "not measured in this run)")


def hbm_rooflines(dev):
    """The HBM-bound kernels of the path timed live: algorithmic bytes / time against the 8 TB/s peak at the
    C4 / C5 sizes, and microseconds per launch at the C2 size (512 KB tensors: latency-bound, report us)."""
    import torch
    from sdeflow_light_amd import ops, _lib as L
    st = L.sde_struct(0, 0.1, 20.0, 1.0, 1e-3)
    rng = L.PhiloxState(7, dev)
    res = []

    def add(name, byts, fn, note):
        t = time_kernel_events(fn, 30, dev)
        res.append({"kernel": name, "bound": "hbm", "achieved": byts / t / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": byts / t / 1e9 / PEAK_HBM_GBS, "us_per_launch": t * 1e6, "algorithmic_bytes": byts, "size": note})
    for B, d, tag in ((4096, 12288, "C5 chunk 4096 x 12288"), (65536, 2, "C2 65536 x 2 (latency-bound: read us_per_launch)")):
        n = B * d
        x, a = torch.randn(B, d, device=dev), torch.randn(B, d, device=dev)
        y, v = torch.empty_like(x), torch.empty_like(x)
        t = torch.empty(B, device=dev)
        step = torch.zeros(1, dtype=torch.int64, device=dev)
        add("k_stage_diag_flat (msgm_sde_stage, EM step)", 12 * n,
            lambda: ops.sde_stage(x, x, 1.0, x, a, st, L.PROC_REVERSE, False, 0.5, 1e-3, 0.0, rng=rng, rng_step=1), tag)
        add("k_perturb_vp (msgm_perturb_vp)", 8 * n, lambda: ops.perturb_vp(x, st, rng=rng), tag)
        lib = ops.lib()
        add("k_ssm_prep (msgm_ssm_prep: perturb + probe + tick)", 16 * n,
            lambda: ops.check(lib.msgm_ssm_prep(x.data_ptr(), y.data_ptr(), t.data_ptr(), v.data_ptr(), B, d, st, rng.ptr(),
                                                step.data_ptr(), ops.stream()), "msgm_ssm_prep"), tag)
        del x, a, y, v
    npar = 4023233 + 9 * 3 * 32 * 2          # the C4 net (3-channel in / out convs)
    p, g, m, vv = (torch.randn(npar, device=dev) for _ in range(4)); vv.abs_()
    add("k_adam (msgm_adam_step)", 28 * npar, lambda: ops.adam_step(p, g, m, vv, step=3, lr=1e-4), "C4 net, 4.02 M parameters")
    del p, g, m, vv
    Bp, P, C = 256, 64 * 64, 32               # the largest GroupNorm of the C4 step
    xg = torch.randn(2 * Bp * P * C, device=dev)
    ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    og = torch.empty_like(xg)
    stats = torch.empty(Bp * 32 * 4, device=dev)
    add("k_gn_fwd_reduce + k_gn_fwd_apply (msgm_groupnorm_dual_forward)", 2 * 4 * xg.numel(),
        lambda: ops.groupnorm_dual_forward(xg, ga, be, Bp, P, C, 32, True, True, stats=stats, out=og), "C4 64x64x32, dual batch 512")
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    gg = torch.randn_like(xg)
    add("k_gn_bwd_reduce + param reduce + k_gn_bwd_apply (msgm_groupnorm_dual_backward)", 3 * 4 * xg.numel(),
        lambda: ops.groupnorm_dual_backward(xg, ga, be, stats, gg, dg, db, Bp, P, C, 32, True, gx=og), "C4 64x64x32, dual batch 512")
    return res


# ----------------------------------------------------------------------------------------------------- MSGM leg
def leg_msgm(dev, steps=5, warmup=2):
    """The paper's model — the MULTIPLICATIVE SDE (MSGMsde, sparse rotation tensor for d >= 256, nsf = 16 forward RK4 steps,
    NormalizeLogRadius conditioning: MSGM_higherDim.py:704-746) — through the graph-captured trainers, reported separately
    as SURVEY.md §8(d) asks: UNet1D d = 1024 (batch 4096) and the C4 net at d = 12288 (batch 256), plus the forward-SDE
    stage kernel's algorithmic HBM rate (12 B / element / stage)."""
    import torch
    from sdeflow_light_amd import ops, _lib as L
    from sdeflow_light_amd.SDEs import MSGMsde, PluginReverseSDE
    from sdeflow_light_amd.train import UNetScoreTrainer
    from sdeflow_light_amd.data import signals_1d, random_images
    out = {}
    for tag in ("unet1d_d1024", "unet2d_d12288"):
        torch.manual_seed(0)
        if tag == "unet1d_d1024":
            from sdeflow_light_amd.NNUnet1D import UNet1D
            B, d, lr = 4096, 1024, 1e-4
            net = UNet1D(input_dim=d, base_channels=32, channel_mults=(1, 2, 4), num_res_blocks=2, emb_dim=128,
                         premodule="NormalizeLogRadius").to(dev)
            x = signals_1d(B, seed=1234, device=dev)
            fwd = 0.4554e9
        else:
            from sdeflow_light_amd.NNUnet import VorticityUNet
            B, d, lr = 256, 3 * 64 * 64, 1e-4
            net = VorticityUNet(base_channels=32, channel_mults=(1, 2, 4), num_res_blocks=2, in_space=64, attention_resolutions=(2, 4),
                                flatten_order="F", channels=3, premodule="NormalizeLogRadius").to(dev)
            with torch.no_grad():
                for p_ in net.parameters():
                    if p_.dim() > 1 and float(p_.abs().sum()) == 0.0:
                        p_.normal_(0, 0.02)
            x = random_images(B, seed=1234, device=dev)
            fwd = UNET_FWD_FLOP["c4"]
        T = torch.nn.Parameter(torch.FloatTensor([1.0]), requires_grad=False)
        base = MSGMsde(x[:1024].cpu(), beta_min=0.1, beta_max=20.0, t_epsilon=1e-3, T=T, num_steps_forward=16, device=dev,
                       denseTensor=False, norm_map="log")
        gen = PluginReverseSDE(base, net, T, vtype="rademacher", deviceReverseSDE=dev).to(dev)
        tr = UNetScoreTrainer(gen, B, d, lr=lr, world=1, seed=1)
        tr.set_data(x)
        log(f"msgm {tag}: capture + timing")
        med, blocks = timed_blocks(tr.step, steps, warmup, dev, min_seconds=0.5)
        per = med / steps
        n = B * d
        xs, o = torch.randn(B, d, device=dev), torch.empty(B, d, device=dev)
        st = base.struct()
        tk = time_kernel_events(lambda: ops.sde_stage(o, xs, 0.5, xs, None, st, L.PROC_FORWARD, True, 0.1, 1.0 / 16, 0.0,
                                                      rng=tr.rng, rng_step=1), 30, dev)
        out[tag] = {"workload": f"MSGM sparse tensor, nsf=16, {'UNet1D L=1024' if d == 1024 else 'VorticityUNet 64x64x3'}, batch {B}, "
                                "NormalizeLogRadius, SSM + Adam, hipGraph-captured step (kernel nodes only)",
                    "train_steps_per_s": 1.0 / per, "ms_per_step": per * 1e3, "final_loss": float(tr.loss),
                    "whole_step_frac_of_f32_mfma_peak": 6 * fwd * B / per / 1e12 / PEAK_F32_MFMA_TFLOPS,
                    "graph_kernel_nodes": ops.graph_node_kinds(tr.graph).get("kernel"),
                    "forward_sde_stage": {"kernel": "k_stage_rows (msgm_sde_stage, sparse stencil, Stratonovich stage)", "bound": "hbm",
                                          "achieved": 12 * n / tk / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                          "frac": 12 * n / tk / 1e9 / PEAK_HBM_GBS, "us_per_launch": tk * 1e6,
                                          "launches_per_step": 4 * 16 + 4}}
        del tr, gen, net, xs, o
        free_gpu()
    return out


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
        log(f"{w}: training leg")
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
            log("c5: sampling leg")
            s = leg_sample_unet(gen, d, a, rank, world, dev)
            out.update(sample_steps_per_s=s["sample_steps_per_s"], sample=s)
            if world == 1 and not a.no_extra:
                # opt-in experiment, reported beside the line, never on it: the SAME sampling leg with the 3x3
                # convolutions in bf16-split arithmetic (tests: fp32-grade against float64 and against the default sampler)
                log("c5: sampling leg, bf16-split 3x3 convolutions (opt-in experiment)")
                free_gpu()
                os.environ["MSGM_SAMPLER_BF16X3"] = "1"
                try:
                    s6 = leg_sample_unet(gen, d, a, rank, world, dev)
                finally:
                    del os.environ["MSGM_SAMPLER_BF16X3"]
                s6["dtype"] = ("3x3 convolutions: bf16 split, 3 bf16 pieces per fp32 operand, 6 v_mfma_f32_16x16x32_bf16 products per fp32 "
                               "product, fp32 accumulate (fp32-grade: tests/test_conv_gpu.py::test_bf16_split_conv_is_fp32_grade); everything else f32")
                s6["switch"] = "MSGM_SAMPLER_BF16X3=1 (opt-in; the default line and `sample` above stay f32)"
                out["sample_bf16_split_experiment"] = s6
        del gen
        free_gpu()
        if solo:
            if w == "c4":
                log("c4: per-launch profile of one eager step")
                fams, total = leg_step_profile(dev, GLOBAL_BATCH[w])
                top = next(f for f in fams if f.get("bound") == "mfma")          # largest measured share among the MFMA families
                kern = ("k_conv_wino" if "k_conv_wino" in top["family"] else "k_conv_tile" if "k_conv_tile" in top["family"] else
                        "k_attn_dual_bwd" if "backward" in top["family"] else "k_wgrad_tile" if "wgrad" in top["family"] else None)
                traffic, src = pmc_traffic(kern) if kern else (None, None)
                out["roofline"] = {"kernel": top["family"], "bound": "mfma", "achieved": top["achieved"], "peak": PEAK_F32_MFMA_TFLOPS,
                                   "unit": "TFLOP/s", "frac": top["frac"], "traffic": traffic,
                                   "traffic_source": src or "no PMC pass committed for this kernel",
                                   "algorithmic_bytes_per_launch": top.get("algorithmic_bytes_per_launch"),
                                   "kernel_ms_per_step": top["ms"], "launches_per_step": top["calls"], "flop_per_step": top["flop"],
                                   "share_of_c4_step": top["share_of_step"],
                                   "measured": "this run: one eager C4 step, every launch bracketed by HIP events on the launch stream",
                                   "whole_step_frac": r["whole_step"]["frac_of_f32_mfma_peak"]}
                if kern == "k_conv_wino":
                    out["roofline"]["note"] = ("achieved = AS-WRITTEN FLOPs of the 3x3 convolutions (2 x 9 x Cin x Cout per output pixel) / measured "
                                               "time; the Winograd F(2x2,3x3) kernel executes 2.25x fewer multiplications: executed rate = achieved / 2.25")
                    out["roofline"]["executed_tflops"] = top["achieved"] / 2.25
                    out["roofline"]["executed_frac"] = top["frac"] / 2.25
                out["step_profile"] = {"eager_step_ms": total, "families": fams}
                free_gpu()
                log("HBM-bound kernels")
                out["roofline_hbm"] = hbm_rooflines(dev)
            else:
                ws = r["whole_step"]
                out["roofline"] = {"kernel": "whole step (all kernels)", "bound": "mfma", "achieved": ws["algorithmic_tflops_per_gpu"],
                                   "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ws["frac_of_f32_mfma_peak"], "traffic": None}
            free_gpu()
            if w == "c4" and not a.no_extra:
                extra = {}
                log("extra: c2")
                extra["c2"] = leg_mlp(a, rank, world, dev)
                free_gpu()
                log("extra: c3")
                ac3 = argparse.Namespace(steps=5, warmup=2)
                r3, g3, _ = leg_train_unet("c3", ac3, rank, world, dev)
                del g3
                extra["c3"] = {"workload": "C3: UNet1D L=1024, batch 4096, SGM, SSM + Adam", "train_steps_per_s": r3["value"],
                               "ms_per_step": r3["ms_per_step"], "whole_step": r3["whole_step"], "final_loss": r3["final_loss"]}
                free_gpu()
                log("extra: msgm")
                extra["msgm"] = leg_msgm(dev)
                out["extra"] = extra
                free_gpu()
            log("cpu baseline (oracle on the host cores)")
            out["cpu_baseline"] = None if a.no_cpu_baseline else cpu_baseline_unet(w)
            log("done")
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
