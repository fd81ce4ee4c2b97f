#!/usr/bin/env python3
"""Headline benchmark of the MI355X hot path (driver contract: see task brief).

Workload at N=1 (BASELINE.json configs[1], "C2"): 2-D Gaussian mixture, MLP
score net (33 794 params), SGM, batch 65 536 per GPU, fp32.
  step   = one score-matching training step: perturb (K1) + Rademacher probe +
           fused forward/tangent/loss/backward (K5) [+ RCCL all-reduce of the
           flat gradient bucket when N>1] + fused Adam (K13), inputs resident
           in HBM, hipGraph-replayed on one GPU.
  value  = global training samples per second / 65 536, i.e. C2-sized train
           steps per second summed over all ranks (weak scaling: the per-GPU
           batch is fixed).
Also reported on the same line: reverse-SDE Euler–Maruyama sampler steps/s on
the same config (whole loop = one hipGraph), the roofline of the dominant
kernel (k_mlp<train>), and the CPU oracle timed on this host (rank 0, N=1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_C2 = 65536
MLP_FWD_FLOP = 66816                 # SURVEY.md App. A.3 (2*MACs, d=2)
TRAIN_FLOP_PER_SAMPLE = 6 * MLP_FWD_FLOP
PEAK_F32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md: v_mfma_f32_* = fp32 vector rate


def build_model(dev, B):
    from sdeflow_light_amd.NN import MLP
    from sdeflow_light_amd.SDEs import SGMsde, PluginReverseSDE
    torch.manual_seed(0)                                   # model init seed as upstream (MSGM_higherDim.py:41)
    net = MLP(input_dim=2, index_dim=1, hidden_dim=128).to(dev)
    T = torch.nn.Parameter(torch.FloatTensor([1.0]), requires_grad=False)
    sde = SGMsde(beta_min=0.1, beta_max=20.0, t_epsilon=1e-3, T=T, num_steps_forward=16, device=dev)
    return PluginReverseSDE(sde, net, T, vtype="rademacher", deviceReverseSDE=dev).to(dev)


def time_kernel_events(fn, iters, dev):
    """Average duration of ONE launch of fn: every launch is bracketed by its own pair of HIP events on the launch
    stream (start-to-end of the kernel, as rocprofv3's per-kernel duration counts it — the cross-check committed under
    profiles/).  A loop of back-to-back launches timed as a whole reads ~5 % lower (the next launch's workgroups start
    while the previous one drains) and an eager loop ~3 % higher (host launch gaps); neither is the kernel's duration."""
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


def cpu_baseline(budget_s=12.0):
    """The CPU oracle (restatement of the reference, pinned by golden vectors)
    on the same workload, bounded to ~10-20 s of CPU work."""
    from oracle import sde_ref as S, nets_ref as N, ssm_ref as LR
    from sdeflow_light_amd.data import gaussian_mixture_2d
    torch.manual_seed(0)
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    cores = min(avail, 16)            # a 1-GPU box is given a 16-CPU share; more threads only oversubscribe
    torch.set_num_threads(cores)
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


UNET_FWD_FLOP = {"c3": 0.4554e9, "c4": 5.974e9 + 9.4e6, "c5": 5.974e9 + 9.4e6}    # per sample, SURVEY.md App. A


def build_unet(workload, dev):
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


def cpu_baseline_unet(workload, budget_s=15.0):
    """CPU oracle at a reduced batch, scaled linearly to the config batch."""
    from oracle import sde_ref as S, nets_ref as N, ssm_ref as LR
    from oracle.det_params import det_state_dict
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_golden import unet1d_shapes, unet2d_shapes
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    torch.set_num_threads(min(avail, 16))
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
            "sample": f"{n} oracle train steps at batch {b} ({dt:.2f} s each, double-backward SSM, no Adam), scaled linearly to batch {full}"}


def bench_unet(a, workload):
    """Secondary workloads (not the driver's default line): c3 = UNet1D L=1024 B=4096/GPU (weak);
    c4 = UNet2D 64x64x3, global batch 256 split over the ranks (strong); c5 = EM sampling with the c4 net,
    8192 rows split over the ranks (strong), --sample-steps steps."""
    from sdeflow_light_amd import parallel, ops, _lib as L
    from sdeflow_light_amd.train import UNetScoreTrainer
    from sdeflow_light_amd.data import signals_1d, random_images
    rank, local, world = parallel.init_distributed()
    dev = parallel.local_device(local)
    torch.cuda.set_device(dev)
    gen, d = build_unet(workload, dev)
    flat, _ = gen.a.flat_parameters()
    parallel.broadcast_(flat, 0)
    out = {"n_gpus": world, "dtype": "f32", "data": "synthetic", "higher_is_better": True, "vs_baseline": None}
    if workload in ("c3", "c4"):
        B = 4096 if workload == "c3" else 256 // world
        tr = UNetScoreTrainer(gen, B, d, lr=1e-4, world=world, seed=1 + rank)
        tr.set_data(signals_1d(B, seed=1234 + rank, device=dev) if workload == "c3" else random_images(B, seed=1234 + rank, device=dev))
        for _ in range(a.warmup):
            tr.step()
        parallel.barrier(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            tr.step()
        torch.cuda.synchronize(dev); parallel.barrier()
        dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
        gb = B * world
        units = (world if workload == "c3" else 1) * a.steps          # c3 weak: B/GPU fixed; c4 strong: global 256
        flops = 6 * UNET_FWD_FLOP[workload] * gb * a.steps
        out.update(metric="score-matching train steps/sec", value=units / dt, steps=a.steps, warmup=a.warmup,
                   unit=f"train_steps/s (B={4096 if workload == 'c3' else 256} per step)", ms_per_step=dt / a.steps * 1e3,
                   scaling="weak" if workload == "c3" else "strong",
                   config={"workload": {"c3": "C3: UNet1D L=1024, batch 4096/GPU, SGM, SSM + Adam",
                                        "c4": "C4: UNet2D 64x64x3, global batch 256, SGM, SSM + Adam"}[workload],
                           "global_batch": gb, "parallelism": f"dp{world}"},
                   final_loss=float(tr.loss), algorithmic_tflops=flops / dt / 1e12,
                   roofline={"bound": "mfma", "achieved": flops / dt / 1e12 / world, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                             "frac": flops / dt / 1e12 / world / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                             "note": "whole step, all kernels (as-written FLOPs 6 x forward)"},
                   cpu_baseline=cpu_baseline_unet(workload) if (rank == 0 and world == 1 and not a.no_cpu_baseline) else None)
    else:
        from sdeflow_light_amd.sde_scheme import GraphedStepSampler
        rows = 8192 // world
        chunk = min(rows, 1024)                       # rows are independent: integrate them in chunks of 1024
        N = a.sample_steps
        gs = GraphedStepSampler(gen, chunk, d, N)     # ONE EM step captured as a hipGraph, device-side clock
        x = gen.latent_sample(rows, d)
        gs.run(x[:chunk])                             # warm-up replay
        parallel.barrier(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for c0 in range(0, rows, chunk):
            x[c0:c0 + chunk].copy_(gs.run(x[c0:c0 + chunk]))
        torch.cuda.synchronize(dev); parallel.barrier()
        dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
        flops = UNET_FWD_FLOP["c5"] * 8192 * N
        out.update(metric="reverse-SDE sample steps/sec", value=N / dt, steps=N, warmup=1, unit="EM steps/s (8192 samples per step)",
                   ms_per_step=dt / N * 1e3, scaling="strong",
                   config={"workload": "C5: EM sampling, UNet2D 64x64x3, 8192 samples, hipGraph-captured step", "parallelism": f"dp{world}",
                           "rows_per_gpu": rows, "chunk": chunk},
                   algorithmic_tflops=flops / dt / 1e12, sampler_finite=bool(torch.isfinite(x).all()),
                   roofline={"bound": "mfma", "achieved": flops / dt / 1e12 / world, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                             "frac": flops / dt / 1e12 / world / PEAK_F32_MFMA_TFLOPS, "traffic": None}, cpu_baseline=None)
    if rank == 0:
        print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["c2", "c3", "c4", "c5"], default="c2",
                    help="c2 (default, the driver's line): MLP d=2 B=65536; c3/c4/c5: U-Net configs of BASELINE.json")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--sample-steps", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    if a.workload != "c2":
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
        return bench_unet(a, a.workload)

    from sdeflow_light_amd import parallel, ops
    from sdeflow_light_amd.train import MLPScoreTrainer
    from sdeflow_light_amd.sde_scheme import GraphedEMSampler
    from sdeflow_light_amd.data import gaussian_mixture_2d

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    rank, local, world = parallel.init_distributed()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dev = parallel.local_device(local)
    torch.cuda.set_device(dev)
    ops.lib()

    gen = build_model(dev, B_C2)
    flat, _ = gen.a.flat_parameters()
    parallel.broadcast_(flat, 0)                            # identical params on all ranks
    tr = MLPScoreTrainer(gen, B_C2, lr=1e-3, world=world, use_graph=(world == 1), seed=1 + rank)
    tr.set_data(gaussian_mixture_2d(B_C2, seed=1234 + rank, device=dev))

    # ---- timed training loop ------------------------------------------------
    for _ in range(a.warmup):
        tr.step()
    parallel.barrier(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.step()
    torch.cuda.synchronize(dev); parallel.barrier()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    loss_end = float(tr.loss.item())
    ms_per_step = dt / a.steps * 1e3
    value = world * a.steps / dt                            # C2-sized steps/s, all ranks

    # ---- sampler (same config): whole EM loop as one hipGraph --------------
    smp = GraphedEMSampler(gen, B_C2, a.sample_steps)
    x0 = gen.latent_sample(B_C2, 2)
    smp.run(x0)
    parallel.barrier(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        xs = smp.run(x0)
    torch.cuda.synchronize(dev); parallel.barrier()
    dts = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    sample_steps_per_s = world * reps * a.sample_steps / dts
    finite = bool(torch.isfinite(xs).all())

    # ---- roofline of the dominant kernel (rank 0) ---------------------------
    roof = None
    if rank == 0:
        import ctypes as C
        lib = ops.lib()
        nsl = C.c_int32(0)
        fn = lambda: ops.check(lib.msgm_mlp_ssm_partial(tr.P, tr.y.data_ptr(), tr.t.data_ptr(), tr.vp.data_ptr(), None, None, B_C2,
                                                        tr.st, tr.inv_batch, None, tr.ws.data_ptr(), tr.ws.numel() * 4,
                                                        C.byref(nsl), ops.stream()), "partial")
        tk = time_kernel_events(fn, 50, dev)
        flop = TRAIN_FLOP_PER_SAMPLE * B_C2
        ach = flop / tk / 1e12
        traffic = None
        pj = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(pj):
            try:
                traffic = json.load(open(pj)).get("k_mlp_train_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roof = {"kernel": "k_mlp<MODE_TRAIN> (msgm_mlp_ssm_partial)", "bound": "mfma", "achieved": ach,
                "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
                "traffic": traffic, "kernel_ms": tk * 1e3, "flop_per_launch": flop}

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        out = {"metric": "score-matching train steps/sec (+ reverse-SDE sample steps/sec)", "value": value,
               "unit": "train_steps/s (B=65536 per step)", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "C2: 2-D Gaussian mixture, MLP score net (d=2, hidden 128), SGM, batch 65536/GPU, "
                                      "SSM loss + Adam", "global_batch": B_C2 * world, "parallelism": f"dp{world}",
                          "graph": world == 1},
               "sample_steps_per_s": sample_steps_per_s, "sample_config": f"EM, {a.sample_steps} steps, 65536 rows/GPU, whole loop = one launch (msgm_mlp_em_loop) replayed as a hipGraph",
               "final_loss": loss_end, "sampler_finite": finite, "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
