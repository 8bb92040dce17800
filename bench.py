#!/usr/bin/env python
"""Benchmark of the DiffuScene DDPM hot path on MI355X (contract: see the task brief / DESIGN.md section 6).

    python bench.py --gpus N --steps K --warmup W [--mode both|sample|train] [--batch 256] [--objects 80]

Workload (BASELINE.json `metric`): uncond living/dining rooms scaled to N=80 objects, B=256 scenes per GPU,
C=65 channels, fp32, synthetic scenes with the real encoders' value distribution, random-init weights.
  mode=sample : a step = one reverse-diffusion denoiser step (Unet1D forward + fused posterior step) of a
                1000-step p_sample_loop, replayed from the captured hipGraph.
  mode=train  : a step = train_on_batch semantics (q_sample, forward, loss incl. IoU, backward, clip(10), Adam).
  mode=both   : (default, BASELINE.json metric "train + 1000-step sample") the K timed denoiser steps are ceil(K/2)
                sampling steps followed by floor(K/2) training steps inside ONE timed region; the two rates are also
                reported separately ("sample", "train").
N > 1: one process per GPU (torchrun), batch sharded by rank (weak scaling: per-GPU batch fixed); sampling
needs no collective, training all-reduces the gradients over RCCL.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32
_T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def unet_forward_flops(B, N):
    """F(B,N) = B*N*(65.01e6 + 512*N) + B*90.2e6 (uncond, C=62/65)."""
    return B * N * (65.01e6 + 512.0 * N) + B * 90.2e6


def build_model(args, device):
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import DiffusionSceneLayout_DDPM
    from oracle import weights as W
    import tempfile
    stats = os.path.join(tempfile.mkdtemp(), "dataset_stats.txt")
    with open(stats, "w") as f:
        json.dump(W.DATASET_STATS, f)
    kw = dict(W.UNCOND_LIVING)
    cfg = {"type": "diffusion_scene_layout_ddpm", "net_type": "unet1d", "point_dim": 65, "latent_dim": 0,
           "room_mask_condition": False, "sample_num_points": args.objects, "objectness_dim": 0, "objfeat_dim": 32,
           "class_dim": 25, "angle_dim": 2, "learnable_embedding": True, "instance_condition": True,
           "instance_emb_dim": 128,
           "diffusion_kwargs": dict(schedule_type="linear", beta_start=1e-4, beta_end=0.02, time_num=1000,
                                    loss_type="mse", model_mean_type="v", model_var_type="fixedsmall",
                                    loss_separate=True, loss_iou=True, train_stats_file=stats),
           "net_kwargs": kw}
    torch.manual_seed(0)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        model = DiffusionSceneLayout_DDPM(26, None, cfg)
    return model.to(device), cfg


def barrier(ws):
    if ws > 1:
        dist.barrier()
    torch.cuda.synchronize()


class SampleRunner:
    """Reverse-diffusion steps replayed from the captured hipGraph (sampler._StepGraph)."""

    def __init__(self, args, model, device):
        from diffuscene_amd.sampler import _StepGraph
        B, N, C = args.batch, args.objects, 65
        cond = model._instance_condition(B, device)
        with torch.no_grad():
            self.g = _StepGraph(model.diffusion.diffusion, model.diffusion.model, (B, N, C), device, cond, None, True)
        log("graph captured")
        self.g.x.normal_()
        self.g.t.fill_(999)

    def run(self, n):
        for _ in range(n):
            self.g.graph.replay()

    def reset(self):
        self.g.t.fill_(999)


class TrainRunner:
    """train_on_batch on a fixed synthetic batch (per-rank shard of the global batch)."""

    def __init__(self, args, model, device, ws):
        from diffuscene_amd.networks import optimizer_factory
        from oracle import weights as W
        B, N = args.batch, args.objects
        rank = dist.get_rank() if ws > 1 else 0
        x = W.synth_scene_batch(B, N, 25, 32, seed=100 + rank).to(device)
        self.sample = {"translations": x[:, :, 0:3].contiguous(), "sizes": x[:, :, 3:6].contiguous(),
                       "angles": x[:, :, 6:8].contiguous(), "class_labels": x[:, :, 8:33].contiguous(),
                       "objfeats_32": x[:, :, 33:65].contiguous(),
                       "room_layout": torch.zeros(B, 1, 64, 64, device=device)}
        self.model = model
        self.opt = optimizer_factory({"optimizer": "Adam", "lr": 2e-4},
                                     filter(lambda p: p.requires_grad, model.parameters()))
        self.tcfg = {"training": {"max_grad_norm": 10}}

    def run(self, n):
        from diffuscene_amd.networks.diffusion_scene_layout_ddpm import train_on_batch
        for _ in range(n):
            train_on_batch(self.model, self.opt, self.sample, self.tcfg)


def timed(ws, fn):
    barrier(ws)
    t0 = time.perf_counter()
    fn()
    barrier(ws)
    return time.perf_counter() - t0


def roofline_dominant_kernel(plan, B, N):
    """Time the dominant kernel -- the fused WS-conv + GroupNorm + SiLU GEMM (gemm_kernel<...,GN=true>, K=512) --
    with HIP events on the launch stream, using the very argument structs of the timed plan."""
    import ctypes as C
    from diffuscene_amd import _lib, ops
    fn = _lib.fn("dsc_gemm_gn_silu_f32")
    if getattr(plan, "n_chains", 0) > 0:
        return roofline_scene_chain(plan, B, N)
    steps = [a for f, a in plan.tiled_steps if f is fn]
    structs = [a[0]._obj for a in steps]          # ctypes.byref(struct) keeps the struct in ._obj
    sel = [(a, s) for a, s in zip(steps, structs) if s.k1 + s.k2 == 512]
    s = ops.stream_ptr()
    for a, _ in sel[:4]:
        fn(*a, s)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    ev0.record()
    for _ in range(reps):
        for a, _ in sel:
            fn(*a, s)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / (reps * len(sel))
    B = plan.B                      # the graph sampler runs the batch as independent half-batch chains
    flops = 2.0 * B * N * 512 * 512
    achieved = flops / (ms * 1e-3) / 1e12
    traffic, traffic_src = None, None
    tfile = os.path.join(ROOT, "profiles", "r01_gemm_gn_hbm_traffic.json")
    if os.path.exists(tfile) and B * N == 20480:
        # HBM bytes per launch of this kernel from rocprofv3 PMC passes (FETCH_SIZE x2 correction + WRITE_SIZE), measured
        # offline on the same workload -- bench.py cannot run the profiler on itself
        with open(tfile) as f:
            traffic = round(json.load(f)["hbm_bytes_per_launch"])
        traffic_src = "profiles/r01_gemm_gn_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
    return {"bound": "mfma", "kernel": "gemm_kernel<GN=true> (WS-conv+GroupNorm+SiLU, M=%d,N=512,K=512)" % (B * N),
            "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "avg_launch_us": round(ms * 1e3, 2),
            "launches_per_step": len(steps), "algorithmic_flops_per_launch": flops,
            "traffic": traffic, "traffic_source": traffic_src}


def roofline_scene_chain(plan, B, N):
    """When the plan runs whole ResnetBlocks as scene-resident chains, that kernel carries most of the FLOPs: time every
    chain launch of the plan with HIP events on the launch stream; algorithmic FLOPs = sum over the chained layers of
    2*M*n*K.  The tiled GN-GEMM of the one-launch-per-layer plan is timed next to it for comparison."""
    from diffuscene_amd import _lib, ops
    chain_f = _lib.fn("dsc_scene_chain_f32")
    gn_f = _lib.fn("dsc_gemm_gn_silu_f32")
    chains = [a for f, a in plan.steps if f is chain_f]
    s = ops.stream_ptr()
    flops, layers = 0.0, 0
    for arr, flags, cnt, _n in chains:
        for j in range(cnt):
            flops += 2.0 * arr[j].m * arr[j].n * (arr[j].k1 + arr[j].k2)
            layers += 1

    def timeit(calls, f, reps=3):
        for a in calls[:2]:
            f(*a, s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            for a in calls:
                f(*a, s)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps          # ms for one pass over `calls`
    ms_chain = timeit(chains, chain_f)
    achieved = flops / (ms_chain * 1e-3) / 1e12
    tiled = [a for f, a in plan.tiled_steps if f is gn_f and a[0]._obj.k1 + a[0]._obj.k2 == 512]
    ms_tiled = timeit(tiled, gn_f) / max(len(tiled), 1)
    tiled_tf = 2.0 * plan.B * N * 512 * 512 / (ms_tiled * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "scene_chain_kernel (one workgroup per scene walks %d fused 512-wide layers in %d launches; "
                                       "WS-conv+GroupNorm+SiLU / res_conv / MLP trunk layers)" % (layers, len(chains)),
            "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "avg_launch_us": round(ms_chain * 1e3 / len(chains), 2),
            "launches_per_step": len(chains), "layers_per_step": layers, "algorithmic_flops_per_step": flops,
            "traffic": None, "traffic_source": None,
            "tiled_gn_gemm": {"avg_launch_us": round(ms_tiled * 1e3, 2), "achieved": round(tiled_tf, 2),
                              "frac": round(tiled_tf / PEAK_FP32_MFMA_TFLOPS, 4)}}


def cpu_baseline(args, mode):
    """The oracle (CPU restatement of the reference path, kind 'port') timed on this box's host cores on a bounded
    sample: a few steps on a slice of the batch, scaled linearly to the full batch (scenes are independent)."""
    from oracle import ref_torch as R
    from oracle import weights as W
    kw = dict(W.UNCOND_LIVING)
    sd = W.synth_state_dict(kw)
    Bs, N = min(args.batch, 64), args.objects
    x = W.synth_scene_batch(Bs, N, 25, 32, seed=0)
    cond = W.synth_condition(Bs, N, 128, 0).contiguous()
    t = torch.full((Bs,), 500, dtype=torch.int64)
    tb = R.schedule_tables(1e-4, 0.02, 1000, "v")
    noise = torch.randn(Bs, N, 65)

    def one_sample_step():
        with torch.no_grad():
            out = R.unet1d_forward(sd, kw, x, t, cond, None)
            return R.p_sample_step(tb, x, t, out, noise, True, "v")

    def one_train_step():
        params = [p.requires_grad_(True) for p in sd.values()]
        lw, _, _ = R.p_losses(tb, lambda xt, tt: R.unet1d_forward(sd, kw, xt, tt, cond, None), x, t, noise,
                              R.dims_from_kwargs(kw), True, True, W.DATASET_STATS)
        lw.mean().backward()
        for p in params:
            p.grad = None

    if mode == "both":
        def fn():
            one_sample_step()
            one_train_step()
    else:
        fn = one_sample_step if mode == "sample" else one_train_step
    # pick the host thread count that runs the oracle fastest on this box (all logical CPUs is rarely it)
    ncpu = os.cpu_count() or 1
    best = None
    for th in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(th)
        fn()
        t1 = time.perf_counter()
        fn()
        d = time.perf_counter() - t1
        log("cpu_baseline: %d threads -> %.3f s per %s step on %d scenes" % (th, d, mode, Bs))
        if best is None or d < best[1]:
            best = (th, d)
        if d > 2.5 * best[1]:
            break                                   # oversubscribed: more threads only get slower
    cores = best[0]
    torch.set_num_threads(cores)
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        el = time.perf_counter() - t0
        if el > 12.0 or n >= 30:
            break
    per_full = (el / n) * (args.batch / Bs) / (2.0 if mode == "both" else 1.0)
    return {"value": round(1.0 / per_full, 4), "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": "%d %s steps of the oracle on %d of %d scenes (N=%d), scaled x%d to the full batch%s"
                      % (n, "sample+train pairs" if mode == "both" else mode, Bs, args.batch, N, args.batch // Bs,
                         "" if mode == "sample" else " (train = fwd+bwd only, no optimizer)")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default=os.environ.get("DSC_BENCH_MODE", "both"), choices=["both", "sample", "train"])
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--objects", type=int, default=80)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    ws = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if ws > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl")
    rank = dist.get_rank() if ws > 1 else 0
    assert ws == args.gpus, "launch with torchrun --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, ws)

    log("building model")
    model, cfg = build_model(args, device)
    log("model on device")
    plan = None
    n_s = {"both": (args.steps + 1) // 2, "sample": args.steps, "train": 0}[args.mode]
    n_t = args.steps - n_s
    sr = SampleRunner(args, model, device) if n_s else None
    tr = TrainRunner(args, model, device, ws) if n_t else None
    if sr:
        sr.run(args.warmup)
        sr.reset()
        plan = sr.g.plan
    if tr:
        tr.run(args.warmup)

    def region():
        if sr:
            sr.run(n_s)
        if tr:
            tr.run(n_t)

    dt = timed(ws, region)
    log("timed region done: %.3f s for %d steps (%d sample + %d train)" % (dt, args.steps, n_s, n_t))
    parts = {}
    if args.mode == "both":          # the two rates separately (outside the headline region)
        sr.reset()
        parts["sample"] = timed(ws, lambda: sr.run(n_s)) / n_s
        parts["train"] = timed(ws, lambda: tr.run(n_t)) / max(n_t, 1)
    tmax = torch.tensor([dt] + [parts.get(k, 0.0) for k in ("sample", "train")], device=device, dtype=torch.float64)
    if ws > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax[0].item())
    if parts:
        parts = {"sample": float(tmax[1].item()), "train": float(tmax[2].item())}

    if rank == 0:
        B, N = args.batch, args.objects
        steps_per_s = args.steps * ws / dt        # whole job: every rank advances its own B scenes one step
        out = {
            "metric": "denoiser steps/sec (%s) at B=256, N=80 objects" % (
                {"sample": "1000-step sample loop", "train": "train step", "both": "train + 1000-step sample"}[args.mode]),
            "value": round(steps_per_s, 3), "unit": "steps/s", "n_gpus": ws, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "uncond living/dining rooms (config/uncond/diffusion_livingrooms_instancond_lat32_v.yaml "
                                   "scaled to N=%d), B=%d scenes per GPU, C=65, T=1000, mode=%s" % (N, B, args.mode),
                       "global_batch": B * ws, "parallelism": "dp%d" % ws if ws > 1 else "single"},
        }
        F = unet_forward_flops(B, N)
        out["model_tflops"] = round(F * (n_s + 3.0 * n_t) / dt / 1e12, 2)        # per GPU, algorithmic (train = 3F)
        for k, v in parts.items():
            out[k] = {"steps_per_s": round(ws / v, 3), "ms_per_step": round(v * 1e3, 3),
                      "tflops_per_gpu": round((1.0 if k == "sample" else 3.0) * F / v / 1e12, 2)}
        out["model_frac_of_fp32_mfma_peak"] = round(out["model_tflops"] / PEAK_FP32_MFMA_TFLOPS, 4)
        if plan is None:
            with torch.no_grad():
                plan = model.diffusion.model.engine(device).prepare(B, N, model._instance_condition(B, device), None)
                plan.x_in.normal_(); plan.t_in.fill_(500); plan.run()
        out["roofline"] = roofline_dominant_kernel(plan, B, N)
        log("roofline done")
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.mode)
        print(json.dumps(out))
    if ws > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
