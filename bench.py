#!/usr/bin/env python
"""Benchmark of the DiffuScene DDPM hot path on MI355X (contract: see the task brief / DESIGN.md section 6).

    python bench.py --gpus N --steps K --warmup W [--config living80|bedroom21|text|complete|arrange]
                    [--mode both|sample|train] [--no-cpu-baseline] [--no-full-loop]

Workloads = the five BASELINE.json configs (`--config`, default = the headline one):
  living80   uncond living/dining rooms scaled to N=80, B=256 scenes per GPU, C=65            (BASELINE configs[2], metric)
  bedroom21  uncond bedrooms scaled to N=21, B=256, C=62                                       (configs[1])
  text       text-conditioned bedrooms, B=128, N=12, L=32 cross-attention tokens               (configs[3])
  complete   scene completion (masked p_sample_loop), living N=80, B=128, P=20 given objects   (configs[4])
  arrange    re-arrangement (5-channel model, 512-d instance+arrange condition), N=80, B=128   (configs[4])
All fp32, synthetic scenes with the real encoders' value distribution, random-init weights.
  mode=sample : a step = one reverse-diffusion denoiser step (Unet1D forward + fused posterior step, plus the in-painting
                overwrite for `complete`) of a 1000-step loop, replayed from the captured hipGraph.
  mode=train  : a step = train_on_batch semantics (q_sample, forward, loss incl. IoU, backward, clip(10), Adam).
  mode=both   : (default, BASELINE.json metric "train + 1000-step sample") the K timed denoiser steps are ceil(K/2)
                sampling steps followed by floor(K/2) training steps inside ONE timed region; the two rates are also
                reported separately ("sample", "train").  `full_loop` adds the wall time of one whole 1000-step
                p_sample_loop(graph=True) (outside the timed region).
`--scaling strong` treats the config's batch as the GLOBAL batch (split over the ranks) instead of the per-GPU batch.
N > 1: one process per GPU; `python bench.py --gpus N` spawns the ranks itself (torch.distributed.run, 127.0.0.1) when it
is not already running under torchrun.  Batch sharded by rank (weak scaling: per-GPU batch fixed); sampling needs no
collective, training all-reduces the flat gradient buffer over RCCL.  Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import io
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA
SPLIT_PRODUCTS = 6                 # split-bf16 arithmetic: 6 bf16 MFMA products per f32 product (csrc/gemm_split.hip)
DTYPE_SPLIT = "f32 (operands split 3xbf16, 6 products, f32 accumulate)"
_T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------ workloads
CONFIGS = {
    "living80": dict(batch=256, objects=80, kw="UNCOND_LIVING", class_dim=25, kind="uncond",
                     yaml="config/uncond/diffusion_livingrooms_instancond_lat32_v.yaml scaled to N=80",
                     title="uncond living/dining rooms"),
    "bedroom21": dict(batch=256, objects=21, kw="UNCOND_BEDROOM", class_dim=22, kind="uncond",
                      yaml="config/uncond/diffusion_bedrooms_instancond_lat32_v.yaml scaled to N=21",
                      title="uncond bedrooms"),
    "text": dict(batch=128, objects=12, kw="TEXT_BEDROOM", class_dim=22, kind="text", text_len=32,
                 yaml="config/text/diffusion_bedrooms_instancond_lat32_v_bert.yaml (cached BERT features, L=32)",
                 title="text-conditioned bedrooms"),
    "complete": dict(batch=128, objects=80, kw="UNCOND_LIVING", class_dim=25, kind="complete", partial=20,
                     yaml="config/uncond/diffusion_livingrooms_instancond_lat32_v.yaml scaled to N=80, "
                          "p_sample_loop_complete with 20 given objects",
                     title="scene completion, living rooms"),
    "arrange": dict(batch=128, objects=80, kw="REARRANGE_LIVING", class_dim=25, kind="arrange",
                    yaml="config/rearrange/diffusion_livingrooms_instancond_lat32_v_rearrange.yaml scaled to N=80",
                    title="scene re-arrangement, living rooms"),
}


def forward_flops(kind, B, N, L=0):
    """Algorithmic FLOPs of one Unet1D forward (FlopCounterMode probes of the reference)."""
    if kind == "arrange":
        return B * N * (58.86e6 + 512.0 * N) + B * 90.2e6          # 6.147e11 @ (128, 80), 1.698e11 @ (128, 21)
    f = B * N * (65.01e6 + 512.0 * N) + B * 90.2e6                   # uncond, C=62/65
    if kind == "text":
        f += 2.43e6 * B * N + 2.43e6 * B * L                         # 9 cross-attention layers
    return f


def build_model(spec, device, time_num=1000):
    import torch
    from diffuscene_amd.flat import ensure_flat
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import DiffusionSceneLayout_DDPM
    from diffuscene_amd import workloads as W          # (the product's benchmark never needs oracle/: only the cpu_baseline leg does)
    stats = os.path.join(tempfile.mkdtemp(), "dataset_stats.txt")
    with open(stats, "w") as f:
        json.dump(W.DATASET_STATS, f)
    kw = dict(getattr(W, spec["kw"]))
    nc = spec["class_dim"]
    cfg = {"type": "diffusion_scene_layout_ddpm", "net_type": "unet1d", "point_dim": 8 + nc + 32, "latent_dim": 0,
           "room_mask_condition": False, "sample_num_points": spec["objects"], "objectness_dim": 0, "objfeat_dim": 32,
           "class_dim": nc, "angle_dim": 2, "learnable_embedding": True, "instance_condition": True,
           "instance_emb_dim": 128,
           "diffusion_kwargs": dict(schedule_type="linear", beta_start=1e-4, beta_end=0.02, time_num=time_num,
                                    loss_type="mse", model_mean_type="v", model_var_type="fixedsmall",
                                    loss_separate=True, loss_iou=True, train_stats_file=stats),
           "net_kwargs": kw}
    if spec["kind"] == "text":
        # BERT weights cannot be downloaded here: the model consumes cached last_hidden_state features (B, L, 768), the
        # trainable fc_text_f (768 -> 512) stays inside the step (diffusion_scene_layout_ddpm.py:210-221)
        cfg.update(text_condition=True, text_embed_dim=512, text_bert_cached=True)
    if spec["kind"] == "arrange":
        cfg.update(room_arrange_condition=True, arrange_emb_dim=384)
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = DiffusionSceneLayout_DDPM(nc + 1, None, cfg)
    model = model.to(device)
    # The first training step re-homes every parameter into the flat buffer P (flat.FlatStorage) and frees the 442 original
    # storages.  Do it NOW, before any sampling graph bakes parameter pointers into its launches: rounds 1-4 built the SampleRunner
    # first, so every sampling replay after the first train_on_batch -- the timed region included -- read FREED parameter storages
    # (same kernels and timings, wrong weights), and once torch.cuda.graph's empty_cache() had returned those blocks to the driver a
    # launch from the stale plan could fault: the `Memory access fault by GPU` of round 4's arrange side line.
    ensure_flat(model)
    return model, cfg


def synth_batch(spec, device, seed):
    import torch
    from diffuscene_amd import workloads as W
    B, N, nc = spec["batch"], spec["objects"], spec["class_dim"]
    x = W.synth_scene_batch(B, N, nc, 32, seed=seed).to(device)
    sample = {"translations": x[:, :, 0:3].contiguous(), "sizes": x[:, :, 3:6].contiguous(),
              "angles": x[:, :, 6:8].contiguous(), "class_labels": x[:, :, 8:8 + nc].contiguous(),
              "objfeats_32": x[:, :, 8 + nc:].contiguous(), "room_layout": torch.zeros(B, 1, 64, 64, device=device)}
    if spec["kind"] == "text":
        g = torch.Generator().manual_seed(seed)
        sample["desc_bert"] = torch.randn(B, spec["text_len"], 768, generator=g).to(device)
        sample["description"] = ["synthetic"] * B
    return x, sample


def barrier(ws):
    import torch
    import torch.distributed as dist
    if ws > 1:
        dist.barrier()
    torch.cuda.synchronize()


def sampling_inputs(spec, model, device, seed):
    """-> (shape of the diffused tensor, condition, cross condition, given objects of a completion or None, the scene batch) -- what
    the wrapper's generate_layout / complete_scene / arrange_scene hand to the reverse loop for this configuration."""
    import torch
    B, N = spec["batch"], spec["objects"]
    kind = spec["kind"]
    x, sample = synth_batch(spec, device, seed)
    C = x.shape[-1]
    partial = None
    with torch.no_grad():
        cond = model._instance_condition(B, device)
        cross = None
        if kind == "text":
            cross = model._text_condition(sample["description"], None, device, desc_bert=sample["desc_bert"])
        if kind == "arrange":
            cond = torch.cat([cond, model.fc_arrange_condition(model._arrange_input(x))], dim=-1).contiguous()
            C = model.translation_dim + model.angle_dim
        if kind == "complete":
            partial = x[:, :spec["partial"], :].contiguous()
    return (B, N, C), cond, cross, partial, x


class SampleRunner:
    """Reverse-diffusion steps replayed from the captured hipGraph (sampler._StepGraph)."""

    def __init__(self, spec, model, device, seed):
        import torch
        from diffuscene_amd.sampler import _StepGraph
        self.shape, cond, cross, self.partial, _ = sampling_inputs(spec, model, device, seed)
        pshape = tuple(self.partial.shape) if self.partial is not None else None
        with torch.no_grad():
            self.g = _StepGraph(model.diffusion.diffusion, model.diffusion.model, self.shape, device, cond, cross, True,
                                partial_shape=pshape)
        log("graph captured")
        self.reset()

    def run(self, n):
        self.g.replay_steps(n)              # refuses a graph whose parameter pointers are stale (sampler._StepGraph.check_current)

    def sync_weights(self):
        """After optimizer steps (weights updated in place): re-derive the standardised / packed / split weights and the per-timestep
        table the captured launches read -- outside every timed region; the graph itself stays valid (same buffers)."""
        import torch
        eng = self.g.plan.eng
        with torch.no_grad():
            eng.refresh()
            self.g.check_current()
            if self.g.plan.time_table:
                eng.ss_table()
            for p in self.g.plans:
                p.run_pre()

    def reset(self):
        self.g.x.normal_()
        self.g.t.fill_(999)
        if self.partial is not None:
            self.g.partial.copy_(self.partial)

    def full_loop(self):
        """wall time of one whole 1000-step reverse loop through the graph path (x_T draw + 1000 replays + result copy)."""
        import torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x_T = torch.randn(self.shape, device=self.g.x.device)
        out = self.g.run(x_T, 1000, partial=self.partial)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert bool(torch.isfinite(out).all()), "full loop produced non-finite values"
        return dt


class TrainRunner:
    """train_on_batch on a fixed synthetic batch (per-rank shard of the global batch)."""

    def __init__(self, spec, model, device, rank):
        from diffuscene_amd.networks import optimizer_factory
        _, self.sample = synth_batch(spec, device, seed=100 + rank)
        self.model = model
        self.opt = optimizer_factory({"optimizer": "Adam", "lr": 2e-4},
                                     filter(lambda p: p.requires_grad, model.parameters()))
        self.tcfg = {"training": {"max_grad_norm": 10}}

    def run(self, n):
        from diffuscene_amd.networks.diffusion_scene_layout_ddpm import train_on_batch
        for _ in range(n):
            train_on_batch(self.model, self.opt, self.sample, self.tcfg)


class phase:
    """`with phase("name", ws):` -- first-run safety of the multi-GPU line (no 8-GPU node was ever available to the builder).  Every rank
    says which phase it enters / leaves on stderr; an exception names the rank and the phase before it propagates (torchrun then
    ends the other ranks); a phase that exceeds DSC_BENCH_PHASE_TIMEOUT seconds (default 300) dumps every thread's Python stack of
    the hung rank and exits with code 3 instead of hanging the node until the driver's limit."""
    current = "start"

    def __init__(self, name, ws):
        self.name, self.ws = name, ws
        self.rank = int(os.environ.get("RANK", "0"))

    def __enter__(self):
        import faulthandler
        phase.current = self.name
        if self.ws > 1:
            print("[bench rank %d %7.1fs] phase %s ..." % (self.rank, time.perf_counter() - _T0, self.name), file=sys.stderr, flush=True)
            faulthandler.dump_traceback_later(float(os.environ.get("DSC_BENCH_PHASE_TIMEOUT", "300")), exit=True)
        return self

    def __exit__(self, et, ev, tb):
        import faulthandler
        if self.ws > 1:
            faulthandler.cancel_dump_traceback_later()
            if et is not None:
                print("[bench rank %d] FAILED in phase %s: %s: %s" % (self.rank, self.name, et.__name__, ev), file=sys.stderr, flush=True)
            else:
                print("[bench rank %d %7.1fs] phase %s ok" % (self.rank, time.perf_counter() - _T0, self.name), file=sys.stderr, flush=True)
        return False


def all_ranks_ok(ok, device, ws):
    """True when `ok` holds on every rank (one MIN all-reduce of a flag): ranks must agree before they change what they run next."""
    import torch
    import torch.distributed as dist
    if ws == 1:
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def compare_ddp_schedules(spec, model, device, rank, ws, steps=6, warm=3):
    """N > 1: time the training step under each data-parallel flush schedule (train_step.DDP_FLUSH_SCHEDULES) in THIS run, so that one
    driver invocation compares them on the node -> ({schedule: {...}}, fastest schedule or None).  A schedule that fails on any rank
    is recorded with its error and skipped by all ranks together; when the hipGraph segments of a schedule cannot be captured the
    step runs the same launches eagerly (train_step._capture) and the row says so."""
    from diffuscene_amd.train_step import DDP_FLUSH_SCHEDULES
    rows, best = {}, None
    prev = os.environ.get("DSC_DDP_FLUSH")
    for flush in DDP_FLUSH_SCHEDULES:
        os.environ["DSC_DDP_FLUSH"] = flush
        err, ms, seg = None, None, None
        try:
            with phase("schedule:%s" % flush, ws):
                tr = TrainRunner(spec, model, device, rank)
                tr.run(warm)                      # eager step, capture, replay
                ms = timed(ws, lambda: tr.run(steps)) / steps * 1e3
                ent = next(reversed(model._dsc_plan_runner.plans.values()))
                seg = len(ent["graph"].segments) if ent.get("graph") is not None else 0
        except Exception as e:                     # noqa: BLE001 -- recorded; the ranks agree below on whether the schedule counts
            err = "%s: %s" % (type(e).__name__, e)
        ok = all_ranks_ok(err is None, device, ws)
        if ok:
            import torch
            import torch.distributed as dist
            t = torch.tensor([ms], device=device, dtype=torch.float64)
            if ws > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms = float(t.item())
            rows[flush] = {"ms_per_step": round(ms, 3), "graph_segments": seg,
                           "captured": bool(seg), "steps": steps}
            if best is None or ms < rows[best]["ms_per_step"]:
                best = flush
        else:
            rows[flush] = {"error": err or "failed on another rank"}
        log("ddp schedule %s: %s" % (flush, rows[flush]))
    if prev is None:
        os.environ.pop("DSC_DDP_FLUSH", None)
    else:
        os.environ["DSC_DDP_FLUSH"] = prev
    return rows, best


def timed(ws, fn):
    barrier(ws)
    t0 = time.perf_counter()
    fn()
    barrier(ws)
    return time.perf_counter() - t0


def git_head():
    try:
        return subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, stderr=subprocess.DEVNULL,
                                       text=True).strip()
    except Exception:
        pass
    try:                                   # the GPU box gets a snapshot without .git: tools/gpu_round.sh leaves the head here
        with open(os.path.join(ROOT, "GIT_HEAD")) as f:
            return f.read().strip() or None
    except OSError:
        return None


def csrc_sha():
    """Fingerprint of the kernel sources (what a PMC traffic file must have been measured on to be quoted)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "diffuscene_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def roofline_dominant_kernel(plan, N, config_name):
    """Time the dominant kernel -- the fused WS-conv + GroupNorm + SiLU GEMM (Block.forward in one launch) -- with HIP events on the
    launch stream, using the very argument structs of the timed plan.  `frac` is quoted on the K=512 launches (47 of the 56 per
    forward), `frac_mix` over all GN launches of a step.  With the split-bf16 arithmetic the kernel EXECUTES 6 bf16 MFMA products per
    f32 product: achieved / peak are executed TFLOP/s against the dense bf16 peak; the algorithmic (f32-equivalent) rate stands
    beside it."""
    import torch
    from diffuscene_amd import _lib, ops
    plan.eng.params_moved()
    plan.check_current()                          # the structs hold raw parameter pointers: never launch from a stale plan
    fn = _lib.fn("dsc_gemm_gn_silu_f32")
    steps = [a for f, a in plan.tiled_steps if f is fn]
    structs = [a[0]._obj for a in steps]          # ctypes.byref(struct) keeps the struct in ._obj
    s = ops.stream_ptr()

    def timed_launches(sel):
        for _ in range(3):                 # the clocks take milliseconds to ramp after an idle sync: warm up, then time
            for a in sel:
                fn(*a, s)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        ev0.record()
        for _ in range(reps):
            for a in sel:
                fn(*a, s)
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / (reps * len(sel))

    sel512 = [a for a, st in zip(steps, structs) if st.k1 + st.k2 == 512]
    ms = timed_launches(sel512)
    ms_mix = timed_launches(steps)
    M = plan.B * N
    flops = 2.0 * M * 512 * 512
    flops_mix = sum(2.0 * M * st.n * (st.k1 + st.k2) for st in structs) / len(structs)
    split = _lib.fn("dsc_gemm_arithmetic")(sel512[0][0], 1) == 1       # the library's own dispatch decision for this launch
    mult, peak = (SPLIT_PRODUCTS, PEAK_BF16_MFMA_TFLOPS) if split else (1, PEAK_FP32_MFMA_TFLOPS)
    alg = flops / (ms * 1e-3) / 1e12
    alg_mix = flops_mix / (ms_mix * 1e-3) / 1e12
    tile = _lib.fn("dsc_gemm_split_tile")(sel512[0][0], 1)             # ... and which kernel family / tile it picks
    wave = tile in (_lib.TILE_WAVE_GN, _lib.TILE_WAVE_DENSE, _lib.TILE_WAVE_GN_64)
    kname = (("dsc_wave::gemm_split_wave_kernel<GN=true> (wave-autonomous, tile %d; " % tile if wave else "dsc_split::gemm_split_kernel<GN=true> (block-staged, tile %d; " % tile)
             + "WS-conv+GroupNorm+SiLU, M=%d,N=512,K=512; 3xbf16 split, 6 MFMA products)" % M
             if split else "dsc_gemm::gemm_kernel<GN=true> (WS-conv+GroupNorm+SiLU, M=%d,N=512,K=512; f32 MFMA)" % M)
    # HBM bytes per launch of this kernel from rocprofv3 PMC passes (FETCH_SIZE x2 correction + WRITE_SIZE, separate passes:
    # tools/gpu_round.sh pmc + tools/pmc_summary.py on the same workload -- bench.py cannot run the profiler on itself).  The file
    # is only quoted when it was measured on THESE kernel sources (csrc fingerprint) and for the arithmetic that is running.
    traffic, traffic_src = None, None
    here = csrc_sha()
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_gn_hbm_traffic*.json")), reverse=True) + \
        sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "r*_gemm_gn_hbm_traffic*.json")), reverse=True)   # same gpurun call: pmc ran first
    for path in cands:
        with open(path) as f:
            rec = json.load(f)
        if (M != 20480 or rec.get("csrc_sha") != here or "hbm_bytes_per_launch" not in rec
                or ("gemm_split" in (rec.get("kernel") or "")) != split or (split and ("gemm_split_wave" in (rec.get("kernel") or "")) != wave)):
            continue
        traffic = round(rec["hbm_bytes_per_launch"])
        traffic_src = "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; csrc %s, git %s; per-launch average over the K=512 " \
                      "and K=1024 launches of a forward)" % (os.path.relpath(path, ROOT), here, rec.get("git_head", "?"))
        break
    if traffic is None:
        traffic_src = "no PMC file measured on these kernel sources (csrc %s): run tools/gpu_round.sh <tag> pmc" % here
    out = {"bound": "mfma", "kernel": kname,
           "achieved": round(alg * mult, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(alg * mult / peak, 4),
           "avg_launch_us": round(ms * 1e3, 2), "launches_per_step": len(steps), "algorithmic_flops_per_launch": flops,
           "algorithmic_tflops": round(alg, 2),
           "frac_mix": round(alg_mix * mult / peak, 4), "avg_launch_us_mix": round(ms_mix * 1e3, 2),
           "mix": "%d launches with K=512, %d with K=1024" % (len(sel512), len(steps) - len(sel512)),
           "traffic": traffic, "traffic_source": traffic_src}
    if split:
        out["executed_flops_per_launch"] = flops * mult
        out["note"] = ("achieved = executed bf16-MFMA flops (6 x algorithmic) / time against the dense bf16 peak; algorithmic_tflops is the "
                       "f32-equivalent rate (the exact-f32 MFMA kernel -- DSC_GEMM=f32 / dsc_set_gemm_arithmetic(0) -- peaks at %.1f)" % PEAK_FP32_MFMA_TFLOPS)
    return out


def plan_executed_flops(plan):
    """FLOPs one run of the plan's per-step launch list EXECUTES (algorithmic, f32-equivalent): its GEMM launches 2 m n k (x batch) from
    the plan's own argument structs plus the attention cores.  For a sampling plan this is less than the reference's forward
    (forward_flops): the time MLP is tabulated once per weight version and the conditioning-only launches (context MLPs, text K/V) run
    once per reverse loop (Plan.pre_steps), not per step."""
    f = 0.0
    for _, a in plan.gemm_args():
        f += 2.0 * a.m * a.n * (a.k1 + a.k2) * max(a.batch, 1)
    from diffuscene_amd import _lib
    la, at, gl = _lib.fn("dsc_linear_attention_f32"), _lib.fn("dsc_attention_f32"), _lib.fn("dsc_gemm_layernorm_f32")
    for fn_, a in plan.steps:
        if fn_ is la:                      # (..., scenes, nq, nk, scale): context 2 nk 32 32 + output 2 nq 32 32 per head
            scenes, nq, nk = a[8], a[9], a[10]
            f += scenes * 4 * (2.0 * nk * 32 * 32 + 2.0 * nq * 32 * 32)
        elif fn_ is at:                    # (..., scenes, n, scale): QK^T and PV
            scenes, n = a[8], a[9]
            f += scenes * 4 * (4.0 * n * n * 32)
        elif fn_ is gl:
            g = a[0]._obj
            f += 2.0 * g.m * g.n * (g.k1 + g.k2)
    return f


def dtype_label():
    """The arithmetic the path computes in, from the library's own switch (dsc_get_gemm_arithmetic) -- not from the environment."""
    from diffuscene_amd import _lib
    return DTYPE_SPLIT if _lib.split_enabled() else "f32 (exact f32 MFMA)"


def split_flop_fraction(plan):
    """Fraction of the GEMM flops of one forward that run on the split-bf16 kernel, by the library's own per-launch decision."""
    from diffuscene_amd import ops
    tot = on = 0.0
    for kind, a in plan.gemm_args():
        f = 2.0 * a.m * a.n * (a.k1 + a.k2) * max(a.batch, 1)
        tot += f
        if ops.gemm_uses_split(a, gn=(kind == "gn")):
            on += f
    return on / tot if tot else 0.0


# ------------------------------------------------------------------------------------------ CPU baseline
def _cpu_info():
    model, phys = None, None
    try:
        cores = set()
        pid = cid = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and model is None:
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    pid = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    cid = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if pid is not None and cid is not None:
                        cores.add((pid, cid))
                    pid = cid = None
        phys = len(cores) or None
    except OSError:
        pass
    return model, phys


def spec_name(spec):
    return next(k for k, v in CONFIGS.items() if v["title"] == spec["title"] and v["kind"] == spec["kind"])


def cpu_slice_probe(arg, config):
    """Child of cpu_baseline's all-cores leg: `THREADS:SCENES` -> one JSON line with the seconds of the second sampling-step call of the
    oracle on that many scenes at that many threads (the parent enforces the time limit)."""
    import torch
    from oracle import ref_torch as R
    from oracle import weights as W
    th, n = (int(x) for x in arg.split(":"))
    spec = dict(CONFIGS[config])
    kw = dict(getattr(W, spec["kw"]))
    sd = W.synth_state_dict(kw)
    N, nc, kind = spec["objects"], spec["class_dim"], spec["kind"]
    x = W.synth_scene_batch(n, N, nc, 32, seed=0)
    tb = R.schedule_tables(1e-4, 0.02, 1000, "v")
    cond = W.synth_condition(n, N, 128, 0).contiguous()
    cross = W.synth_text_condition(n, spec.get("text_len", 32), 512, 0) if kind == "text" else None
    if kind == "arrange":
        cond = torch.cat([cond, torch.randn(n, N, 384)], dim=-1)
        x = torch.cat([x[:, :, 0:3], x[:, :, 6:8]], dim=-1).contiguous()
    t = torch.randint(0, 1000, (n,), generator=torch.Generator().manual_seed(1))
    noise = torch.randn(x.shape, generator=torch.Generator().manual_seed(2))
    torch.set_num_threads(th)
    out = None
    for _ in range(2):
        t1 = time.perf_counter()
        with torch.no_grad():
            o = R.unet1d_forward(sd, kw, x, t, cond, cross)
            R.p_sample_step(tb, x, t, o, noise, True, "v")
        out = time.perf_counter() - t1
    print(json.dumps({"threads": th, "scenes": n, "seconds": out}), flush=True)


def cpu_baseline(spec, mode, sweep=False):
    """The oracle (CPU restatement of the reference path, kind 'port', pinned against the real reference by
    tests/test_oracle.py) timed on this box's host cores: FULL-batch steps of the same workload -- sampling step =
    Unet1D forward + posterior step; training step = q_sample, forward, p_losses incl. IoU, backward,
    clip_grad_norm_(10), torch.optim.Adam.step() (diffusion_scene_layout_ddpm.py:456-473)."""
    import torch
    from oracle import ref_torch as R
    from oracle import weights as W
    kw = dict(getattr(W, spec["kw"]))
    sd = W.synth_state_dict(kw)
    B, N, nc, kind = spec["batch"], spec["objects"], spec["class_dim"], spec["kind"]
    x_full = W.synth_scene_batch(B, N, nc, 32, seed=0)
    tb = R.schedule_tables(1e-4, 0.02, 1000, "v")
    cond = W.synth_condition(B, N, 128, 0).contiguous()
    cross = W.synth_text_condition(B, spec.get("text_len", 32), 512, 0) if kind == "text" else None
    if kind == "arrange":
        cond = torch.cat([cond, torch.randn(B, N, 384)], dim=-1)
        x = torch.cat([x_full[:, :, 0:3], x_full[:, :, 6:8]], dim=-1).contiguous()
    else:
        x = x_full
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(1))
    noise = torch.randn(x.shape, generator=torch.Generator().manual_seed(2))
    P = spec.get("partial", 0)
    pnoise = torch.randn(B, P, x.shape[-1]) if P else None
    params = [p.requires_grad_(True) for p in sd.values()]
    opt = torch.optim.Adam(params, lr=2e-4)
    dims = R.dims_from_kwargs(kw)

    def sample_step(sl=slice(None)):
        with torch.no_grad():
            xt = x[sl].clone()
            if P:       # p_sample_loop_complete: re-noise the given objects, overwrite, then the model call (:461-466)
                a = tb["sqrt_alphas_cumprod"][t[sl]].view(-1, 1, 1)
                b = tb["sqrt_one_minus_alphas_cumprod"][t[sl]].view(-1, 1, 1)
                xt[:, :P] = a * x[sl][:, :P] + b * pnoise[sl]
            out = R.unet1d_forward(sd, kw, xt, t[sl], cond[sl], None if cross is None else cross[sl])
            return R.p_sample_step(tb, xt, t[sl], out, noise[sl], True, "v")

    def train_step(sl=slice(None)):
        opt.zero_grad()
        if kind == "arrange":
            lw, _ = R.p_losses_arrange(tb, lambda xt, tt: R.unet1d_forward(sd, kw, xt, tt, cond[sl], None), x[sl], t[sl],
                                       noise[sl])
        else:
            lw, _, _ = R.p_losses(tb, lambda xt, tt: R.unet1d_forward(sd, kw, xt, tt, cond[sl],
                                                                      None if cross is None else cross[sl]),
                                  x[sl], t[sl], noise[sl], dims, True, True, W.DATASET_STATS)
        lw.mean().backward()
        torch.nn.utils.clip_grad_norm_(params, 10)
        opt.step()

    legs = {"sample": [sample_step], "train": [train_step], "both": [sample_step, train_step]}[mode]
    # Host thread count.  Round 4 picked it with a short sweep on a QUARTER batch, which made the reported rate move 0.16-1.7 steps/s
    # between runs: the quarter batch peaks at 16 threads with 0.28 s per sampling step, while the FULL-batch step -- the batch the metric
    # is quoted on -- takes 4.5-5.9 s on the same box at any count from 8 to 32 and 10.8 s at 64 (a full-batch sweep per step kind,
    # profiles/r05_cpu_baseline_sweep.txt: sampling 5.9 / 5.7 / 6.1 / 10.8 s, training 13.2 / 12.8 / 15.9 / 26.7 s at 8 / 16 / 32 / 64
    # threads).  The optimum is flat and the same for both step kinds, so the count is FIXED at 16 (DSC_CPU_BASELINE_THREADS overrides;
    # `--cpu-sweep` repeats the full-batch sweep, ~5 minutes) and the bounded sample is a handful of full-batch steps (~1 minute).
    ncpu = os.cpu_count() or 1
    times, threads_of, sweep_log = {}, {}, {}
    fixed = min(int(os.environ.get("DSC_CPU_BASELINE_THREADS", "16")), ncpu)
    for f in legs:
        best = (fixed, None)
        if sweep:
            best = None
            for th in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
                torch.set_num_threads(th)
                f()                                 # warm-up at this thread count
                d = None
                for _ in range(2 if f is sample_step else 1):
                    t1 = time.perf_counter()
                    f()
                    e = time.perf_counter() - t1
                    d = e if d is None else min(d, e)
                sweep_log.setdefault(f.__name__, {})[th] = round(d, 4)
                log("cpu_baseline: %s, %d threads -> %.3f s at the full batch" % (f.__name__, th, d))
                if best is None or d < best[1]:
                    best = (th, d)
                if d > 1.5 * best[1]:
                    break                           # past the optimum: more threads only get slower
        threads_of[f.__name__] = best[0]
        torch.set_num_threads(best[0])
        f()                                         # warm-up at full batch
        per_step, t0 = [], time.perf_counter()
        n_min = 3 if f is sample_step else 2
        while True:
            t1 = time.perf_counter()
            f()
            per_step.append(time.perf_counter() - t1)
            if len(per_step) >= n_min and (time.perf_counter() - t0 > 12.0 or len(per_step) >= 20):
                break
        per_step.sort()
        times[f.__name__] = (per_step[len(per_step) // 2], len(per_step))          # median of the timed full-batch steps
        log("cpu_baseline: %s, %d threads: median %.3f s over %d full-batch steps" % (f.__name__, best[0], times[f.__name__][0], len(per_step)))
    # SURVEY 8d prescribes set_num_threads(os.cpu_count()): that setting too -- on a BOUNDED slice (B / 16 scenes, sampling step), in a CHILD
    # process under a hard time limit.  On the 256-thread hosts of this pool the full-batch step took 147 s at all threads and even the
    # 16-scene slice 134 s (0.08 s at 16 threads): oversubscribed memory-bound ops; an in-process call cannot be interrupted.
    all_cores = None
    if not sweep and os.environ.get("DSC_CPU_BASELINE_ALL_CORES", "1") != "0" and ncpu > fixed and sample_step in legs:
        n_slice, limit = max(B // 16, 1), 25
        t1 = time.perf_counter()
        sample_step(slice(0, n_slice))
        sample_step(slice(0, n_slice))
        t_fixed = (time.perf_counter() - t1) / 2
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-slice-probe", "%d:%d" % (ncpu, n_slice), "--config", spec_name(spec)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=limit + 20,
                               env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")})
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            t_all = json.loads(line[-1])["seconds"] if line else None
            how = "second call in a child process" if t_all is not None else "child failed: %s" % (r.stderr or "").strip()[-200:]
        except subprocess.TimeoutExpired:
            t_all, how = None, "child killed after %d s (model build + two calls): slower than %.0fx the %d-thread time" % (limit + 20, limit / max(t_fixed, 1e-9), fixed)
        all_cores = {"threads": ncpu, "what": "sampling step on a slice of %d scenes (B / 16); %s" % (n_slice, how),
                     "seconds_all_threads": None if t_all is None else round(t_all, 3), "seconds_at_%d_threads" % fixed: round(t_fixed, 3),
                     "all_over_fixed": None if t_all is None else round(t_all / t_fixed, 2),
                     "note": "os.cpu_count() threads is the slower setting on this host; the headline value uses the measured optimum"}
        log("cpu_baseline: sampling slice of %d scenes: %.3f s at %d threads, %s at ALL %d threads" % (
            n_slice, t_fixed, fixed, "%.3f s" % t_all if t_all is not None else "over the limit", ncpu))
    threads = max(threads_of.values())
    per = sum(v[0] for v in times.values()) / len(times)            # 'both': mean of the two step kinds (1:1 mix)
    model, phys = _cpu_info()
    out = {"value": round(1.0 / per, 4), "unit": "steps/s", "cores": threads, "kind": "port",
           "why_port": "the reference is Python and, by the rules of this build, cannot travel to the GPU box in any form (source, bytecode "
                       "or otherwise; /root/reference does not exist there), so its own modules cannot be timed here; the port is the same "
                       "PyTorch-CPU ops in the same order (oracle/ref_torch.py), pinned to the real modules as below",
           "all_cores": all_cores,
           "statistic": "median of >= 3 (sampling) / >= 2 (training) full-batch steps after a warm-up step, at a fixed thread count (the "
                        "full-batch optimum is flat from 8 to 32 threads for both step kinds: profiles/r05_cpu_baseline_sweep.txt)",
           "threads_per_step_kind": {k.replace("_step", ""): v for k, v in threads_of.items()},
           "thread_sweep_seconds_full_batch": {k.replace("_step", ""): v for k, v in sweep_log.items()} or None,
           "pinned_by": "tests/test_oracle.py (the port vs the real reference modules, <= 2e-5; schedule tables bit-exact) and "
                        "tests/golden/*.npz (outputs of the real reference, regenerated by oracle/make_golden*.py)",
           "sample": "full-batch oracle steps (B=%d, N=%d), median: %s; train = q_sample + fwd + p_losses(IoU) + bwd + "
                     "clip_grad_norm_(10) + Adam.step()" % (B, N, ", ".join("%d x %s %.2f s" % (v[1], k, v[0])
                                                                           for k, v in times.items())),
           "threads": threads, "logical_cpus": ncpu, "physical_cores": phys, "cpu_model": model,
           "torch_version": torch.__version__}
    for k, v in times.items():
        out[k.replace("_step", "") + "_steps_per_s"] = round(1.0 / v[0], 4)
    return out


# ------------------------------------------------------------------------------------------ the other configurations
def side_line(name, device, arith=None, steps=6, warm=3):
    """A compact line for another BASELINE.json configuration (or for the metric configuration under the exact-f32 arithmetic), run
    AFTER and OUTSIDE the headline line's timed region so that the driver's default `python bench.py` sees every configuration this
    repo quotes: `steps` sampling steps (hipGraph replay) and `steps` training steps, timed separately with barrier + synchronize."""
    import torch
    from diffuscene_amd import _lib
    prev = _lib.set_gemm_arithmetic(arith) if arith else None
    model = sr = tr = None
    try:
        spec = dict(CONFIGS[name])
        B, N = spec["batch"], spec["objects"]
        model, _ = build_model(spec, device)
        with contextlib.redirect_stderr(io.StringIO()):
            sr = SampleRunner(spec, model, device, seed=0)
        sr.run(warm)
        sr.reset()
        ts = timed(1, lambda: sr.run(steps)) / steps
        tr = TrainRunner(spec, model, device, 0)
        tr.run(max(warm, 3))                       # step 1 eager, step 2 captures the graph, step 3 replays
        tt = timed(1, lambda: tr.run(steps)) / steps
        rf = roofline_dominant_kernel(sr.g.plan, N, name)
        F = forward_flops(spec["kind"], B, N, spec.get("text_len", 0))
        return {"workload": "%s, B=%d, N=%d" % (spec["title"], B, N), "steps": "%d sample + %d train" % (steps, steps),
                "arithmetic": dtype_label(), "steps_per_s": round(2.0 / (ts + tt), 2),
                "sample": {"ms_per_step": round(ts * 1e3, 3), "steps_per_s": round(1.0 / ts, 2), "tflops": round(F / ts / 1e12, 1)},
                "train": {"ms_per_step": round(tt * 1e3, 3), "steps_per_s": round(1.0 / tt, 2), "tflops": round(3.0 * F / tt / 1e12, 1)},
                "gemm_flops_on_split_kernel": round(split_flop_fraction(sr.g.plan), 4),
                "gn_gemm": {k: rf[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "frac_mix", "avg_launch_us", "algorithmic_tflops")}}
    finally:
        del sr, tr, model
        torch.cuda.empty_cache()
        if prev is not None:
            _lib.set_gemm_arithmetic(prev)


def public_api_line(device):
    """What a user of the drop-in gets from the PUBLIC sampling entry points, wall time including everything the method does (condition
    build, x_T draw, 1000 reverse steps, post-filter, copy to the host) -- round-5 review items 4 / 5:
      generate_b1_n12 / _n21   network.generate_layout(batch_size=1), the call of scripts/generate_diffusion.py:314-323 (one scene per call),
                               with the default environment (captured hipGraph loop since round 6) and with DSC_GRAPH=0 (eager loop:
                               ~110 ctypes launches per step from Python); first call (capture) and steady state apart; and with the
                               K-parallel small-launch GEMM off (DSC_SKINNY=0: the tile kernels of rounds 1-5)
      generate_batched_b256    network.generate_layout_batched(batch_size=256) at the metric shape (N = 80)"""
    import torch
    out = {}

    def wall(fn, reps):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return ts

    for name, N in (("generate_b1_n12", 12), ("generate_b1_n21", 21)):
        spec = dict(CONFIGS["bedroom21"], batch=1, objects=N)
        model, cfg = build_model(spec, device)
        model.eval()
        room = torch.zeros(1, 1, 64, 64, device=device)
        call = lambda: model.generate_layout(room_mask=room, num_points=N, point_dim=cfg["point_dim"], batch_size=1, text=None,  # noqa: E731
                                             device="cpu", clip_denoised=True)
        row = {"workload": "uncond bedrooms, generate_layout(batch_size=1), N=%d, T=1000" % N}
        with contextlib.redirect_stdout(io.StringIO()):
            os.environ.pop("DSC_GRAPH", None)
            first = wall(call, 1)[0]
            ts = sorted(wall(call, 3))
            row["default_env"] = {"loop": "captured hipGraph step (default)", "first_call_s": round(first, 3), "seconds_per_scene": round(ts[1], 4),
                                  "denoiser_steps_per_s": round(1000.0 / ts[1], 1)}
            os.environ["DSC_GRAPH"] = "0"
            wall(call, 1)
            te = sorted(wall(call, 2))
            os.environ.pop("DSC_GRAPH", None)
            row["DSC_GRAPH=0"] = {"loop": "eager Python loop", "seconds_per_scene": round(te[0], 4), "denoiser_steps_per_s": round(1000.0 / te[0], 1)}
        row["graph_over_eager"] = round(te[0] / ts[1], 2)
        # the same call with the K-parallel small-launch GEMM (csrc/gemm_skinny.h, round 6) switched off: the tile kernels of rounds 1-5.  A fresh
        # model: the captured step of the first one has its kernels baked in
        from diffuscene_amd import _lib
        lib = _lib.load()
        prev = lib.dsc_set_skinny(0)
        try:
            model0, _ = build_model(spec, device)
            model0.eval()
            call0 = lambda: model0.generate_layout(room_mask=room, num_points=N, point_dim=cfg["point_dim"], batch_size=1, text=None,  # noqa: E731
                                                   device="cpu", clip_denoised=True)
            with contextlib.redirect_stdout(io.StringIO()):
                wall(call0, 1)
                t0s = sorted(wall(call0, 2))
            row["DSC_SKINNY=0"] = {"loop": "captured hipGraph step, every launch on the tile kernels (K-parallel small-launch GEMM off)",
                                   "seconds_per_scene": round(t0s[0], 4), "denoiser_steps_per_s": round(1000.0 / t0s[0], 1)}
            row["k_parallel_gain"] = round(t0s[0] / ts[1], 2)
            del model0
        finally:
            lib.dsc_set_skinny(prev)
        out[name] = row
        log("public_api: %s default %.3f s / scene (first call %.2f s), eager %.3f s, tile kernels only %.3f s"
            % (name, ts[1], first, te[0], row["DSC_SKINNY=0"]["seconds_per_scene"]))
        del model
        torch.cuda.empty_cache()
    spec = dict(CONFIGS["living80"])
    model, cfg = build_model(spec, device)
    model.eval()
    B, N = spec["batch"], spec["objects"]
    room = torch.zeros(B, 1, 64, 64, device=device)
    call = lambda: model.generate_layout_batched(room_mask=room, num_points=N, point_dim=cfg["point_dim"], batch_size=B, text=None,  # noqa: E731
                                                 clip_denoised=True)
    with contextlib.redirect_stdout(io.StringIO()):
        first = wall(call, 1)[0]
        ts = sorted(wall(call, 2))
    out["generate_batched_b256"] = {"workload": "uncond living rooms, generate_layout_batched(batch_size=%d), N=%d, T=1000: condition build, x_T, "
                                                "1000 replayed steps, per-scene post-filter on the device, one copy to the host" % (B, N),
                                    "first_call_s": round(first, 3), "seconds": round(ts[0], 3), "scenes_per_s": round(B / ts[0], 1),
                                    "denoiser_steps_per_s": round(1000.0 / ts[0], 1)}
    log("public_api: generate_layout_batched B=%d %.3f s" % (B, ts[0]))
    return out


def side_line_child(what, timeout=600):
    """side_line(CONFIG[:ARITH]) in a CHILD process (`python bench.py --side-line ...`): a side line must never cost the headline line --
    an exception, a hang or a GPU fault there ends the child and is recorded as {"error": ...}; the parent still prints its line."""
    cmd = [sys.executable, os.path.abspath(__file__), "--side-line", what]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": "timed out after %d s" % timeout}
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode == 0 and lines:
        try:
            return json.loads(lines[-1])
        except ValueError:
            pass
    tail = " | ".join((r.stderr or "").strip().splitlines()[-3:])
    return {"error": "child exit code %d: %s" % (r.returncode, tail[-400:])}


# ------------------------------------------------------------------------------------------ launcher
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n):
    """`python bench.py --gpus N` from a plain shell: re-exec under torch.distributed.run, one rank per GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def ddp_selftest(args, device):
    """--ddp-selftest: the N > 1 training step on ONE GPU.  torch.distributed 'nccl' (= RCCL) with world_size 1 and
    DSC_DDP_FORCE=1 runs exactly what every rank of an 8-GPU job runs -- the bucket schedule of the plan, the hipGraph segments cut
    at the bucket launches, 8 in-place all-reduces of the flat gradient buffer issued between the segment replays (each a
    world-1 RCCL call), clip + Adam on the reduced gradients -- so its step time next to the single-GPU graph step is the
    overhead of the data-parallel form itself (no link traffic).  Both flush schedules (DSC_DDP_FLUSH=block|end) are timed."""
    import torch
    import torch.distributed as dist
    base = dict(CONFIGS[args.config])
    if args.batch:
        base["batch"] = args.batch
    n, warm = max(args.steps // 2, 4), max(args.warmup, 3)
    res = {"config": args.config, "steps": n, "warmup": warm, "rows": []}

    def step_ms(spec, model):
        tr = TrainRunner(spec, model, device, 0)
        tr.run(warm)
        return timed(1, lambda: tr.run(n)) / n * 1e3, tr

    for B in (base["batch"], max(base["batch"] // 8, 1)):
        spec = dict(base, batch=B)
        os.environ.pop("DSC_DDP_FORCE", None)
        model, _ = build_model(spec, device)
        single, _ = step_ms(spec, model)
        row = {"batch_per_gpu": B, "single_gpu_graph_ms": round(single, 3)}
        log("B=%d single-GPU graph step %.3f ms" % (B, single))
        del model
        if not dist.is_initialized():
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1, device_id=device)
        os.environ["DSC_DDP_FORCE"] = "1"
        for flush in ("block", "thirds", "single"):
            os.environ["DSC_DDP_FLUSH"] = flush
            model, _ = build_model(spec, device)
            ms, tr = step_ms(spec, model)
            ent = next(iter(model._dsc_plan_runner.plans.values()))
            red, sg = ent["reducer"], ent["graph"]
            row["ddp_%s" % flush] = {"ms": round(ms, 3), "vs_single": round(ms / single, 4), "buckets": len(red.buckets),
                                     "graph_segments": len(sg.segments) if sg is not None else 0,
                                     # launches of the backward still to run when each bucket's all-reduce is issued (0 for every
                                     # bucket = nothing left to overlap with: the `single` schedule)
                                     "bwd_launches_remaining_at_each_bucket": sorted((len(ent["plan"].bwd) - 1 - c for c, bs in red.at_launch.items() for _ in bs), reverse=True)}
            log("B=%d DDP-mode step (%s flush) %.3f ms = %.3f x single" % (B, flush, ms, ms / single))
            if flush == "block" and B == base["batch"]:
                from diffuscene_amd import ddp
                fs = model._dsc_flat
                g = torch.zeros_like(fs.G)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    for w in [dist.all_reduce(g[s_:e_], op=dist.ReduceOp.SUM, async_op=True) for s_, e_ in fs.buckets(ddp.N_BUCKETS)]:
                        w.wait()
                e1.record()
                torch.cuda.synchronize()
                res["allreduce_world1"] = {"ms_per_step": round(e0.elapsed_time(e1) / 5, 3), "bytes": fs.numel * 4,
                                           "what": "the 8 bucket all-reduces of one step alone, world_size 1 (launch + local copy cost, no links)"}
            del model, tr
        res["rows"].append(row)
    os.environ.pop("DSC_DDP_FORCE", None)
    dist.destroy_process_group()
    print(json.dumps({"ddp_selftest": res, "git_head": git_head(),
                      "dtype": dtype_label()}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default=os.environ.get("DSC_BENCH_CONFIG", "living80"), choices=sorted(CONFIGS))
    ap.add_argument("--mode", default=os.environ.get("DSC_BENCH_MODE", "both"), choices=["both", "sample", "train"])
    ap.add_argument("--batch", type=int, default=None, help="override the config's per-GPU batch")
    ap.add_argument("--objects", type=int, default=None, help="override the config's objects per scene")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): the config's batch per GPU; strong: the config's batch is the GLOBAL batch, split over "
                         "the ranks (SURVEY.md 8e: B=256 global = 32 scenes per GPU at 8 GPUs, communication-dominated)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sweep", action="store_true", help="cpu_baseline: sweep the host thread count on the full-batch steps (~5 minutes)")
    ap.add_argument("--no-full-loop", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the `other_configs` / `exact_f32` blocks the default single-GPU run appends after the headline line's "
                         "timed region (the four other BASELINE configs and the metric config on the exact-f32 kernels, 6+6 steps each)")
    ap.add_argument("--side-line", default=None, metavar="CONFIG[:ARITH]",
                    help="(internal) print the compact side line of one configuration and exit: the default run launches one child "
                         "process per side line, so that nothing that happens there can cost the headline line")
    ap.add_argument("--cpu-slice-probe", default=None, metavar="THREADS:SCENES", help=argparse.SUPPRESS)
    ap.add_argument("--ddp-selftest", action="store_true",
                    help="one GPU: time the DATA-PARALLEL form of the training step (world-1 RCCL group, reducer forced on: bucket "
                         "schedule, hipGraph segments, 8 in-place all-reduces) next to the single-GPU graph step, at the config's "
                         "batch and at 1/8 of it (the strong-scaling shard)")
    args = ap.parse_args()
    if args.cpu_slice_probe:                      # child of cpu_baseline's all-cores leg: CPU only
        cpu_slice_probe(args.cpu_slice_probe, args.config)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    ws = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("DSC_BENCH_DRYRUN"):          # launcher plumbing only (CPU test): what each rank would run
        sys.stdout.write(json.dumps({"rank": int(os.environ.get("RANK", "0")), "local_rank": local, "world": ws, "gpus": args.gpus,
                          "config": args.config, "scaling": args.scaling, "master": os.environ.get("MASTER_ADDR"),
                          "ddp_flush": os.environ.get("DSC_DDP_FLUSH", "single"), "ipc_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                          "batch_per_rank": (CONFIGS[args.config]["batch"] // ws) if args.scaling == "strong"
                          else CONFIGS[args.config]["batch"]}) + "\n")        # ONE write per rank: the ranks share a pipe, print() would emit the newline separately
        sys.stdout.flush()
        if ws != args.gpus:
            raise SystemExit(2)
        return
    import torch
    import torch.distributed as dist
    if ws != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torchrun --nproc-per-node %d, or from a "
                         "plain shell so that bench.py spawns the ranks itself)" % (args.gpus, ws, args.gpus))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # DSC_BENCH_DDP_REHEARSAL=1 (one GPU): run the MULTI-GPU code path of this script -- process group, broadcast, the three-schedule
    # comparison, reducer, MAX all-reduces -- on a world-1 RCCL group with the reducer forced on, so that every line the driver's
    # 8-GPU run executes has executed on hardware before (no multi-GPU box was ever available to the builder).  `mw` = what the helpers
    # that branch on "more than one rank" see.
    rehearsal = ws == 1 and os.environ.get("DSC_BENCH_DDP_REHEARSAL") == "1" and not (args.ddp_selftest or args.side_line)
    mw = 2 if rehearsal else ws
    if mw > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        with phase("init_process_group (RCCL)", mw):
            if rehearsal:
                os.environ["DSC_DDP_FORCE"] = "1"
                dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1, device_id=device,
                                        timeout=datetime.timedelta(seconds=240))
            else:
                dist.init_process_group(backend="nccl", device_id=device, timeout=datetime.timedelta(seconds=240))
    rank = dist.get_rank() if mw > 1 else 0
    if args.ddp_selftest:
        if ws != 1:
            raise SystemExit("bench.py: --ddp-selftest is a single-GPU run")
        return ddp_selftest(args, device)
    if args.side_line:
        if ws != 1:
            raise SystemExit("bench.py: --side-line is a single-GPU run")
        name, _, arith = args.side_line.partition(":")
        print(json.dumps(public_api_line(device) if name == "public_api" else side_line(name, device, arith=arith or None)), flush=True)
        return

    spec = dict(CONFIGS[args.config])
    if args.batch:
        spec["batch"] = args.batch
    if args.objects:
        spec["objects"] = args.objects
    if args.scaling == "strong":
        if spec["batch"] % ws:
            raise SystemExit("bench.py: --scaling strong needs the global batch (%d) divisible by the ranks (%d)" % (spec["batch"], ws))
        spec["batch"] //= ws
    B, N = spec["batch"], spec["objects"]
    log("building model (%s: B=%d per GPU, N=%d, %s scaling)" % (args.config, B, N, args.scaling))
    with phase("build_model", mw):
        model, cfg = build_model(spec, device)
    if mw > 1:
        from diffuscene_amd import ddp
        with phase("broadcast_parameters (first RCCL collective)", mw):
            ddp.broadcast_parameters(model)         # every replica starts from rank 0's weights
            barrier(mw)                             # no collective in flight while the sampling graph is being captured
    log("model on device")
    n_s = {"both": (args.steps + 1) // 2, "sample": args.steps, "train": 0}[args.mode]
    n_t = args.steps - n_s
    with phase("capture sampling graph", mw):
        sr = SampleRunner(spec, model, device, seed=rank) if n_s else None
        if sr:
            sr.run(args.warmup)
            sr.reset()
    schedules, chosen = None, None
    if mw > 1 and n_t:
        # one driver run compares the three data-parallel schedules on the node; the headline regions use the fastest
        schedules, chosen = compare_ddp_schedules(spec, model, device, rank, mw)
        if chosen is None:
            raise SystemExit("bench.py: no data-parallel schedule ran on all ranks: %s" % json.dumps(schedules))
        os.environ["DSC_DDP_FLUSH"] = chosen
    with phase("training warm-up (eager step, graph capture, first all-reduces)", mw):
        tr = TrainRunner(spec, model, device, rank) if n_t else None
        if tr:
            tr.run(args.warmup)
        if sr and tr:
            sr.sync_weights()

    def region():
        if sr:
            sr.run(n_s)
        if tr:
            tr.run(n_t)

    # Headline = the MEDIAN of three timed regions of exactly K steps each (barrier + synchronize on both sides, max over ranks per
    # region): box-to-box and run-to-run spread (2-4 %) is larger than a single optimisation of the later rounds, so one region
    # is not a measurement; min / max are reported beside it.
    regions = []
    for _ in range(3):
        if sr:
            sr.reset()
        regions.append(timed(mw, region))
    log("timed regions: %s s for %d steps each (%d sample + %d train)" % (", ".join("%.3f" % r for r in regions), args.steps, n_s, n_t))
    parts = {}
    if args.mode == "both":          # the two rates separately (outside the headline regions), median of three as well
        ps, pt = [], []
        for _ in range(3):
            sr.reset()
            ps.append(timed(mw, lambda: sr.run(n_s)) / n_s)
            pt.append(timed(mw, lambda: tr.run(n_t)) / max(n_t, 1))
        parts = {"sample": ps, "train": pt}
    full = None
    if sr and not args.no_full_loop:
        if tr:
            sr.sync_weights()        # the loop samples from the weights as the optimizer left them (derived copies re-made, untimed)
        full = sr.full_loop()
        log("full 1000-step loop: %.3f s" % full)
    flat = regions + parts.get("sample", [0.0] * 3) + parts.get("train", [0.0] * 3) + [full or 0.0]
    tmax = torch.tensor(flat, device=device, dtype=torch.float64)
    if mw > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tmax = [float(v) for v in tmax.tolist()]
    regions = tmax[0:3]
    dt = sorted(regions)[1]
    if parts:
        parts = {"sample": sorted(tmax[3:6]), "train": sorted(tmax[6:9])}
    if full is not None:
        full = tmax[9]
    comm = None
    if mw > 1 and tr:
        from diffuscene_amd import ddp
        comm = ddp.measure_allreduce(model, reps=5)         # the gradient exchange alone (no compute to hide behind)

    if rank == 0:
        kind = spec["kind"]
        steps_per_s = args.steps * ws / dt        # whole job: every rank advances its own B scenes one step
        what = {"sample": "1000-step sample loop", "train": "train step", "both": "train + 1000-step sample"}[args.mode]
        out = {
            "metric": "denoiser steps/sec (%s) at B=%d, N=%d objects" % (what, B, N),
            "value": round(steps_per_s, 3), "unit": "steps/s", "n_gpus": ws, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "statistic": "median of 3 timed regions of %d steps each" % args.steps,
            "regions_steps_per_s": {"min": round(args.steps * ws / max(regions), 3), "median": round(steps_per_s, 3),
                                    "max": round(args.steps * ws / min(regions), 3)},
            "scaling": args.scaling, "vs_baseline": None,
            "dtype": dtype_label(), "data": "synthetic",
            "config": {"workload": "%s (%s), B=%d scenes per GPU, N=%d, C=%d, T=1000, mode=%s"
                                   % (spec["title"], spec["yaml"], B, N, 8 + spec["class_dim"] + 32, args.mode),
                       "name": args.config, "global_batch": B * ws,
                       "parallelism": "dp%d" % ws if ws > 1 else "single"},
            "git_head": git_head(),
        }
        F = forward_flops(kind, B, N, spec.get("text_len", 0))
        # a sampling step executes LESS than the reference's forward: the time MLP is a table, conditioning-only launches run once per
        # loop -- its rate is quoted on the flops the captured step executes (the plan's own launch list); training runs all of 3 F
        F_s = plan_executed_flops(sr.g.plan) if sr else F
        out["model_tflops"] = round((F_s * n_s + 3.0 * F * n_t) / dt / 1e12, 2)        # per GPU, algorithmic (train = 3F)
        for k, v in parts.items():
            med = v[1]
            out[k] = {"steps_per_s": round(ws / med, 3), "ms_per_step": round(med * 1e3, 3),
                      "ms_per_step_min_max": [round(v[0] * 1e3, 3), round(v[2] * 1e3, 3)],
                      "tflops_per_gpu": round((F_s if k == "sample" else 3.0 * F) / med / 1e12, 2)}
            if k == "sample":
                out[k]["flops_per_step"] = F_s
                out[k]["reference_forward_flops"] = F
                out[k]["note"] = ("tflops_per_gpu counts the flops the captured step executes (time MLP tabulated, conditioning-only "
                                  "launches hoisted out of the loop); the reference's forward is reference_forward_flops")
        if full is not None:
            out["full_loop"] = {"steps": 1000, "seconds": round(full, 3), "steps_per_s": round(1000.0 * ws / full, 2),
                                "what": "wall time of one whole 1000-step p_sample_loop via the captured graph "
                                        "(x_T draw, 1000 replays, result copy), per-GPU batch %d" % B}
        if comm is not None:
            out["allreduce"] = comm
        if schedules is not None:
            out["ddp_schedules"] = {"rows": schedules, "used_for_the_headline": chosen, "env": "DSC_DDP_FLUSH",
                                    "what": "training step (ms, max over ranks, 6 steps) under each flush schedule of the gradient "
                                            "exchange, measured in this run; UNMEASURED ON HARDWARE before the first SCALE run"}
        plan = sr.g.plan if sr else None
        if plan is None:
            with torch.no_grad():
                eng = model.diffusion.model.engine(device)
                ctx_dim = 512 if kind == "arrange" else 128
                cond = torch.zeros(N, ctx_dim, device=device)[None].expand(B, -1, -1)
                cross = torch.zeros(B, spec["text_len"], 512, device=device) if kind == "text" else None
                plan = eng.prepare(B, N, cond, cross)
                plan.x_in.normal_(); plan.t_in.fill_(500); plan.run()
        out["roofline"] = roofline_dominant_kernel(plan, N, args.config)
        log("roofline done")
        # whole-model roofline fraction.  Exact-f32 arithmetic: algorithmic flops against the f32-MFMA peak.  Split arithmetic: only the
        # launches the library's dispatcher really runs on the split kernel (dsc_gemm_arithmetic on the plan's own argument structs)
        # execute 6 bf16 products per f32 product -- small launches stay on the f32-MFMA kernel (the text config: all of them) and are
        # counted once; quoted only when the split launches carry the bulk of the forward's GEMM flops
        sf = split_flop_fraction(plan)
        out["gemm_flops_on_split_kernel"] = round(sf, 4)
        from diffuscene_amd import _lib
        if not _lib.split_enabled():
            out["model_frac_of_fp32_mfma_peak"] = round(out["model_tflops"] / PEAK_FP32_MFMA_TFLOPS, 4)
        elif sf >= 0.9:
            mult = 1.0 + (SPLIT_PRODUCTS - 1) * sf
            out["model_executed_tflops"] = round(out["model_tflops"] * mult, 1)
            out["model_frac_of_bf16_mfma_peak"] = round(out["model_tflops"] * mult / PEAK_BF16_MFMA_TFLOPS, 4)
        if not args.no_cpu_baseline and ws == 1:          # the CPU baseline is a single-GPU-run figure (rank 0, N = 1 only)
            out["cpu_baseline"] = cpu_baseline(spec, args.mode, sweep=args.cpu_sweep)
        default_run = (ws == 1 and args.config == "living80" and args.mode == "both" and not args.batch and not args.objects
                       and args.scaling == "weak")
        if default_run and not args.no_other_configs:
            # everything else this repo quotes, in the driver's own run (outside the timed region above; its memory is released first)
            from diffuscene_amd import _lib
            del sr, tr, plan, model
            torch.cuda.empty_cache()
            out["other_configs"] = {}
            for name in ("bedroom21", "text", "complete", "arrange"):
                out["other_configs"][name] = side_line_child(name)
                log("other_configs: %s done" % name)
            if _lib.split_enabled():
                out["exact_f32"] = side_line_child("living80:f32")
                log("exact_f32 done")
            out["public_api"] = side_line_child("public_api")
            log("public_api done")
        if rehearsal:
            out["ddp_rehearsal"] = "world-1 RCCL group, reducer forced on: the multi-GPU code path of this script on ONE GPU (not a scaling number)"
        print(json.dumps(out), flush=True)
    if mw > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
