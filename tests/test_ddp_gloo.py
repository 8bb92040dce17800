"""CPU, world_size 2, gloo: the data-parallel gradient step (ddp.py) -- bucketed all-reduce mean, shard_batch and the
fused clip -- reproduces the single-process gradient on the concatenated batch."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.SiLU(), torch.nn.Linear(64, 64), torch.nn.SiLU(),
                               torch.nn.Linear(64, 8))


def _batch():
    g = torch.Generator().manual_seed(1)
    return {"x": torch.randn(12, 16, generator=g), "y": torch.randn(12, 8, generator=g), "description": list("abcdefghijkl")}


def _loss(m, b):
    return ((m(b["x"]) - b["y"]) ** 2).mean()


def _worker(rank, ws, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from diffuscene_amd import ddp
    m = _model()
    shard = ddp.shard_batch(_batch())
    assert shard["x"].shape[0] == 6 and len(shard["description"]) == 6
    _loss(m, shard).backward()
    nb = ddp.average_gradients(m, bucket_bytes=4096)          # tiny buckets: exercises the multi-bucket path
    assert nb >= 3
    total = ddp.clip_grad_norm_fused(m.parameters(), 0.05)
    flat = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(ws)]
    dist.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1])               # every rank clips / steps identically
    # overlapped form: bucket all-reduces launched from gradient hooks during backward give the same averaged gradients
    m2 = _model()
    red = ddp.overlapped_reducer(m2, bucket_bytes=4096)
    assert red is ddp.overlapped_reducer(m2) and len(red.buckets) >= 3
    for _ in range(2):                                          # two steps: state resets between steps
        m2.zero_grad(set_to_none=True)
        _loss(m2, shard).backward()
        assert red.launched_during_backward >= 3                # complete buckets left before backward ended
        assert red.finish() == len(red.buckets)
    flat2 = torch.cat([p.grad.reshape(-1) for p in m2.parameters()])
    ddp.clip_grad_norm_fused(m2.parameters(), 0.05)
    flat2 = torch.cat([p.grad.reshape(-1) for p in m2.parameters()])
    assert torch.allclose(flat2, flat, rtol=1e-6, atol=1e-9)
    # a parameter without gradient (frozen branch) must not dead-lock the bucket schedule
    m3 = _model()
    m3[4].weight.requires_grad_(False)
    red3 = ddp.OverlappedGradientReducer(m3, bucket_bytes=4096)
    _loss(m3, shard).backward()
    red3.finish()
    if rank == 0:
        torch.save({"grad": flat, "norm": total}, out)
    dist.destroy_process_group()


def test_ddp_mean_gradients_match_single_process(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    from diffuscene_amd import ddp
    m = _model()
    _loss(m, _batch()).backward()
    ref_norm = torch.nn.utils.clip_grad_norm_(m.parameters(), 0.05)
    ref = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    assert torch.allclose(got["norm"], ref_norm, rtol=1e-5)
    assert torch.allclose(got["grad"], ref, rtol=1e-5, atol=1e-8)
    assert ddp.world() == 1 and ddp.average_gradients(m) == 0   # no-op when not distributed


# ---------------------------------------------------------------------------------------------------------------------
# the product path: static training plan + flat gradient buffer + FlatGradientReducer (world_size 2, gloo).  The plan is
# lowered by the CPU backend of tests/plan_sim.py; everything else (flat storage, bucket schedule, in-place all-reduce
# launched from the backward's progress callback, parameter broadcast, 1/world folded into the loss gradient) is the product.
# ---------------------------------------------------------------------------------------------------------------------
def _plan_setup(B, N, seed, tmp, grad_scale, per_block):
    import contextlib
    import io
    import json
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from plan_sim import SimBackend
    from oracle import weights as W
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    from diffuscene_amd.flat import FlatStorage
    from diffuscene_amd.train_plan import TrainPlan
    from diffuscene_amd._lib import SS_PER_SLOT
    kw = dict(W.UNCOND_BEDROOM)
    stats = os.path.join(tmp, "stats_%d.txt" % os.getpid())
    with open(stats, "w") as f:
        json.dump(W.DATASET_STATS, f)
    with contextlib.redirect_stdout(io.StringIO()):
        net = Unet1D(**kw)
        net.load_state_dict(W.synth_state_dict(kw, seed=seed))
        dp = DiffusionPoint(net, dict(objectness_dim=0, class_dim=22, angle_dim=2, objfeat_dim=32), time_num=1000,
                            model_mean_type="v", loss_separate=True, loss_iou=True, train_stats_file=stats)

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.positional_embedding = torch.nn.Parameter(W.synth_condition(1, N, 128, seed)[0].clone())
            self.net = net
    holder = Holder()
    flat = FlatStorage(holder)
    tb = {n: getattr(dp.diffusion, n) for n in dp.diffusion._TABLE_NAMES}

    def make_plan():
        return TrainPlan(net, flat, dp.diffusion, B, N, SS_PER_SLOT, 128, 0, 0, SimBackend(), per_block_grads=per_block,
                         ctx_param=holder.positional_embedding, tables=tb, grad_scale=grad_scale)
    return holder, flat, make_plan


def _plan_inputs(Bg, N):
    from oracle import weights as W
    x0 = W.synth_scene_batch(Bg, N, 22, 32, seed=11)
    noise = W.synth_noise((Bg, N, 62), 12)
    t = torch.tensor([5, 300, 650, 999])[:Bg]
    return x0, noise, t


def _plan_worker(rank, ws, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    torch.set_num_threads(2)
    from diffuscene_amd import ddp
    Bg, N = 4, 5
    Bl = Bg // ws
    # ranks start from DIFFERENT weights: broadcast_parameters must make them identical
    holder, flat, make_plan = _plan_setup(Bl, N, seed=rank, tmp=tmp, grad_scale=1.0 / Bg, per_block=True)
    ddp.broadcast_parameters(holder)
    plan = make_plan()
    red = ddp.FlatGradientReducer(flat, plan, n_buckets=8)
    x0, noise, t = _plan_inputs(Bg, N)
    sl = slice(rank * Bl, (rank + 1) * Bl)
    plan.x0.copy_(x0[sl]); plan.noise.copy_(noise[sl]); plan.t.copy_(t[sl])
    for p in flat.params:                      # every gradient must be WRITTEN by the plan (alignment gaps of G stay 0)
        flat.grad_view(p).fill_(float("nan"))
    plan.run_forward()
    plan.run_backward(on_progress=red.on_progress)
    n = red.finish()
    assert n == len(red.buckets) >= 8
    assert red.launched_during_backward >= 5, red.launched_during_backward     # overlapped with the backward
    assert torch.isfinite(flat.G).all()
    # the captured form of the same step (train_step._StepGraphs): the launch list cut into segments at the launches that finish a
    # bucket, the reducer called between segment replays -- same buckets in the same order, same reduced gradients
    from diffuscene_amd.train_step import _capture
    g_eager, order_eager = flat.G.clone(), list(red.last_order)
    for p in flat.params:
        flat.grad_view(p).fill_(float("nan"))
    sg = _capture(plan, red, torch.device("cpu"))
    assert len(sg.segments) >= 5 and sg.segments[0][0] == 0 and sg.segments[-1][1] == len(plan.bwd)
    assert all(a[1] == b[0] for a, b in zip(sg.segments, sg.segments[1:]))          # contiguous, nothing skipped or repeated
    sg.replay()
    assert red.finish() == n and red.last_order == order_eager
    assert torch.equal(flat.G, g_eager)
    if rank == 0:
        torch.save({"G": flat.G.clone(), "P": flat.P.clone(), "order": red.last_order,
                    "losses": plan.losses.clone()}, os.path.join(tmp, "plan_r0.pt"))
    else:
        torch.save({"P": flat.P.clone()}, os.path.join(tmp, "plan_r1.pt"))
    dist.destroy_process_group()


def test_flat_reducer_with_training_plan_matches_single_process(tmp_path):
    tmp = str(tmp_path)
    mp.spawn(_plan_worker, args=(2, _free_port(), tmp), nprocs=2, join=True)
    r0, r1 = torch.load(os.path.join(tmp, "plan_r0.pt")), torch.load(os.path.join(tmp, "plan_r1.pt"))
    assert torch.equal(r0["P"], r1["P"]), "broadcast_parameters must leave identical replicas"
    # buckets leave in the order the backward finishes them (decoder heads / last blocks first), the first bucket (wrapper-level
    # parameters, context MLPs, encoders) last
    assert r0["order"][-1] == 0 and r0["order"] != sorted(r0["order"]), r0["order"]
    # single process, whole batch
    Bg, N = 4, 5
    holder, flat, make_plan = _plan_setup(Bg, N, seed=0, tmp=tmp, grad_scale=1.0 / Bg, per_block=False)
    plan = make_plan()
    x0, noise, t = _plan_inputs(Bg, N)
    plan.x0.copy_(x0); plan.noise.copy_(noise); plan.t.copy_(t)
    plan.run_forward()
    plan.run_backward()
    assert torch.equal(flat.P, r0["P"])
    assert torch.allclose(plan.losses[:2], r0["losses"], rtol=1e-5, atol=1e-7)
    err = float((flat.G - r0["G"]).norm() / flat.G.norm())
    assert err < 1e-5, err


# ---------------------------------------------------------------------------------------------------------------------
# the RUNNER path of train_on_batch under gloo, world_size 2: loss_step() itself -- parameter broadcast on the first step, plan cache,
# FlatGradientReducer built from the plan, the eager first step, the segmented ("captured") second step, finish() -- with the torch
# backend of tests/plan_sim.py lowering the plan.  Each rank's gradients must equal the mean of the two single-process gradients of
# the same shards under the same seeds.
# ---------------------------------------------------------------------------------------------------------------------
def _wrapper_model(tmp, seed):
    import contextlib
    import io
    import json
    from oracle import weights as W
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import DiffusionSceneLayout_DDPM
    stats = os.path.join(tmp, "stats_w_%d.txt" % os.getpid())
    with open(stats, "w") as f:
        json.dump(W.DATASET_STATS, f)
    nc, N = 22, 5
    cfg = {"type": "diffusion_scene_layout_ddpm", "net_type": "unet1d", "point_dim": 8 + nc + 32, "latent_dim": 0,
           "room_mask_condition": False, "sample_num_points": N, "objectness_dim": 0, "objfeat_dim": 32, "class_dim": nc,
           "angle_dim": 2, "learnable_embedding": True, "instance_condition": True, "instance_emb_dim": 128,
           "diffusion_kwargs": dict(schedule_type="linear", beta_start=1e-4, beta_end=0.02, time_num=1000, loss_type="mse",
                                    model_mean_type="v", model_var_type="fixedsmall", loss_separate=True, loss_iou=True,
                                    train_stats_file=stats),
           "net_kwargs": dict(W.UNCOND_BEDROOM)}
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        m = DiffusionSceneLayout_DDPM(nc + 1, None, cfg)
    return m, nc, N


def _shard(nc, N, lo, hi):
    from oracle import weights as W
    x = W.synth_scene_batch(4, N, nc, 32, seed=31)[lo:hi]
    return {"translations": x[:, :, 0:3].contiguous(), "sizes": x[:, :, 3:6].contiguous(), "angles": x[:, :, 6:8].contiguous(),
            "class_labels": x[:, :, 8:8 + nc].contiguous(), "objfeats_32": x[:, :, 8 + nc:].contiguous(),
            "room_layout": torch.zeros(hi - lo, 1, 64, 64)}


def _use_sim_backend():
    """Swap the three HIP-only pieces for their CPU stand-ins; returns (train_step, undo)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from plan_sim import SimBackend
    from diffuscene_amd import train_step
    from diffuscene_amd.networks.diffusion_ddpm import GaussianDiffusion
    saved = (train_step.PlanRunner.backend_factory, train_step.plan_supported, GaussianDiffusion.tables)

    def cpu_tables(self, device):                              # (the product's tables() insists on a HIP device)
        return {n: getattr(self, n).float() for n in self._TABLE_NAMES}

    def undo():
        train_step.PlanRunner.backend_factory, train_step.plan_supported, GaussianDiffusion.tables = saved
    train_step.PlanRunner.backend_factory = staticmethod(lambda dev: SimBackend())
    train_step.plan_supported = lambda model: True             # (and so does the product's plan_supported)
    GaussianDiffusion.tables = cpu_tables
    return train_step, undo


def _runner_worker(rank, ws, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    torch.set_num_threads(2)
    ts, _ = _use_sim_backend()
    m, nc, N = _wrapper_model(tmp, seed=rank)                  # different weights per rank: the first step must broadcast rank 0's
    s = _shard(nc, N, 2 * rank, 2 * rank + 2)
    out = {}
    for step in range(2):                                      # step 0: eager with per-launch progress; step 1: segment replay
        torch.manual_seed(50 + 10 * step + rank)
        loss, parts, ent = ts.loss_step(m, s, backward=True)
        out["G%d" % step] = m._dsc_flat.G.clone()
        out["loss%d" % step] = float(loss)
    assert ent["reducer"] is not None and ent["graph"] is not None and len(ent["graph"].segments) >= 2
    out["P"] = m._dsc_flat.P.clone()
    torch.save(out, os.path.join(tmp, "runner_r%d.pt" % rank))
    dist.destroy_process_group()


def test_loss_step_runner_path_world2_gloo(tmp_path):
    tmp = str(tmp_path)
    mp.spawn(_runner_worker, args=(2, _free_port(), tmp), nprocs=2, join=True)
    r0, r1 = torch.load(os.path.join(tmp, "runner_r0.pt")), torch.load(os.path.join(tmp, "runner_r1.pt"))
    assert torch.equal(r0["P"], r1["P"]), "the first step must broadcast rank 0's parameters"
    for step in range(2):
        assert torch.equal(r0["G%d" % step], r1["G%d" % step]), "every rank must hold the same reduced gradients"
    # single process: the two shards one after the other on rank 0's weights, same seeds; DDP gradient = their mean
    ts, undo = _use_sim_backend()
    try:
        for step in range(2):
            acc = None
            for rank in range(2):
                m, nc, N = _wrapper_model(tmp, seed=0)
                torch.manual_seed(50 + 10 * step + rank)
                ts.loss_step(m, _shard(nc, N, 2 * rank, 2 * rank + 2), backward=True)
                g = m._dsc_flat.G.clone()
                acc = g if acc is None else acc + g
            want = acc / 2
            err = float((r0["G%d" % step] - want).norm() / want.norm())
            assert err < 1e-5, (step, err)
    finally:
        undo()


# ---------------------------------------------------------------------------------------------------------------------
# DSC_DDP_FLUSH=auto (the default with more than one rank, round 6): the runner tries the three flush schedules on the first training
# steps and every rank must settle on the SAME one -- decided on the max over ranks, not on a rank's own clock.
# ---------------------------------------------------------------------------------------------------------------------
_FAKE_MS = ({"single": 30.0, "block": 25.0, "thirds": 28.0},      # rank 0 alone would keep "block"
            {"single": 24.0, "block": 29.0, "thirds": 26.0})      # rank 1 alone would keep "single"; max over ranks: thirds 28 < block 29 < single 30


def _tuner_worker(rank, ws, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("DSC_DDP_FLUSH", None)                      # default: auto with world > 1
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    torch.set_num_threads(2)
    ts, _ = _use_sim_backend()
    ts.ScheduleTuner.WARM, ts.ScheduleTuner.TIMED = 1, 1       # two steps per schedule instead of six
    ts.PlanRunner.tuner_interval = staticmethod(lambda schedule: _FAKE_MS[rank][schedule] * 1e-3)
    m, nc, N = _wrapper_model(tmp, seed=rank)
    s = _shard(nc, N, 2 * rank, 2 * rank + 2)
    used = []
    for step in range(8):
        torch.manual_seed(70 + step)
        ts.loss_step(m, s, backward=True)
        used.append(m._dsc_plan_runner.tuner.current())
    tuner = m._dsc_plan_runner.tuner
    torch.save({"choice": tuner.choice, "totals": dict(tuner.totals), "used": used, "G": m._dsc_flat.G.clone(),
                "plans": len(m._dsc_plan_runner.plans)}, os.path.join(tmp, "tuner_r%d.pt" % rank))
    dist.destroy_process_group()


def test_auto_flush_schedule_is_agreed_on_the_slowest_rank(tmp_path):
    tmp = str(tmp_path)
    mp.spawn(_tuner_worker, args=(2, _free_port(), tmp), nprocs=2, join=True)
    r0, r1 = torch.load(os.path.join(tmp, "tuner_r0.pt")), torch.load(os.path.join(tmp, "tuner_r1.pt"))
    assert r0["choice"] == r1["choice"] == "thirds", (r0["choice"], r1["choice"])
    assert r0["totals"] == r1["totals"] and abs(r0["totals"]["single"] - 0.030) < 1e-9 and abs(r0["totals"]["block"] - 0.029) < 1e-9
    assert r0["used"] == r1["used"] == ["single", "single", "block", "block", "thirds", "thirds", "thirds", "thirds"], r0["used"]
    assert torch.equal(r0["G"], r1["G"]) and r0["plans"] == 3          # one plan per schedule tried; the gradients stay in step


def test_flush_schedule_names(monkeypatch):
    from diffuscene_amd import train_step as ts
    monkeypatch.delenv("DSC_DDP_FLUSH", raising=False)
    assert ts.ddp_flush_schedule(1) == "single" and ts.ddp_flush_schedule(8) == "auto"
    monkeypatch.setenv("DSC_DDP_FLUSH", "end")
    monkeypatch.setattr(ts, "_warned_end", False)
    with pytest.warns(UserWarning, match="deprecated"):
        assert ts.ddp_flush_schedule(8) == "single"
    monkeypatch.setenv("DSC_DDP_FLUSH", "bogus")
    with pytest.raises(ValueError):
        ts.ddp_flush_schedule(2)


def test_failed_capture_is_not_retried_every_step(tmp_path, monkeypatch):
    """after ONE failed hipGraph capture the runner stays eager (sticky ``eager_only``); it must not try to capture
    again -- a warning, a device synchronize and a partial capture -- on every later step."""
    ts, undo = _use_sim_backend()
    calls = []

    def failing_capture(plan, reducer, device):
        calls.append(1)
        return None
    monkeypatch.setattr(ts, "_capture", failing_capture)
    try:
        m, nc, N = _wrapper_model(str(tmp_path), seed=0)
        s = _shard(nc, N, 0, 2)
        losses = []
        for step in range(4):
            torch.manual_seed(7)
            loss, _, ent = ts.loss_step(m, s, backward=True)
            losses.append(float(loss))
        assert len(calls) == 1, "capture attempted %d times" % len(calls)
        assert ent["graph"] is None and ent["eager_only"] is True and ent["warm"] == 4
        assert max(losses) - min(losses) < 1e-6 * abs(losses[0])          # same seed, same (eager) launches
    finally:
        undo()


# ---------------------------------------------------------------------------------------------------------------------
# world_size 4, UNEQUAL last shard (7 scenes -> 2 / 2 / 2 / 1), the data-parallel default schedule of round 4 (the pending weight-
# gradient group is launched whenever it holds a third of G -> 3 grouped launches, segments cut at the bucket launches): every rank
# builds the plan of ITS batch size; the bucket -> launch schedule, the order of the collectives and the reduced gradients must
# agree on all ranks, and the result must be the sum of the ranks' local gradients (each scaled 1 / (B_rank * world), i.e. the mean
# over ranks of the per-rank batch means -- what torch DDP computes for a ragged last batch as well).
# ---------------------------------------------------------------------------------------------------------------------
_SHARDS4 = [(0, 2), (2, 4), (4, 6), (6, 7)]


def _plan_setup_third(B, N, seed, tmp, grad_scale):
    holder, flat, _ = _plan_setup(B, N, seed, tmp, grad_scale, per_block=False)
    from plan_sim import SimBackend
    from diffuscene_amd.train_plan import TrainPlan
    from diffuscene_amd._lib import SS_PER_SLOT
    from diffuscene_amd.networks.diffusion_ddpm import GaussianDiffusion  # noqa: F401
    net = holder.net
    import contextlib
    import io
    import json
    from oracle import weights as W
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    stats = os.path.join(tmp, "stats3_%d.txt" % os.getpid())
    with open(stats, "w") as f:
        json.dump(W.DATASET_STATS, f)
    with contextlib.redirect_stdout(io.StringIO()):
        dp = DiffusionPoint(net, dict(objectness_dim=0, class_dim=22, angle_dim=2, objfeat_dim=32), time_num=1000,
                            model_mean_type="v", loss_separate=True, loss_iou=True, train_stats_file=stats)
    tb = {n: getattr(dp.diffusion, n) for n in dp.diffusion._TABLE_NAMES}

    def make_plan():
        return TrainPlan(net, flat, dp.diffusion, B, N, SS_PER_SLOT, 128, 0, 0, SimBackend(), ctx_param=holder.positional_embedding,
                         tables=tb, grad_scale=grad_scale, tn_flush_floats=flat.numel // 3)
    return holder, flat, make_plan


def _world4_worker(rank, ws, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=ws, timeout=datetime.timedelta(seconds=600))
    torch.set_num_threads(2)
    from diffuscene_amd import ddp
    from diffuscene_amd.train_step import _capture
    from oracle import weights as W
    N = 5
    lo, hi = _SHARDS4[rank]
    Bl = hi - lo
    holder, flat, make_plan = _plan_setup_third(Bl, N, seed=rank, tmp=tmp, grad_scale=1.0 / (Bl * ws))
    ddp.broadcast_parameters(holder)
    plan = make_plan()
    x0 = W.synth_scene_batch(7, N, 22, 32, seed=11)
    noise = W.synth_noise((7, N, 62), 12)
    t = torch.tensor([5, 300, 650, 999, 17, 480, 731])
    plan.x0.copy_(x0[lo:hi]); plan.noise.copy_(noise[lo:hi]); plan.t.copy_(t[lo:hi])
    # local gradient of this rank's shard (no exchange)
    plan.run_forward()
    plan.run_backward()
    g_local = flat.G.clone()
    red = ddp.FlatGradientReducer(flat, plan, n_buckets=8)
    sched = sorted((k, tuple(v)) for k, v in red.at_launch.items())
    sg = _capture(plan, red, torch.device("cpu"))
    for p in flat.params:                      # every gradient must be WRITTEN by the plan (alignment gaps of G stay 0)
        flat.grad_view(p).fill_(float("nan"))
    sg.replay()
    assert red.finish() == len(red.buckets) >= 8
    torch.save({"G": flat.G.clone(), "g_local": g_local, "order": list(red.last_order), "sched": sched, "segments": sg.segments,
                "n_bwd": len(plan.bwd), "tn_launches": plan.be.counts.get("gemm_tn_grouped", 0), "B": Bl,
                "before_last": red.launched_during_backward}, os.path.join(tmp, "w4_r%d.pt" % rank))
    dist.destroy_process_group()


def test_world4_unequal_last_shard_schedule_and_gradients(tmp_path):
    tmp = str(tmp_path)
    mp.spawn(_world4_worker, args=(4, _free_port(), tmp), nprocs=4, join=True)
    r = [torch.load(os.path.join(tmp, "w4_r%d.pt" % k)) for k in range(4)]
    assert [x["B"] for x in r] == [2, 2, 2, 1]
    # one launch list structure and one bucket schedule on every rank, whatever its batch size: the collectives match by construction
    assert len({x["n_bwd"] for x in r}) == 1 and len({tuple(x["sched"]) for x in r}) == 1 and len({tuple(x["order"]) for x in r}) == 1
    assert all(x["tn_launches"] == 3 for x in r), [x["tn_launches"] for x in r]
    assert all(len(x["segments"]) >= 3 for x in r) and all(x["before_last"] >= 4 for x in r)       # buckets leave before the last segment
    for k in range(1, 4):
        assert torch.equal(r[0]["G"], r[k]["G"]), "every rank must hold the same reduced gradients"
    want = sum(x["g_local"].double() for x in r)
    got = r[0]["G"].double()
    assert torch.isfinite(got).all()
    assert float((got - want).norm() / want.norm()) < 1e-6
