"""Dataflow of the static training plan (diffuscene_amd/train_plan.py) checked on the CPU: the plan is lowered by the torch
backend of tests/plan_sim.py and every parameter gradient it leaves in the flat buffer G is compared with torch.autograd over
the oracle (evaluated in float64).  Kernel numerics are covered by the GPU tests; this test pins WHICH buffers each launch
reads/writes, the accumulation of multi-consumer gradients and the slices of G."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import ref_torch as R          # noqa: E402
from oracle import weights as W            # noqa: E402


def _build(kw, N, tmp_path, arrange=False, ctx_dim=128):
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    import contextlib
    import io
    stats = os.path.join(str(tmp_path), "dataset_stats.txt")
    with open(stats, "w") as f:
        json.dump(W.DATASET_STATS, f)
    with contextlib.redirect_stdout(io.StringIO()):
        net = Unet1D(**kw)
        sd = W.synth_state_dict(kw)
        net.load_state_dict(sd)
        cfg = dict(objectness_dim=0, class_dim=kw["class_dim"], angle_dim=2, objfeat_dim=32, room_arrange_condition=arrange)
        dp = DiffusionPoint(net, cfg, time_num=1000, model_mean_type="v", loss_separate=True, loss_iou=not arrange,
                            train_stats_file=stats)

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.positional_embedding = torch.nn.Parameter(W.synth_condition(1, N, ctx_dim, 0)[0].clone())
            self.net = net
    return Holder(), dp.diffusion, sd


def _oracle_grads(sd, kw, diff_cfg, x0, t, noise, cond, cross, arrange):
    """fp64 autograd over the oracle: loss = losses.mean(); returns {name: grad}, d cond, d cross, losses"""
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
        cond64 = cond.double().requires_grad_(True)
        cross64 = cross.double().requires_grad_(True) if cross is not None else None
        tb = R.schedule_tables(1e-4, 0.02, 1000, "v")
        den = lambda xt, tt: R.unet1d_forward(sd64, kw, xt, tt, cond64, cross64)       # noqa: E731
        if arrange:
            lw, _ = R.p_losses_arrange(tb, den, x0.double(), t, noise.double())
        else:
            lw, _, _ = R.p_losses(tb, den, x0.double(), t, noise.double(), R.dims_from_kwargs(kw), True, True, W.DATASET_STATS)
        lw.mean().backward()
        return ({k: v.grad for k, v in sd64.items()}, cond64.grad, cross64.grad if cross64 is not None else None,
                lw.detach())
    finally:
        torch.set_default_dtype(old)


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("case", ["uncond_slot", "text_token", "arrange_token", "uncond_slot_ddp", "uncond_slot_unfused",
                                  "uncond_slot_ddp_third"])
def test_plan_gradients_match_autograd(case, tmp_path):
    """Cases: conditioning modes; `_ddp` = per-block gradient flushes; `_ddp_third` = the data-parallel default (the pending
    weight-gradient group is launched whenever it holds a third of G); `_unfused` = no activation epilogues (what the product backend
    plans for launches its split kernel does not take: GEMM + activation launch, activation-backward launch) -- every other case fuses
    them (pre-activation stored by the producing GEMM, act' applied by the consumer's input-gradient GEMM)."""
    from plan_sim import SimBackend
    from diffuscene_amd.flat import FlatStorage
    from diffuscene_amd.train_plan import TrainPlan
    from diffuscene_amd._lib import SS_PER_SLOT, SS_PER_TOKEN
    torch.manual_seed(0)
    B, N = 3, 6
    arrange = case.startswith("arrange")
    text = case.startswith("text")
    kw = dict(W.REARRANGE_LIVING if arrange else (W.TEXT_BEDROOM if text else W.UNCOND_BEDROOM))
    ctx_dim = 512 if arrange else 128
    holder, diff, sd = _build(kw, N, tmp_path, arrange=arrange, ctx_dim=ctx_dim)
    flat = FlatStorage(holder)
    assert flat.valid()
    tb = {n: getattr(diff, n) for n in diff._TABLE_NAMES}
    slot = case.startswith("uncond_slot")
    L = 5 if text else 0
    be = SimBackend()
    if case.endswith("unfused"):
        be.fuse_rows = 10 ** 9
    plan = TrainPlan(holder.net, flat, diff, B, N, SS_PER_SLOT if slot else SS_PER_TOKEN, ctx_dim, L, 512 if text else 0,
                     be, per_block_grads=case.endswith("ddp"),
                     ctx_param=holder.positional_embedding if slot else None, tables=tb,
                     tn_flush_floats=flat.numel // 3 if case.endswith("ddp_third") else None)
    if case.endswith("unfused"):
        assert "gemm_fused_act" not in be.counts and be.counts["act"] == 15 and be.counts["act_bwd"] >= 14
    elif arrange:                                       # no per-attribute encoder / decoder MLPs: only the time MLP's activations
        assert be.counts["gemm_fused_act"] >= 3
    else:
        assert be.counts["gemm_fused_act"] >= 20 and be.counts["act"] == 4
    if case.endswith("ddp_third"):
        assert be.counts["gemm_tn_grouped"] == 3
    elif not case.endswith("ddp"):
        assert be.counts["gemm_tn_grouped"] == 1, "single GPU: ONE grouped weight-gradient launch at the end of the backward"
    C = kw["channels"]
    if arrange:
        x0 = torch.rand(B, N, C) * 2 - 1
    else:
        x0 = W.synth_scene_batch(B, N, kw["class_dim"], 32, seed=3)
    noise = W.synth_noise((B, N, C), 5)
    t = torch.tensor([17, 400, 980])
    if slot:
        cond = holder.positional_embedding.detach()[None].expand(B, N, ctx_dim)
    else:
        cond = W.synth_condition(B, N, ctx_dim, 1, shared=False) if ctx_dim == 128 else torch.randn(B, N, ctx_dim)
        plan.ctx_in.t.copy_(cond.reshape(B * N, ctx_dim))
    cross = None
    if text:
        cross = W.synth_text_condition(B, L, 512, 2)
        plan.cross_in.t.copy_(cross.reshape(B * L, 512))
    plan.x0.copy_(x0); plan.noise.copy_(noise); plan.t.copy_(t)
    flat.G.fill_(float("nan"))                 # every gradient must be WRITTEN by the plan, not accumulated
    plan.run_forward()
    plan.run_backward()

    ref, dcond, dcross, lw = _oracle_grads(sd, kw, None, x0, t, noise, cond, cross, arrange)
    assert _rel(plan.losses, lw) < 1e-5
    bad = []
    for name, p in holder.net.named_parameters():
        g = flat.grad_view(p)
        assert torch.isfinite(g).all(), "gradient of %s was not written" % name
        r = _rel(g.reshape(-1), ref[name].reshape(-1))
        if r > 2e-4:
            bad.append((name, r))
    assert not bad, bad[:10]
    if slot:
        # shared instance embedding: d cond summed over the batch lands in the parameter's slice of G
        assert _rel(flat.grad_view(holder.positional_embedding), dcond.sum(0)) < 2e-4
    else:
        assert _rel(plan.d_ctx, dcond.reshape(B * N, ctx_dim)) < 2e-4
    if text:
        assert _rel(plan.d_cross, dcross.reshape(B * L, 512)) < 2e-4
    # bucket schedule: every bucket of G is finished by exactly one launch index (or by the autograd side: key None)
    buckets = flat.buckets(8)
    sched = plan.bucket_schedule(buckets)
    assert sorted(b for bs in sched.values() for b in bs) == list(range(len(buckets)))
    if case.endswith("ddp"):
        idx = [i for i in sched if i is not None]
        assert len(set(idx)) >= 4, "per-block gradients should finish buckets at different points of the backward"
    if case.endswith("ddp_third"):
        idx = [i for i in sched if i is not None]
        assert len(set(idx)) >= 3, "three grouped launches must finish buckets at three points of the backward"


def test_plan_at_the_full_text_batch_matches_the_reference_golden(tmp_path, golden_dir):
    """BASELINE configs[3] at its own batch (B=128, N=12, L=32): the plan on the torch backend against outputs of the REAL reference
    (tests/golden/fullbatch.npz, oracle/make_golden_fullbatch.py): per-scene losses, the nine logged terms, 16 gradient norms and
    the gradient that reaches the text features.  tests/test_gpu_fullbatch.py holds the HIP backend to the same file."""
    import numpy as np
    from plan_sim import SimBackend
    from diffuscene_amd._lib import SS_PER_SLOT
    from diffuscene_amd.flat import FlatStorage
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    from diffuscene_amd.train_plan import TrainPlan
    from oracle.make_golden_fullbatch import fullbatch_inputs
    import contextlib
    import io
    g = np.load(os.path.join(golden_dir, "fullbatch.npz"))
    names = json.load(open(os.path.join(golden_dir, "grad_names_fullbatch.json")))["text"]
    kw, x, t, cond, cross, noise, _ = fullbatch_inputs("text")
    stats = os.path.join(str(tmp_path), "dataset_stats.txt")
    with open(stats, "w") as f:
        json.dump(W.DATASET_STATS, f)
    with contextlib.redirect_stdout(io.StringIO()):
        net = Unet1D(**kw)
        net.load_state_dict(W.synth_state_dict(kw))
        diff = DiffusionPoint(net, dict(objectness_dim=0, class_dim=22, angle_dim=2, objfeat_dim=32), time_num=1000,
                              model_mean_type="v", loss_separate=True, loss_iou=True, train_stats_file=stats).diffusion
    flat = FlatStorage(net)
    B, N, C = x.shape
    L = cross.shape[1]
    tb = {n: getattr(diff, n).float() for n in diff._TABLE_NAMES}
    plan = TrainPlan(net, flat, diff, B, N, SS_PER_SLOT, 128, L, 512, SimBackend(), tables=tb)
    plan.x0.copy_(x); plan.noise.copy_(noise); plan.t.copy_(t)
    plan.ctx_in.t.copy_(cond[0])
    plan.cross_in.t.copy_(cross.reshape(B * L, 512))
    flat.G.fill_(float("nan"))
    flat.zero_head()
    plan.run_forward()
    plan.run_backward()
    assert _rel(plan.losses, torch.from_numpy(g["text.losses"])) < 1e-6
    keys = ('loss.bbox', 'loss.trans', 'loss.size', 'loss.angle', 'loss.class', 'loss.object', 'loss.objfeat', 'loss.liou',
            'loss.bbox_iou')
    means = plan.parts.mean(dim=0)
    for i, k in enumerate(keys):
        assert abs(float(means[i]) - float(g["text." + k])) <= 1e-5 * max(1.0, abs(float(g["text." + k]))), k
    params = dict(net.named_parameters())
    gn = np.array([float(flat.grad_view(params[k]).norm()) for k in names])
    ref = g["text.grad_norms"]
    assert (np.abs(gn - ref) / np.maximum(ref, 1e-3 * ref.max())).max() < 1e-4
    assert abs(float(plan.d_cross.norm()) - float(g["text.d_cross_norm"])) <= 1e-4 * float(g["text.d_cross_norm"])


@pytest.mark.parametrize("variant", ["eps", "x0", "objectness", "objfeat64", "legacy", "flat", "arrange_sep0", "arrange_sep1"])
def test_plan_variants_match_the_reference_goldens(variant, tmp_path, golden_dir):
    """The layouts and loss branches no shipped `_v` YAML reaches ('eps' / 'x0' targets, an objectness channel, the 64-d shape code,
    the reference's constructor defaults, loss_separate = False, the re-arrangement loss) -- the plan on the torch backend against
    outputs of the REAL reference (tests/golden/meantypes.npz, oracle/make_golden_meantypes.py): per-scene losses, logged terms and
    the gradient norm of every parameter.  tests/test_gpu_meantypes.py holds the HIP backend to the same file."""
    import contextlib
    import io
    import numpy as np
    from plan_sim import SimBackend
    from diffuscene_amd._lib import SS_PER_SLOT, SS_PER_TOKEN
    from diffuscene_amd.flat import FlatStorage
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    from diffuscene_amd.train_plan import TrainPlan
    from oracle.make_golden import case_inputs
    g = np.load(os.path.join(golden_dir, "meantypes.npz"))
    arrange = variant.startswith("arrange")
    kw, x, t, cond, _ = case_inputs("rearrange_living" if arrange else "uncond_bedroom")
    B, N = x.shape[:2]
    nc = kw.get("class_dim", 21)
    cfg = dict(objectness_dim=kw.get("objectness_dim", 1), class_dim=nc, angle_dim=kw.get("angle_dim", 1), objfeat_dim=kw.get("objfeat_dim", 0))
    mean_type, separate, iou, noise_tag = "v", True, not arrange, "train_noise_arr" if arrange else "train_noise"
    if variant in ("eps", "x0"):
        mean_type = variant
    elif variant == "objectness":
        kw = dict(kw, objectness_dim=1, channels=kw["channels"] + 1)
        x = torch.cat([x[:, :, :8 + nc], torch.where(x[:, :, 8 + nc - 1:8 + nc] > 0, -1.0, 1.0), x[:, :, 8 + nc:]], dim=-1).contiguous()
        cfg["objectness_dim"], noise_tag = 1, "train_noise_obj"
    elif variant == "objfeat64":
        kw = dict(kw, objfeat_dim=64, channels=8 + nc + 64)
        x = W.synth_scene_batch(B, N, nc, 64, seed=0)
        cfg["objfeat_dim"], noise_tag = 64, "train_noise_64"
    elif variant == "legacy":
        kw = dict(kw, objectness_dim=1, class_dim=21, angle_dim=1, objfeat_dim=0, channels=29)
        base = W.synth_scene_batch(B, N, 21, 0, seed=0)
        x = torch.cat([base[:, :, :6], torch.atan2(base[:, :, 7:8], base[:, :, 6:7]) / np.pi, base[:, :, 8:29],
                       torch.where(base[:, :, 28:29] > 0, -1.0, 1.0)], dim=-1).contiguous()
        cfg, noise_tag = dict(objectness_dim=1, class_dim=21, angle_dim=1, objfeat_dim=0), "train_noise_legacy"
    elif variant == "flat":
        separate, iou = False, False
    elif arrange:
        cfg["room_arrange_condition"] = True
        separate = variant.endswith("1")
    stats = os.path.join(str(tmp_path), "dataset_stats.txt")
    with open(stats, "w") as f:
        json.dump(W.DATASET_STATS, f)
    with contextlib.redirect_stdout(io.StringIO()):
        net = Unet1D(**kw)
        net.load_state_dict(W.synth_state_dict(kw))
        diff = DiffusionPoint(net, cfg, time_num=1000, model_mean_type=mean_type, loss_separate=separate, loss_iou=iou,
                              train_stats_file=stats if iou else None).diffusion
    noise = W.synth_noise(tuple(x.shape), 0, noise_tag)
    flat = FlatStorage(net)
    tb = {n: getattr(diff, n).float() for n in diff._TABLE_NAMES}
    plan = TrainPlan(net, flat, diff, B, N, SS_PER_TOKEN if arrange else SS_PER_SLOT, 512 if arrange else 128, 0, 0, SimBackend(), tables=tb)
    plan.x0.copy_(x); plan.noise.copy_(noise); plan.t.copy_(t)
    plan.ctx_in.t.copy_(cond.reshape(B * N, 512) if arrange else cond[0])
    flat.G.fill_(float("nan"))
    flat.zero_head()
    plan.run_forward()
    plan.run_backward()
    assert _rel(plan.losses, torch.from_numpy(g[variant + ".losses"])) < 2e-6
    keys = ('loss.bbox', 'loss.trans', 'loss.size', 'loss.angle', 'loss.class', 'loss.object', 'loss.objfeat', 'loss.liou', 'loss.bbox_iou')
    means = plan.parts.mean(dim=0)
    parts = {"loss.trans": means[1], "loss.angle": means[3]} if arrange else {k: means[i] for i, k in enumerate(keys)}
    for k, v in parts.items():
        want = float(g[variant + "." + k])
        assert abs(float(v) - want) <= 1e-5 * max(1.0, abs(want)), (variant, k, float(v), want)
    names = [k for k, _ in net.named_parameters()]
    params = dict(net.named_parameters())
    gn = np.array([float(flat.grad_view(params[k]).norm()) for k in names])
    ref = g[variant + ".grad_norms"]
    assert len(ref) == len(gn)
    e = np.abs(gn - ref) / np.maximum(ref, 1e-3 * ref.max())
    assert e.max() < 1e-3, (variant, names[int(e.argmax())], float(e.max()))


@pytest.mark.parametrize("case", ["noinst", "uncond"])
def test_wrapper_step_on_the_plan_matches_the_reference_wrapper_golden(case, tmp_path, golden_dir):
    """One ``get_loss`` of the REAL reference wrapper (tests/golden/wrapper.npz) reproduced on the CPU: OUR wrapper assembles the batch
    (``_loss_inputs``), the seeded generator gives t and the noise in the reference's order, the static plan runs on the torch backend.
    'noinst' is the un-conditioned network (condition None -> SS_NONE, a never-applied Linear(0, 1024) in every context block),
    'uncond' conditions on the learned positional embedding read in place.  (The fc_instance_condition variant needs the HIP ops of the
    wrapper-level MLP: tests/test_gpu_wrapper.py.)"""
    import contextlib
    import io
    import numpy as np
    from plan_sim import SimBackend
    from diffuscene_amd._lib import SS_NONE, SS_PER_SLOT
    from diffuscene_amd.flat import FlatStorage
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import DiffusionSceneLayout_DDPM
    from diffuscene_amd.train_plan import TrainPlan
    from oracle.make_golden_wrapper import SEED_LOSS, network_config, wrapper_batch, wrapper_state_dict
    g = np.load(os.path.join(golden_dir, "wrapper.npz"))
    stats = os.path.join(str(tmp_path), "dataset_stats.txt")
    with open(stats, "w") as f:
        json.dump(W.DATASET_STATS, f)
    cfg = network_config(case, stats)
    with contextlib.redirect_stdout(io.StringIO()):
        m = DiffusionSceneLayout_DDPM(cfg["class_dim"] + 1, None, cfg)
    m.load_state_dict(wrapper_state_dict(m))
    s, _ = wrapper_batch(case)
    with torch.no_grad():
        target, condition, cross = m._loss_inputs(s)
    assert cross is None and torch.equal(target, torch.from_numpy(g[case + ".target"]))
    assert (condition is None) == (case == "noinst")
    B, N, C = target.shape
    torch.manual_seed(SEED_LOSS)                         # the reference's draws: t (get_loss_iter), then the noise (p_losses)
    t = torch.randint(0, 1000, size=(B,))
    noise = torch.randn(target.shape)
    net, diff = m.diffusion.model, m.diffusion.diffusion
    flat = FlatStorage(m)
    tb = {n: getattr(diff, n).float() for n in diff._TABLE_NAMES}
    plan = TrainPlan(net, flat, diff, B, N, SS_NONE if condition is None else SS_PER_SLOT, 0 if condition is None else condition.shape[-1],
                     0, 0, SimBackend(), tables=tb, ctx_param=m.positional_embedding if case == "uncond" else None)
    plan.x0.copy_(target); plan.noise.copy_(noise); plan.t.copy_(t)
    flat.G.fill_(float("nan"))
    flat.zero_head()
    plan.run_forward()
    plan.run_backward()
    want = float(g[case + ".loss"])
    assert abs(float(plan.losses.mean()) - want) <= 2e-6 * abs(want), (case, float(plan.losses.mean()), want)
    keys = ('loss.bbox', 'loss.trans', 'loss.size', 'loss.angle', 'loss.class', 'loss.object', 'loss.objfeat', 'loss.liou', 'loss.bbox_iou')
    means = plan.parts.mean(dim=0)
    for i, k in enumerate(keys):
        w = float(g[case + ".part." + k])
        assert abs(float(means[i]) - w) <= 1e-5 * max(1.0, abs(w)), (case, k, float(means[i]), w)
    # every gradient of the denoiser the plan owns was written (the never-applied context Linear of 'noinst' is not the plan's: it
    # stays untouched, as its .grad stays None in the reference)
    never_applied = {id(q) for rb, kind in net.resblocks_in_order() if kind == "c" and case == "noinst" for q in rb.mlp[1].parameters()}
    for name, p in net.named_parameters():
        if p.numel() and id(p) not in never_applied:
            assert torch.isfinite(flat.grad_view(p)).all(), name
