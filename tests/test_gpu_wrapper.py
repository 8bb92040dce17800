"""GPU: WRAPPER-level parity against outputs of the REAL reference class
``DiffusionSceneLayout_DDPM`` (tests/golden/wrapper.npz, produced by oracle/make_golden_wrapper.py from the reference's own
module and its shipped YAML network sections): the target / condition assembly of ``get_loss`` (attribute cat, instance embedding,
partial mask, arrange slices, text projection), ``get_loss`` itself, ``train_on_batch`` (one Adam step) and ``validate_on_batch``,
``sample`` at B=4 and the post-filtered dicts of ``generate_layout`` / ``complete_scene`` / ``arrange_scene`` at batch_size 1 --
through OUR drop-in entry points, same ``state_dict``, same seeds.

RNG: the reference drew t / noise / x_T from torch's global CPU generator after ``torch.manual_seed``; the fixture below makes
``torch.randint`` / ``torch.randn`` draw from that same CPU generator and move the result to the device, so the ORDER and shapes of
the draws of the product are pinned too.  Tolerances: 1e-4 norm-relative and element-wise on tensors and losses (test_gpu_wide.check),
1e-3 on gradient / parameter-delta norms."""
import contextlib
import copy
import io
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import weights as W  # noqa: E402
from oracle.make_golden_wrapper import (B, CASES, N, PARTIAL_P, SAMPLE_T, SEED_LOSS, SEED_ONE, SEED_SAMPLE, SEED_TRAIN,  # noqa: E402
                                        fake_bert_features, network_config, sample_text_arg, wrapper_batch, wrapper_state_dict)

from test_gpu_wide import check, dev  # noqa: E402


@pytest.fixture
def cpu_rng(monkeypatch):
    real_randn, real_randint = torch.randn, torch.randint

    def randn(*size, **kw):
        d = kw.pop("device", None)
        out = real_randn(*size, **kw)
        return out if d is None else out.to(d)

    def randint(*a, **kw):
        d = kw.pop("device", None)
        out = real_randint(*a, **kw)
        return out if d is None else out.to(d)
    monkeypatch.setattr(torch, "randn", randn)
    monkeypatch.setattr(torch, "randint", randint)
    # the reverse loops take ``noise_fn=torch.randn`` as a DEFAULT ARGUMENT (bound when the function was defined, as in the reference):
    # hand them the CPU-generator form whenever the caller left the default
    from diffuscene_amd.networks import diffusion_ddpm as dd
    for name in ("p_sample_loop", "p_sample_loop_trajectory", "p_sample_loop_complete", "p_sample_loop_arrange"):
        orig = getattr(dd.GaussianDiffusion, name)

        def wrapped(self, *a, _orig=orig, **kw):
            if kw.get("noise_fn", real_randn) is real_randn:
                kw["noise_fn"] = randn
            return _orig(self, *a, **kw)
        monkeypatch.setattr(dd.GaussianDiffusion, name, wrapped)
    # the captured loop (the default since round 6) draws with the DEVICE generator inside the graph: the CPU-generator injection above pins
    # the ORDER and shapes of the product's draws on the eager loop (a patched torch.randn that copies from the host cannot be captured)
    monkeypatch.setenv("DSC_GRAPH", "0")


class _FakeBertCache:
    """text_cache.BertFeatureCache protocol (``batch(texts, device)``) over the golden generator's stand-in encoder."""

    def batch(self, texts, device):
        return fake_bert_features(list(texts)).to(device)


def _build(case, tmp_path, time_num=1000):
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import DiffusionSceneLayout_DDPM
    stats = tmp_path / "dataset_stats.txt"
    stats.write_text(json.dumps(W.DATASET_STATS))
    cfg = network_config(case, str(stats), time_num)
    if case == "text":
        cfg["text_bert_cached"] = True                        # the frozen encoder is not part of the path
    with contextlib.redirect_stdout(io.StringIO()):
        m = DiffusionSceneLayout_DDPM(cfg["class_dim"] + 1, None, cfg)
    return m, cfg


def _to_dev(s):
    return {k: (v.to(dev()) if torch.is_tensor(v) else v) for k, v in s.items()}


def _batch(case):
    s, x = wrapper_batch(case)
    if case == "text":
        s["desc_bert"] = fake_bert_features(s["description"])
    return _to_dev(s), x.to(dev())


@pytest.mark.parametrize("case", CASES)
def test_state_dict_layout_matches_reference_wrapper(case, golden_dir, tmp_path):
    keys = json.load(open(os.path.join(golden_dir, "wrapper_keys.json")))[case]
    m, _ = _build(case, tmp_path)
    ours = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    want = [kv for kv in keys if not kv[0].startswith("bertmodel.")]
    assert ours == want


@pytest.mark.parametrize("case", CASES)
def test_get_loss_inputs_and_loss(case, golden_dir, tmp_path, cpu_rng):
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import validate_on_batch
    g = np.load(os.path.join(golden_dir, "wrapper.npz"))
    m, cfg = _build(case, tmp_path)
    m.load_state_dict(wrapper_state_dict(m))
    m.to(dev())
    s, _ = _batch(case)
    with torch.no_grad():
        target, condition, cross = m._loss_inputs(s)
    check(target, g[case + ".target"], case + " diffusion target", tol=1e-6)
    if case + ".condition" in g.files:
        check(condition.expand(B, -1, -1) if condition.shape[0] != B else condition, g[case + ".condition"], case + " condition", tol=2e-5)
    else:
        assert condition is None                       # instance_condition false: the reference hands None to the diffusion too
    if case + ".cross" in g.files:
        check(cross, g[case + ".cross"], case + " condition_cross", tol=2e-5)
    else:
        assert cross is None
    part_keys = [k[len(case) + 6:] for k in g.files if k.startswith(case + ".part.")]
    assert part_keys
    # (1) the autograd entry point the reference's callers use; (2) validate_on_batch (the static plan where it applies)
    torch.manual_seed(SEED_LOSS)
    loss, parts = m.get_loss(s)
    want = float(g[case + ".loss"])
    assert abs(float(loss) - want) <= 1e-4 * abs(want), (case, float(loss), want)
    assert sorted(parts) == sorted(part_keys)
    for k in part_keys:
        w = float(g[case + ".part." + k])
        assert abs(float(parts[k]) - w) <= 1e-4 * max(abs(w), 1e-3), (case, k, float(parts[k]), w)
    from diffuscene_amd.stats_logger import StatsLogger
    StatsLogger.instance().clear()
    torch.manual_seed(SEED_LOSS)
    v = validate_on_batch(m, s, {"training": {"max_grad_norm": 10}})
    assert abs(v - want) <= 1e-4 * abs(want), (case, v, want)
    for k in part_keys:
        w = float(g[case + ".part." + k])
        assert abs(StatsLogger.instance()[k].value - w) <= 1e-4 * max(abs(w), 1e-3), (case, k)
    StatsLogger.instance().clear()


@pytest.mark.parametrize("case", CASES)
def test_train_on_batch_one_adam_step(case, golden_dir, tmp_path, cpu_rng):
    from diffuscene_amd.networks import optimizer_factory
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import train_on_batch
    from diffuscene_amd.stats_logger import StatsLogger
    g = np.load(os.path.join(golden_dir, "wrapper.npz"))
    names = json.load(open(os.path.join(golden_dir, "wrapper_keys.json")))[case + ".delta_params"]
    m, cfg = _build(case, tmp_path)
    m.load_state_dict(wrapper_state_dict(m))
    m.to(dev())
    s, _ = _batch(case)
    opt = optimizer_factory({"optimizer": "Adam", "lr": 0.0002, "weight_decay": 0.0}, filter(lambda p: p.requires_grad, m.parameters()))
    before = {k: p.detach().clone() for k, p in m.named_parameters() if k in names}
    StatsLogger.instance().clear()
    torch.manual_seed(SEED_TRAIN)
    ret = train_on_batch(m, opt, s, {"training": {"max_grad_norm": 10}})
    # every wrapper case -- the partial condition too (round 6) -- trains on the static plan, not on the autograd fallback
    from diffuscene_amd.train_step import plan_supported
    assert plan_supported(m) and getattr(m, "_dsc_plan_runner", None) is not None, case
    want = float(g[case + ".train.loss"])
    assert abs(ret - want) <= 1e-4 * abs(want), (case, ret, want)
    gn, wgn = StatsLogger.instance()["gradnorm"].value, float(g[case + ".train.gradnorm"])
    assert abs(gn - wgn) <= 1e-3 * wgn, (case, gn, wgn)
    StatsLogger.instance().clear()
    params = dict(m.named_parameters())
    grads = np.array([float(params[k].grad.norm()) for k in names])
    wg = g[case + ".train.grad_norms"]
    e = np.abs(grads - wg) / np.maximum(wg, 1e-3 * wg.max())
    assert e.max() < 1e-3, (case, names[int(e.argmax())], e.max())
    # Adam's first step moves every weight by ~lr * sign(g): the delta norm is lr * sqrt(numel) unless gradients vanish
    deltas = np.array([float((params[k].detach() - before[k]).norm()) for k in names])
    wd = g[case + ".train.delta_norms"]
    e = np.abs(deltas - wd) / np.maximum(wd, 1e-3 * wd.max())
    print("%s: Adam step delta-norm rel err max %.3g" % (case, e.max()))
    assert e.max() < 2e-3, (case, names[int(e.argmax())], e.max())


def _sample_kwargs(case, x):
    if case == "arrange":
        return {"input_boxes": x}
    if case == "partial":
        return {"partial_boxes": x[:, :PARTIAL_P].contiguous()}
    return {}


@pytest.mark.parametrize("case", CASES)
def test_sample_and_layout_entry_points(case, golden_dir, tmp_path, cpu_rng):
    g = np.load(os.path.join(golden_dir, "wrapper.npz"))
    m, cfg = _build(case, tmp_path, time_num=SAMPLE_T)
    m.load_state_dict(wrapper_state_dict(m))
    m.to(dev())
    if case == "text":
        m.attach_bert_cache(_FakeBertCache())
    _, x = _batch(case)
    C = cfg["point_dim"]
    room = torch.zeros(B, 1, 64, 64, device=dev())
    text = sample_text_arg(case)
    if torch.is_tensor(text):
        text = text.to(dev())
    quiet = contextlib.redirect_stdout(io.StringIO())
    torch.manual_seed(SEED_SAMPLE)
    with torch.no_grad(), quiet:
        y = m.sample(room, N, C, batch_size=B, text=text, clip_denoised=True, **_sample_kwargs(case, x))
    check(y, g[case + ".sample"], "%s sample(B=%d, T=%d)" % (case, B, SAMPLE_T))
    one_text = None if text is None else text[:1]
    torch.manual_seed(SEED_ONE)
    with quiet:
        if case == "arrange":
            d = m.arrange_scene(room[:1], N, C, x[:1], batch_size=1, clip_denoised=True)
        elif case == "partial":
            d = m.complete_scene(room[:1], N, C, x[:1, :PARTIAL_P].contiguous(), batch_size=1, clip_denoised=True)
        else:
            d = m.generate_layout(room[:1], N, C, batch_size=1, text=one_text, clip_denoised=True)
    want = {k[len(case) + 8:]: g[k] for k in g.files if k.startswith(case + ".layout.")}
    assert sorted(d) == sorted(want)
    for k, v in d.items():
        assert v.device.type == "cpu" and tuple(v.shape) == tuple(want[k].shape), (case, k, tuple(v.shape), want[k].shape)
        if v.numel():
            check(v, want[k], "%s layout %s" % (case, k))
    if case == "uncond":
        torch.manual_seed(SEED_SAMPLE + 10)
        with torch.no_grad(), quiet:
            y = m.sample(room, N, C, batch_size=B, partial_boxes=x[:, :PARTIAL_P].contiguous(), clip_denoised=True)
        check(y, g["uncond.complete"], "uncond completion sample")
        assert torch.equal(y[:, :PARTIAL_P], x[:, :PARTIAL_P])
        torch.manual_seed(SEED_ONE + 10)
        with quiet:
            d = m.generate_layout(room[:1], N, C, batch_size=1, clip_denoised=False, keep_empty=True)
        for k, v in d.items():
            check(v, g["uncond.layout_noclip_keep." + k], "uncond layout (unclipped, keep_empty) " + k)
        # the progressive entry point (reference :320-333): the loop's trajectory, every 5th step post-filtered on its own
        steps = json.load(open(os.path.join(golden_dir, "wrapper_keys.json")))["uncond.progressive_steps"]
        torch.manual_seed(SEED_ONE + 20)
        with quiet:
            traj = m.generate_layout_progressive(room[:1], N, C, batch_size=1, ret_traj=True, clip_denoised=True, num_step=5)
        assert sorted(traj) == steps == [0, 5, 10, 15, 20]
        for kt, d in traj.items():
            assert set(d) == {k.rsplit(".", 1)[1] for k in g.files if k.startswith("uncond.progressive.%d." % kt)}
            for k, v in d.items():
                assert v.device.type == "cpu"
                check(v, g["uncond.progressive.%d.%s" % (kt, k)], "uncond generate_layout_progressive step %d %s" % (kt, k))
