"""GPU: parity at the BASELINE.json shapes against outputs of the REAL reference (tests/golden/wide.npz, produced by
oracle/make_golden_wide.py): N=80 forward / training loss with every parameter gradient, completion with 20 given objects,
5-channel re-arrangement at N=80, text cross-attention with L=32 tokens, p_sample_loop_trajectory.

Two criteria per tensor: the norm-relative one of the north star (max|a-b| / max|b| < 1e-4) AND an element-wise one,
|a-b| <= 1e-4 * max(|b|, floor) with floor = 5e-2 * max|b|, i.e. an absolute error below 5e-6 of the tensor's range for the small
elements (below that, fp32 summation-order noise -- the reference itself moves by 2.3e-7 of the range between two thread
counts on one forward, SURVEY.md 8c, and our measured forward error is ~2e-6 of the range -- dominates any relative measure).
Gradients are compared with the reference's fp32 CPU gradients at 1e-3 (their own distance to an fp64 evaluation is up to
2.6e-4, tests/test_gpu_train.py) and with each other (plan vs autograd driver) at 2e-5."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import weights as W  # noqa: E402
from oracle.make_golden import noise_list  # noqa: E402
from oracle.make_golden_wide import WIDE, wide_inputs  # noqa: E402

TOL = 1e-4


def dev():
    return torch.device("cuda:0")


def check(a, b, what, tol=TOL):
    """norm-relative AND element-wise (relative to max(|b|, 5 % of the range)) distance below `tol`.  On LONG chains the element-wise figure is
    the chain's own sensitivity to rounding, not a kernel property: the T = 1000 chain sits at 8.0e-5 under the split arithmetic and 7.6e-5 under
    exact f32 (a different but equally correct summation order could move either across 1e-4 on a few elements).  What pins the
    arithmetic independently of that amplification is the distance BETWEEN the two arithmetics on the same chain
    (test_gpu_chain_split.py::test_two_hundred_step_chain_at_b256: 1.75e-7 norm-relative, 1.4e-6 max abs)."""
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bmax = float(b.abs().max())
    r = float((a - b).abs().max() / bmax)
    ew = float(((a - b).abs() / torch.clamp(b.abs(), min=5e-2 * bmax)).max())
    strict = float((((a - b).abs() <= 1e-4 * torch.clamp(b.abs(), min=1e-3)).double()).mean())
    line = "%s: norm-relative %.3g, element-wise %.3g (%.2f%% of elements within 1e-4*max(|b|,1e-3))" % (what, r, ew, 100 * strict)
    print(line)
    if os.environ.get("DSC_PARITY_LOG"):              # tools/gpu_round.sh: the measured distances of a run, kept under profiles/
        with open(os.environ["DSC_PARITY_LOG"], "a") as f:
            f.write(line + "\n")
    assert r < tol and ew < tol, (what, r, ew)


_NETS = {}


def build(name, **diff_kwargs):
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    kw = WIDE[name][0]
    if name not in _NETS:
        net = Unet1D(**kw)
        net.load_state_dict(W.synth_state_dict(kw))
        _NETS[name] = net.to(dev())
    cfg = dict(objectness_dim=0, class_dim=kw["class_dim"], angle_dim=2, objfeat_dim=32)
    cfg.update(diff_kwargs.pop("config_extra", {}))
    return _NETS[name], DiffusionPoint(_NETS[name], cfg, **diff_kwargs)


def _replay(seq):
    from diffuscene_amd.sampler import NoiseReplay
    return NoiseReplay(torch.stack(seq).to(dev()))


def test_forward_at_baseline_shapes(golden_dir):
    g = np.load(os.path.join(golden_dir, "wide.npz"))
    for name in ("living80", "text32"):
        kw, x, t, cond, cross = wide_inputs(name)
        net, _ = build(name)
        with torch.no_grad():
            out = net(x.to(dev()), t.to(dev()), cond.to(dev()), cross.to(dev()) if cross is not None else None)
        check(out, g[name + ".forward"], name + " forward")
    kw, x, t, cond, _ = wide_inputs("arrange80")
    net, _ = build("arrange80")
    with torch.no_grad():
        out = net(x.to(dev()), torch.tensor([91, 468], device=dev()), cond.to(dev()), None)
    check(out, g["arrange80.forward"], "arrange80 forward")


def test_completion_n80_p20_eager_and_graph(golden_dir):
    from diffuscene_amd.sampler import NoiseReplay
    g = np.load(os.path.join(golden_dir, "wide.npz"))
    kw, x, t, cond, _ = wide_inputs("living80")
    B, N, C = x.shape
    net, diff = build("living80", time_num=50, model_mean_type="v")
    shapes = [(B, N, C)]
    for _ in range(50):
        shapes += [(B, 20, C), (B, N, C)]
    seq = noise_list(shapes, 41, "complete80_")
    main = torch.stack([seq[0]] + seq[2::2]).to(dev())
    part = torch.stack(seq[1::2]).to(dev())
    partial = x[:, :20, :].contiguous().to(dev())
    res = []
    for graph in (False, True):
        with torch.no_grad():
            res.append(diff.complete_samples((B, N, C), dev(), condition=cond.to(dev()), noise_fn=NoiseReplay(main, part),
                                             clip_denoised=True, partial_boxes=partial, graph=graph))
        check(res[-1], g["complete80.T50"], "completion N=80 P=20 T=50 (graph=%s)" % graph)
    assert torch.equal(res[0], res[1])
    assert torch.equal(res[1][:, :20], partial)                 # the given objects come back untouched


def test_arrange_n80_and_text_l32_chains(golden_dir):
    g = np.load(os.path.join(golden_dir, "wide.npz"))
    kw, x, t, cond, _ = wide_inputs("arrange80")
    B, N = x.shape[:2]
    full = W.synth_scene_batch(B, N, 25, 32, 45)
    net, diff = build("arrange80", time_num=50, model_mean_type="v", config_extra={"room_arrange_condition": True})
    for graph in (False, True):
        with torch.no_grad():
            s = diff.arrange_samples((B, N, 65), dev(), condition=cond.to(dev()),
                                     noise_fn=_replay(noise_list([(B, N, 5)] * 51, 42, "arrange80_")), clip_denoised=True,
                                     input_boxes=full.to(dev()), graph=graph)
        check(s, g["arrange80.T50"], "re-arrangement N=80 T=50 (graph=%s)" % graph)
    kw, x, t, cond, cross = wide_inputs("text32")
    net, diff = build("text32", time_num=20, model_mean_type="v")
    for graph in (False, True):
        with torch.no_grad():
            s = diff.gen_samples(tuple(x.shape), dev(), condition=cond.to(dev()), condition_cross=cross.to(dev()),
                                 noise_fn=_replay(noise_list([tuple(x.shape)] * 21, 43, "text32_")), graph=graph)
        check(s, g["text32.T20"], "text L=32 T=20 (graph=%s)" % graph)


def test_trajectory_matches_reference(golden_dir):
    """p_sample_loop_trajectory (:373-398): x_T, the first step and every freq-th state."""
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    g = np.load(os.path.join(golden_dir, "wide.npz"))
    kw = dict(W.UNCOND_BEDROOM)
    net = Unet1D(**kw)
    net.load_state_dict(W.synth_state_dict(kw))
    net.to(dev())
    diff = DiffusionPoint(net, dict(objectness_dim=0, class_dim=22, angle_dim=2, objfeat_dim=32), time_num=50,
                          model_mean_type="v")
    cond = W.synth_condition(2, 12, 128, 0).contiguous().to(dev())
    with torch.no_grad():
        imgs = diff.gen_sample_traj((2, 12, 62), dev(), freq=10, condition=cond,
                                    noise_fn=_replay(noise_list([(2, 12, 62)] * 51, 44, "traj_")), clip_denoised=True)
    assert len(imgs) == g["traj.T50"].shape[0] == 7
    check(torch.stack(imgs), g["traj.T50"], "trajectory T=50 freq=10")


def test_training_loss_and_all_gradients_at_n80(golden_dir, tmp_path, monkeypatch):
    """p_losses (+IoU) at N=80 through BOTH drivers (static plan, autograd): losses / logged terms vs the reference, the
    gradient norm of every one of the 442 parameters within 1e-3 of the reference's fp32 CPU result (whose own distance to
    an fp64 evaluation is up to 2.6e-4, tests/test_gpu_train.py) and committed gradient slices within 1e-4."""
    from diffuscene_amd.flat import FlatStorage
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    from diffuscene_amd.train_plan import HipBackend, TrainPlan
    from diffuscene_amd._lib import SS_PER_TOKEN
    g = np.load(os.path.join(golden_dir, "wide.npz"))
    names = json.load(open(os.path.join(golden_dir, "grad_names_living80.json")))
    kw, x, t, cond, _ = wide_inputs("living80")
    stats = tmp_path / "dataset_stats.txt"
    stats.write_text(json.dumps(W.DATASET_STATS))
    net = Unet1D(**kw)
    net.load_state_dict(W.synth_state_dict(kw))
    net.to(dev())
    cfg = dict(objectness_dim=0, class_dim=25, angle_dim=2, objfeat_dim=32)
    diff = DiffusionPoint(net, cfg, time_num=1000, model_mean_type="v", loss_separate=True, loss_iou=True,
                          train_stats_file=str(stats))
    noise = W.synth_noise(tuple(x.shape), 40, "train_noise")
    ref = g["living80.grad_norms"]

    def relerr(a, b):
        return np.abs(a - b) / np.maximum(b, 1e-3 * b.max())

    # driver 1: autograd over the HIP kernels
    losses, scal = diff.diffusion.p_losses(diff._denoise, x.to(dev()), t.to(dev()), noise=noise.to(dev()),
                                           condition=cond.to(dev()), condition_cross=None)
    losses.mean().backward()
    check(losses, g["living80.losses"], "N=80 losses (autograd)")
    for k, v in scal.items():
        assert abs(float(v.detach()) - float(g["living80." + k])) <= 1e-4 * max(1.0, abs(float(g["living80." + k]))), k
    params = dict(net.named_parameters())
    gn_auto = np.array([float(params[k].grad.norm()) for k in names])
    assert relerr(gn_auto, ref).max() < 1e-3
    check(net.init_conv.bias.grad, g["living80.grad.init_conv.bias"], "d init_conv.bias", tol=1e-3)
    # driver 2: the static plan (per-token context so that the same conditioning tensor is used)
    flat = FlatStorage(net)
    B, N, C = x.shape
    plan = TrainPlan(net, flat, diff.diffusion, B, N, SS_PER_TOKEN, 128, 0, 0, HipBackend(dev()))
    plan.x0.copy_(x.to(dev())); plan.noise.copy_(noise.to(dev())); plan.t.copy_(t.to(dev()))
    plan.ctx_in.t.copy_(cond.reshape(B * N, 128).to(dev()))
    plan.run_forward()
    plan.run_backward()
    check(plan.losses, g["living80.losses"], "N=80 losses (plan)")
    gn_plan = np.array([float(flat.grad_view(params[k]).norm()) for k in names])
    e = relerr(gn_plan, ref)
    print("plan grad-norm rel err vs reference fp32: max %.3g at %s" % (e.max(), names[int(e.argmax())]))
    assert e.max() < 1e-3
    assert relerr(gn_plan, gn_auto).max() < 2e-5
    check(flat.grad_view(net.init_conv.bias), g["living80.grad.init_conv.bias"], "d init_conv.bias (plan)", tol=1e-3)
    check(flat.grad_view(net.final_res_block.block2.proj.weight)[:8, :16, 0], g["living80.grad.final.block2.proj"],
          "d final_res_block.block2.proj slice (plan)", tol=1e-3)
