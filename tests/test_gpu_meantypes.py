"""GPU: the other two prediction types of GaussianDiffusion -- 'eps' (three of the twelve shipped configs,
config/uncond/*_eps.yaml) and 'x0' -- against outputs of the REAL reference (tests/golden/meantypes.npz, oracle/make_golden_meantypes.py).
Every other real-reference golden uses 'v'.  Per type: p_losses with the IoU term through the autograd path AND through the static
training plan (losses, logged terms, gradient norms of all parameters, three gradient slices), and a T = 50 reverse chain with
replayed noise, clipped and unclipped, eager and replayed from the hipGraph.  Tolerances as everywhere: 1e-4 on outputs and losses,
1e-3 on gradient norms (the reference's own fp32 CPU gradients are 2.6e-4 from an fp64 evaluation)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import weights as W  # noqa: E402
from oracle.make_golden import case_inputs, noise_list  # noqa: E402

from test_gpu_wide import check, dev  # noqa: E402

_PART_KEYS = ('loss.bbox', 'loss.trans', 'loss.size', 'loss.angle', 'loss.class', 'loss.object', 'loss.objfeat', 'loss.liou',
              'loss.bbox_iou')


def _model(tmp_path, mean_type, time_num):
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    kw = W.UNCOND_BEDROOM
    stats = tmp_path / "dataset_stats.txt"
    stats.write_text(json.dumps(W.DATASET_STATS))
    net = Unet1D(**kw)
    net.load_state_dict(W.synth_state_dict(kw))
    net.to(dev())
    cfg = dict(objectness_dim=0, class_dim=kw["class_dim"], angle_dim=2, objfeat_dim=32)
    return net, DiffusionPoint(net, cfg, time_num=time_num, model_mean_type=mean_type, loss_separate=True, loss_iou=True,
                               train_stats_file=str(stats))


def _grad_checks(g, mt, names, grad_of, what):
    gn = np.array([float(grad_of(k).norm()) for k in names])
    ref = g[mt + ".grad_norms"]
    e = np.abs(gn - ref) / np.maximum(ref, 1e-3 * ref.max())
    print("%s %s: grad-norm rel err vs the reference's fp32 CPU gradients: max %.3g at %s" % (mt, what, e.max(), names[int(e.argmax())]))
    assert e.max() < 1e-3, (mt, what, names[int(e.argmax())], e.max())
    check(grad_of("init_conv.bias"), g[mt + ".grad.init_conv.bias"], "%s %s d init_conv.bias" % (mt, what))
    check(grad_of("mid_attn.fn.fn.to_qkv.weight")[:8, :16, 0], g[mt + ".grad.mid_attn.to_qkv"], "%s %s d mid_attn.to_qkv" % (mt, what))
    check(grad_of("final_res_block.block2.proj.weight")[:8, :16, 0], g[mt + ".grad.final.block2.proj"], "%s %s d final.block2.proj" % (mt, what))


@pytest.mark.parametrize("mean_type", ["eps", "x0"])
def test_p_losses_and_gradients(golden_dir, tmp_path, mean_type):
    from diffuscene_amd._lib import SS_PER_SLOT
    from diffuscene_amd.flat import FlatStorage
    from diffuscene_amd.train_plan import HipBackend, TrainPlan
    g = np.load(os.path.join(golden_dir, "meantypes.npz"))
    kw, x, t, cond, _ = case_inputs("uncond_bedroom")
    noise = W.synth_noise(tuple(x.shape), 0, "train_noise")
    net, diff = _model(tmp_path, mean_type, 1000)
    names = [k for k, _ in net.named_parameters()]
    assert len(names) == len(g[mean_type + ".grad_norms"])
    # autograd path
    losses, scal = diff.diffusion.p_losses(diff._denoise, x.to(dev()), t.to(dev()), noise=noise.to(dev()), condition=cond.to(dev()),
                                           condition_cross=None)
    losses.mean().backward()
    check(losses, g[mean_type + ".losses"], "%s p_losses (autograd path)" % mean_type)
    for k, v in scal.items():
        want = float(g[mean_type + "." + k])
        assert abs(float(v.detach()) - want) <= 1e-4 * max(1.0, abs(want)), (mean_type, k, float(v.detach()), want)
    params = dict(net.named_parameters())
    _grad_checks(g, mean_type, names, lambda k: params[k].grad, "autograd path")
    # static training plan (what train_on_batch runs)
    for p in net.parameters():
        p.grad = None
    flat = FlatStorage(net)
    B, N, C = x.shape
    plan = TrainPlan(net, flat, diff.diffusion, B, N, SS_PER_SLOT, 128, 0, 0, HipBackend(dev()))
    plan.x0.copy_(x.to(dev())); plan.noise.copy_(noise.to(dev())); plan.t.copy_(t.to(dev()))
    plan.ctx_in.t.copy_(cond[0].to(dev()))
    flat.G.fill_(float("nan"))
    flat.zero_head()
    plan.run_forward()
    plan.run_backward()
    torch.cuda.synchronize()
    check(plan.losses, g[mean_type + ".losses"], "%s p_losses (training plan)" % mean_type)
    means = plan.parts.mean(dim=0).cpu()
    for i, k in enumerate(_PART_KEYS):
        want = float(g[mean_type + "." + k])
        assert abs(float(means[i]) - want) <= 1e-4 * max(1.0, abs(want)), (mean_type, k, float(means[i]), want)
    params = dict(net.named_parameters())
    _grad_checks(g, mean_type, names, lambda k: flat.grad_view(params[k]), "training plan")


@pytest.mark.parametrize("mean_type", ["eps", "x0"])
def test_reverse_chain(golden_dir, tmp_path, mean_type):
    from diffuscene_amd.sampler import NoiseReplay
    g = np.load(os.path.join(golden_dir, "meantypes.npz"))
    kw, x, t, cond, _ = case_inputs("uncond_bedroom")
    B, N, C = x.shape
    net, diff = _model(tmp_path, mean_type, 50)
    cd = cond.to(dev())
    for clip, seed, tag in ((True, 11, "clip"), (False, 12, "noclip")):
        seq = torch.stack(noise_list([(B, N, C)] * 51, seed, "mt_%s_" % tag)).to(dev())
        with torch.no_grad():
            eager = diff.gen_samples((B, N, C), dev(), condition=cd, noise_fn=NoiseReplay(seq), clip_denoised=clip, graph=False)
            graph = diff.gen_samples((B, N, C), dev(), condition=cd, noise_fn=NoiseReplay(seq), clip_denoised=clip, graph=True)
        assert torch.equal(eager, graph), "the captured step must reproduce the eager loop bit for bit"
        check(eager, g["%s.T50.%s" % (mean_type, tag)], "%s T=50 chain (%s)" % (mean_type, tag))


def test_fixedlarge_variance_chain(golden_dir, tmp_path):
    """model_var_type='fixedlarge' (sigma_t^2 = beta_t; diffusion_ddpm.py:314-321): the other variance branch, T = 50, eager and graph."""
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    from diffuscene_amd.sampler import NoiseReplay
    g = np.load(os.path.join(golden_dir, "meantypes.npz"))
    kw, x, t, cond, _ = case_inputs("uncond_bedroom")
    B, N, C = x.shape
    net = Unet1D(**kw)
    net.load_state_dict(W.synth_state_dict(kw))
    net.to(dev())
    diff = DiffusionPoint(net, dict(objectness_dim=0, class_dim=kw["class_dim"], angle_dim=2, objfeat_dim=32), time_num=50,
                          model_mean_type="v", model_var_type="fixedlarge")
    seq = torch.stack(noise_list([(B, N, C)] * 51, 13, "mt_large_")).to(dev())
    with torch.no_grad():
        eager = diff.gen_samples((B, N, C), dev(), condition=cond.to(dev()), noise_fn=NoiseReplay(seq), clip_denoised=True, graph=False)
        graph = diff.gen_samples((B, N, C), dev(), condition=cond.to(dev()), noise_fn=NoiseReplay(seq), clip_denoised=True, graph=True)
    assert torch.equal(eager, graph)
    check(eager, g["fixedlarge.T50.clip"], "fixedlarge T=50 chain")


def test_closed_form_helpers_and_model_predictions(golden_dir, tmp_path):
    """q_mean_variance, q_posterior_mean_variance, the four _predict_* conversions (reference diffusion_ddpm.py:217-241, 267-303) and
    model_predictions (:242-265) for every prediction type with its clip / rederive switches, on seeded tensors at t = 0 and t = 999."""
    g = np.load(os.path.join(golden_dir, "meantypes.npz"))
    kw, x, _, cond, _ = case_inputs("uncond_bedroom")
    ht = torch.tensor([0, 999], dtype=torch.int64, device=dev())
    hx = x.to(dev())
    h1, h2 = W.synth_noise(tuple(x.shape), 21, "helper_a").to(dev()), W.synth_noise(tuple(x.shape), 22, "helper_b").to(dev())
    cd = cond.to(dev())
    for mt in ("v", "eps", "x0"):
        net, diff = _model(tmp_path, mt, 1000)
        gd = diff.diffusion
        with torch.no_grad():
            if mt == "v":
                res = {"q_mean_variance": gd.q_mean_variance(hx, ht), "q_posterior_mean_variance": gd.q_posterior_mean_variance(hx, h1, ht),
                       "_predict_xstart_from_eps": (gd._predict_xstart_from_eps(h1, ht, h2),), "_predict_eps_from_start": (gd._predict_eps_from_start(h1, ht, hx),),
                       "_predict_v": (gd._predict_v(hx, ht, h2),), "_predict_start_from_v": (gd._predict_start_from_v(h1, ht, h2),)}
                for k, vs in res.items():
                    for i, v in enumerate(vs):
                        v = v * torch.ones_like(hx) if v.shape != hx.shape else v
                        check(v, g["helper.%s.%d" % (k, i)], "%s[%d]" % (k, i), tol=2e-6)
            for clip in (False, True):
                for rederive in (False, True):
                    mp = gd.model_predictions(diff._denoise, h1, ht, cd, None, clip_x_start=clip, rederive_pred_noise=rederive)
                    tag = "helper.model_predictions.%s.clip%d.rederive%d." % (mt, clip, rederive)
                    check(mp.pred_noise, g[tag + "pred_noise"], tag + "pred_noise")
                    check(mp.pred_x_start, g[tag + "pred_x_start"], tag + "pred_x_start")


def test_objectness_channel(golden_dir, tmp_path):
    """objectness_dim = 1: an extra channel with its own encoder / decoder MLP (reference denoise_net.py:513-516,580-583) and its own loss
    and IoU-mask branch (diffusion_ddpm.py:578-583,591-595,613-616).  No shipped YAML sets it; the reference implements it, so does the
    product: forward, p_losses through the autograd path and through the training plan."""
    from diffuscene_amd._lib import SS_PER_SLOT
    from diffuscene_amd.flat import FlatStorage
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    from diffuscene_amd.train_plan import HipBackend, TrainPlan
    g = np.load(os.path.join(golden_dir, "meantypes.npz"))
    kw, x, t, cond, _ = case_inputs("uncond_bedroom")
    nc = kw["class_dim"]
    kwo = dict(kw, objectness_dim=1, channels=kw["channels"] + 1)
    xo = torch.cat([x[:, :, :8 + nc], torch.where(x[:, :, 8 + nc - 1:8 + nc] > 0, -1.0, 1.0), x[:, :, 8 + nc:]], dim=-1).contiguous()
    stats = tmp_path / "dataset_stats.txt"
    stats.write_text(json.dumps(W.DATASET_STATS))
    net = Unet1D(**kwo)
    net.load_state_dict(W.synth_state_dict(kwo))
    net.to(dev())
    diff = DiffusionPoint(net, dict(objectness_dim=1, class_dim=nc, angle_dim=2, objfeat_dim=32), time_num=1000, model_mean_type="v",
                          loss_separate=True, loss_iou=True, train_stats_file=str(stats))
    with torch.no_grad():
        check(net(xo.to(dev()), t.to(dev()), cond.to(dev()), None), g["objectness.forward"], "objectness forward")
    noise = W.synth_noise(tuple(xo.shape), 0, "train_noise_obj")
    names = [k for k, _ in net.named_parameters()]
    ref = g["objectness.grad_norms"]
    assert len(names) == len(ref)

    def verify(what, losses, parts, grad_of):
        check(losses, g["objectness.losses"], "objectness p_losses (%s)" % what)
        for k, v in parts.items():
            want = float(g["objectness." + k])
            assert abs(float(v) - want) <= 1e-4 * max(1.0, abs(want)), (what, k, float(v), want)
        gn = np.array([float(grad_of(k).norm()) for k in names])
        e = np.abs(gn - ref) / np.maximum(ref, 1e-3 * ref.max())
        print("objectness %s: grad-norm rel err max %.3g at %s" % (what, e.max(), names[int(e.argmax())]))
        assert e.max() < 1e-3, (what, names[int(e.argmax())], e.max())
        check(grad_of("objectness_embedf.0.weight")[:, :, 0], g["objectness.grad.objectness_embedf.0"], "objectness %s d embedf.0" % what)
        check(grad_of("objectness_hidden2output.4.weight")[:, :64, 0], g["objectness.grad.objectness_hidden2output.4"], "objectness %s d hidden2output.4" % what)

    losses, scal = diff.diffusion.p_losses(diff._denoise, xo.to(dev()), t.to(dev()), noise=noise.to(dev()), condition=cond.to(dev()), condition_cross=None)
    losses.mean().backward()
    params = dict(net.named_parameters())
    verify("autograd path", losses, {k: v.detach() for k, v in scal.items()}, lambda k: params[k].grad)
    for p in net.parameters():
        p.grad = None
    flat = FlatStorage(net)
    B, N, C = xo.shape
    plan = TrainPlan(net, flat, diff.diffusion, B, N, SS_PER_SLOT, 128, 0, 0, HipBackend(dev()))
    plan.x0.copy_(xo.to(dev())); plan.noise.copy_(noise.to(dev())); plan.t.copy_(t.to(dev()))
    plan.ctx_in.t.copy_(cond[0].to(dev()))
    flat.G.fill_(float("nan"))
    flat.zero_head()
    plan.run_forward()
    plan.run_backward()
    torch.cuda.synchronize()
    means = plan.parts.mean(dim=0).cpu()
    params = dict(net.named_parameters())
    verify("training plan", plan.losses, {k: means[i] for i, k in enumerate(_PART_KEYS)}, lambda k: flat.grad_view(params[k]))


@pytest.mark.parametrize("tag", ["flat", "arrange_sep1", "arrange_sep0"])
def test_loss_layout_branches(golden_dir, tmp_path, tag):
    """loss_separate = False on the full layout (reference diffusion_ddpm.py:597-600) and both settings on the re-arrangement layout
    (:557-571: two logged terms, 5 diffused channels, per-token context): losses, logged terms and gradient norms of every parameter,
    autograd path and training plan."""
    from diffuscene_amd._lib import SS_PER_SLOT, SS_PER_TOKEN
    from diffuscene_amd.flat import FlatStorage
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    from diffuscene_amd.train_plan import HipBackend, TrainPlan
    g = np.load(os.path.join(golden_dir, "meantypes.npz"))
    arrange = tag.startswith("arrange")
    kw, x, t, cond, _ = case_inputs("rearrange_living" if arrange else "uncond_bedroom")
    net = Unet1D(**kw)
    net.load_state_dict(W.synth_state_dict(kw))
    net.to(dev())
    cfg = dict(objectness_dim=kw.get("objectness_dim", 1), class_dim=kw.get("class_dim", 21), angle_dim=kw.get("angle_dim", 1),
               objfeat_dim=kw.get("objfeat_dim", 0))
    if arrange:
        cfg["room_arrange_condition"] = True
    diff = DiffusionPoint(net, cfg, time_num=1000, model_mean_type="v", loss_separate=tag != "flat" and tag != "arrange_sep0", loss_iou=False)
    noise = W.synth_noise(tuple(x.shape), 0, "train_noise_arr" if arrange else "train_noise")
    names = [k for k, _ in net.named_parameters()]
    ref = g[tag + ".grad_norms"]
    assert len(names) == len(ref)
    logged = sorted(k[len(tag) + 1:] for k in g.files if k.startswith(tag + ".loss."))
    assert logged == (["loss.angle", "loss.trans"] if arrange else sorted(_PART_KEYS))

    def verify(what, losses, parts, grad_of):
        check(losses, g[tag + ".losses"], "%s p_losses (%s)" % (tag, what))
        for k in logged:
            want = float(g[tag + "." + k])
            assert abs(float(parts[k]) - want) <= 1e-4 * max(1.0, abs(want)), (tag, what, k, float(parts[k]), want)
        gn = np.array([float(grad_of(k).norm()) for k in names])
        e = np.abs(gn - ref) / np.maximum(ref, 1e-3 * ref.max())
        print("%s %s: grad-norm rel err max %.3g at %s" % (tag, what, e.max(), names[int(e.argmax())]))
        assert e.max() < 1e-3, (tag, what, names[int(e.argmax())], e.max())

    losses, scal = diff.diffusion.p_losses(diff._denoise, x.to(dev()), t.to(dev()), noise=noise.to(dev()), condition=cond.to(dev()), condition_cross=None)
    losses.mean().backward()
    params = dict(net.named_parameters())
    verify("autograd path", losses, {k: v.detach() for k, v in scal.items()}, lambda k: params[k].grad)
    for p in net.parameters():
        p.grad = None
    flat = FlatStorage(net)
    B, N, C = x.shape
    plan = TrainPlan(net, flat, diff.diffusion, B, N, SS_PER_TOKEN if arrange else SS_PER_SLOT, 512 if arrange else 128, 0, 0, HipBackend(dev()))
    plan.x0.copy_(x.to(dev())); plan.noise.copy_(noise.to(dev())); plan.t.copy_(t.to(dev()))
    plan.ctx_in.t.copy_(cond.reshape(B * N, 512).to(dev()) if arrange else cond[0].to(dev()))
    flat.G.fill_(float("nan"))
    flat.zero_head()
    plan.run_forward()
    plan.run_backward()
    torch.cuda.synchronize()
    means = plan.parts.mean(dim=0).cpu()
    # columns of plan.parts: the nine logged terms in _PART_KEYS order; the re-arrangement layout fills trans (1) and angle (3) (train_step.loss_step)
    parts = {"loss.trans": means[1], "loss.angle": means[3]} if arrange else {k: means[i] for i, k in enumerate(_PART_KEYS)}
    params = dict(net.named_parameters())
    verify("training plan", plan.losses, parts, lambda k: flat.grad_view(params[k]))


def test_objfeat_dim_64(golden_dir, tmp_path):
    """objfeat_dim = 64 (94 channels; the wider latent code the datasets also carry, read from sample_params["objfeats"]): forward,
    p_losses through both training paths, a T = 20 chain eager and from the hipGraph.  The decoder's stacked output projection pads
    its heads to 64 rows for this network (engine.dec_pad) instead of 32."""
    from diffuscene_amd._lib import SS_PER_SLOT
    from diffuscene_amd.flat import FlatStorage
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    from diffuscene_amd.sampler import NoiseReplay
    from diffuscene_amd.train_plan import HipBackend, TrainPlan
    g = np.load(os.path.join(golden_dir, "meantypes.npz"))
    kw, x, t, cond, _ = case_inputs("uncond_bedroom")
    nc = kw["class_dim"]
    kw64 = dict(kw, objfeat_dim=64, channels=8 + nc + 64)
    B, N = x.shape[:2]
    x64 = W.synth_scene_batch(B, N, nc, 64, seed=0)
    stats = tmp_path / "dataset_stats.txt"
    stats.write_text(json.dumps(W.DATASET_STATS))
    net = Unet1D(**kw64)
    net.load_state_dict(W.synth_state_dict(kw64))
    net.to(dev())
    cfg = dict(objectness_dim=0, class_dim=nc, angle_dim=2, objfeat_dim=64)
    diff = DiffusionPoint(net, cfg, time_num=1000, model_mean_type="v", loss_separate=True, loss_iou=True, train_stats_file=str(stats))
    with torch.no_grad():
        check(net(x64.to(dev()), t.to(dev()), cond.to(dev()), None), g["objfeat64.forward"], "objfeat64 forward")
    assert net.engine(dev()).dec_pad == 64
    noise = W.synth_noise(tuple(x64.shape), 0, "train_noise_64")
    names = [k for k, _ in net.named_parameters()]
    ref = g["objfeat64.grad_norms"]

    def verify(what, losses, parts, grad_of):
        check(losses, g["objfeat64.losses"], "objfeat64 p_losses (%s)" % what)
        for k in _PART_KEYS:
            want = float(g["objfeat64." + k])
            assert abs(float(parts[k]) - want) <= 1e-4 * max(1.0, abs(want)), (what, k, float(parts[k]), want)
        gn = np.array([float(grad_of(k).norm()) for k in names])
        e = np.abs(gn - ref) / np.maximum(ref, 1e-3 * ref.max())
        print("objfeat64 %s: grad-norm rel err max %.3g at %s" % (what, e.max(), names[int(e.argmax())]))
        assert e.max() < 1e-3, (what, names[int(e.argmax())], e.max())

    losses, scal = diff.diffusion.p_losses(diff._denoise, x64.to(dev()), t.to(dev()), noise=noise.to(dev()), condition=cond.to(dev()), condition_cross=None)
    losses.mean().backward()
    params = dict(net.named_parameters())
    verify("autograd path", losses, {k: v.detach() for k, v in scal.items()}, lambda k: params[k].grad)
    for p in net.parameters():
        p.grad = None
    flat = FlatStorage(net)
    plan = TrainPlan(net, flat, diff.diffusion, B, N, SS_PER_SLOT, 128, 0, 0, HipBackend(dev()))
    plan.x0.copy_(x64.to(dev())); plan.noise.copy_(noise.to(dev())); plan.t.copy_(t.to(dev()))
    plan.ctx_in.t.copy_(cond[0].to(dev()))
    flat.G.fill_(float("nan"))
    flat.zero_head()
    plan.run_forward()
    plan.run_backward()
    torch.cuda.synchronize()
    means = plan.parts.mean(dim=0).cpu()
    params = dict(net.named_parameters())
    verify("training plan", plan.losses, {k: means[i] for i, k in enumerate(_PART_KEYS)}, lambda k: flat.grad_view(params[k]))
    diff20 = DiffusionPoint(net, cfg, time_num=20, model_mean_type="v")
    seq = torch.stack(noise_list([(B, N, 94)] * 21, 14, "mt_64_")).to(dev())
    with torch.no_grad():
        eager = diff20.gen_samples((B, N, 94), dev(), condition=cond.to(dev()), noise_fn=NoiseReplay(seq), clip_denoised=True, graph=False)
        graph = diff20.gen_samples((B, N, 94), dev(), condition=cond.to(dev()), noise_fn=NoiseReplay(seq), clip_denoised=True, graph=True)
    assert torch.equal(eager, graph)
    check(eager, g["objfeat64.T20"], "objfeat64 T=20 chain")


def test_reference_default_layout(golden_dir, tmp_path):
    """The constructor defaults of the reference (objectness_dim 1, class_dim 21, angle_dim 1 -- a raw angle, bbox_dim 7 --, no shape code:
    29 channels): forward, p_losses with the IoU term through both training paths, a T = 20 chain eager and from the hipGraph."""
    from diffuscene_amd._lib import SS_PER_SLOT
    from diffuscene_amd.flat import FlatStorage
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    from diffuscene_amd.sampler import NoiseReplay
    from diffuscene_amd.train_plan import HipBackend, TrainPlan
    g = np.load(os.path.join(golden_dir, "meantypes.npz"))
    kw, x, t, cond, _ = case_inputs("uncond_bedroom")
    B, N = x.shape[:2]
    kwl = dict(kw, objectness_dim=1, class_dim=21, angle_dim=1, objfeat_dim=0, channels=29)
    base = W.synth_scene_batch(B, N, 21, 0, seed=0)
    xl = torch.cat([base[:, :, :6], torch.atan2(base[:, :, 7:8], base[:, :, 6:7]) / np.pi, base[:, :, 8:29],
                    torch.where(base[:, :, 28:29] > 0, -1.0, 1.0)], dim=-1).contiguous()
    stats = tmp_path / "dataset_stats.txt"
    stats.write_text(json.dumps(W.DATASET_STATS))
    net = Unet1D(**kwl)
    net.load_state_dict(W.synth_state_dict(kwl))
    net.to(dev())
    cfg = dict(objectness_dim=1, class_dim=21, angle_dim=1, objfeat_dim=0)
    diff = DiffusionPoint(net, cfg, time_num=1000, model_mean_type="v", loss_separate=True, loss_iou=True, train_stats_file=str(stats))
    with torch.no_grad():
        check(net(xl.to(dev()), t.to(dev()), cond.to(dev()), None), g["legacy.forward"], "default-layout forward")
    noise = W.synth_noise(tuple(xl.shape), 0, "train_noise_legacy")
    names = [k for k, _ in net.named_parameters()]
    ref = g["legacy.grad_norms"]
    assert len(names) == len(ref)

    def verify(what, losses, parts, grad_of):
        check(losses, g["legacy.losses"], "default-layout p_losses (%s)" % what)
        for k in _PART_KEYS:
            want = float(g["legacy." + k])
            assert abs(float(parts[k]) - want) <= 1e-4 * max(1.0, abs(want)), (what, k, float(parts[k]), want)
        gn = np.array([float(grad_of(k).norm()) for k in names])
        e = np.abs(gn - ref) / np.maximum(ref, 1e-3 * ref.max())
        print("default layout %s: grad-norm rel err max %.3g at %s" % (what, e.max(), names[int(e.argmax())]))
        assert e.max() < 1e-3, (what, names[int(e.argmax())], e.max())

    losses, scal = diff.diffusion.p_losses(diff._denoise, xl.to(dev()), t.to(dev()), noise=noise.to(dev()), condition=cond.to(dev()), condition_cross=None)
    losses.mean().backward()
    params = dict(net.named_parameters())
    verify("autograd path", losses, {k: v.detach() for k, v in scal.items()}, lambda k: params[k].grad)
    for p in net.parameters():
        p.grad = None
    flat = FlatStorage(net)
    plan = TrainPlan(net, flat, diff.diffusion, B, N, SS_PER_SLOT, 128, 0, 0, HipBackend(dev()))
    plan.x0.copy_(xl.to(dev())); plan.noise.copy_(noise.to(dev())); plan.t.copy_(t.to(dev()))
    plan.ctx_in.t.copy_(cond[0].to(dev()))
    flat.G.fill_(float("nan"))
    flat.zero_head()
    plan.run_forward()
    plan.run_backward()
    torch.cuda.synchronize()
    means = plan.parts.mean(dim=0).cpu()
    params = dict(net.named_parameters())
    verify("training plan", plan.losses, {k: means[i] for i, k in enumerate(_PART_KEYS)}, lambda k: flat.grad_view(params[k]))
    diff20 = DiffusionPoint(net, cfg, time_num=20, model_mean_type="v")
    seq = torch.stack(noise_list([(B, N, 29)] * 21, 15, "mt_legacy_")).to(dev())
    with torch.no_grad():
        eager = diff20.gen_samples((B, N, 29), dev(), condition=cond.to(dev()), noise_fn=NoiseReplay(seq), clip_denoised=True, graph=False)
        graph = diff20.gen_samples((B, N, 29), dev(), condition=cond.to(dev()), noise_fn=NoiseReplay(seq), clip_denoised=True, graph=True)
    assert torch.equal(eager, graph)
    check(eager, g["legacy.T20"], "default-layout T=20 chain")
