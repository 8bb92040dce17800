"""tests/golden/meantypes.npz: the REAL reference run with the other two prediction types of GaussianDiffusion (build container only).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Usage:  python -m oracle.make_golden_meantypes     (~2 minutes)

Every other golden of this repo uses model_mean_type='v' (the `_v` YAMLs).  Three of the twelve shipped configs
(config/uncond/diffusion_*_instancond_lat32_eps.yaml) train and sample with 'eps'; 'x0' is the third branch of
diffusion_ddpm.py:248-262 (model_predictions) / :536-545 (the p_losses target).  Per type, on the bedroom network of
oracle/make_golden.CASES with seeded weights and inputs:
  * p_losses with loss_separate + the IoU term (B = 2): per-scene losses, the logged terms, gradient norms of every parameter
    and three gradient slices;
  * a T = 50 reverse chain with replayed noise, clipped and unclipped (the clip acts on x0, which each type derives differently).
Plus one T = 50 chain with model_var_type='fixedlarge' (the other variance branch of p_mean_variance, :314-321).
"""
import contextlib
import io
import json
import os
import sys
import tempfile

import numpy as np
import torch

from . import weights as W
from .make_golden import GOLDEN, Replay, build_ref, case_inputs, noise_list


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    stats_file = os.path.join(tempfile.mkdtemp(), "dataset_stats.txt")
    with open(stats_file, "w") as f:
        json.dump(W.DATASET_STATS, f)
    out = {}
    kw, x, t, cond, _ = case_inputs("uncond_bedroom")
    B, N, C = x.shape
    T = 50
    for mt in ("eps", "x0"):
        net, diff = build_ref(kw, time_num=1000, model_mean_type=mt, loss_separate=True, loss_iou=True, train_stats_file=stats_file)
        noise = W.synth_noise(tuple(x.shape), 0, "train_noise")
        with contextlib.redirect_stdout(io.StringIO()):
            losses, scal = diff.diffusion.p_losses(diff._denoise, x, t, noise=noise, condition=cond, condition_cross=None)
        losses.mean().backward()
        out[mt + ".losses"] = losses.detach().numpy()
        for k, v in scal.items():
            out[mt + "." + k] = np.float32(v.item())
        out[mt + ".grad_norms"] = np.array([float(p.grad.norm()) for _, p in net.named_parameters()], dtype=np.float32)
        out[mt + ".grad.init_conv.bias"] = net.init_conv.bias.grad.numpy().copy()
        out[mt + ".grad.mid_attn.to_qkv"] = net.mid_attn.fn.fn.to_qkv.weight.grad.numpy()[:8, :16, 0].copy()
        out[mt + ".grad.final.block2.proj"] = net.final_res_block.block2.proj.weight.grad.numpy()[:8, :16, 0].copy()
        print(mt, "p_losses", losses.detach().numpy(), {k: round(float(v), 5) for k, v in scal.items()})
        net, diff = build_ref(kw, time_num=T, model_mean_type=mt)
        for clip, seed, tag in ((True, 11, "clip"), (False, 12, "noclip")):
            seq = noise_list([(B, N, C)] * (T + 1), seed, "mt_%s_" % tag)
            with torch.no_grad():
                s = diff.gen_samples((B, N, C), "cpu", condition=cond, condition_cross=None, noise_fn=Replay(seq), clip_denoised=clip)
            out["%s.T50.%s" % (mt, tag)] = s.numpy()
            print(mt, "chain T=50", tag, float(s.abs().mean()), float(s.abs().max()))
    # model_var_type='fixedlarge' (diffusion_ddpm.py:314-321: sigma_t^2 = beta_t, the log-variance of step 0 taken from the posterior): the
    # other variance branch of p_mean_variance; no shipped config selects it, the product implements it (tb["_sigma_large"])
    net, diff = build_ref(kw, time_num=T, model_mean_type="v", model_var_type="fixedlarge")
    seq = noise_list([(B, N, C)] * (T + 1), 13, "mt_large_")
    with torch.no_grad():
        s = diff.gen_samples((B, N, C), "cpu", condition=cond, condition_cross=None, noise_fn=Replay(seq), clip_denoised=True)
    out["fixedlarge.T50.clip"] = s.numpy()
    print("fixedlarge chain T=50", float(s.abs().mean()), float(s.abs().max()))
    # the warm-up beta schedules of get_betas (diffusion_ddpm.py:62-79; 'cosine' never assigns betas there) with the tables
    # GaussianDiffusion derives from them, for 'eps' (the only type that adds tables of its own to the common set)
    for sched in ("warm0.1", "warm0.2", "warm0.5"):
        _, diff = build_ref(kw, time_num=1000, model_mean_type="eps", schedule_type=sched)
        gd = diff.diffusion
        for k in ("betas", "alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
                  "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2", "loss_weight"):
            out["%s.%s" % (sched, k)] = getattr(gd, k).numpy()
    # the small closed-form helpers every branch above is built from (diffusion_ddpm.py:217-241, 267-303) and model_predictions (:242-265)
    # with its clip / rederive switches, on their own: seeded tensors, T = 1000
    hx, ht = case_inputs("uncond_bedroom")[1], torch.tensor([0, 999], dtype=torch.int64)
    h1, h2 = W.synth_noise(tuple(hx.shape), 21, "helper_a"), W.synth_noise(tuple(hx.shape), 22, "helper_b")
    for mt in ("v", "eps", "x0"):
        net, diff = build_ref(kw, time_num=1000, model_mean_type=mt)
        gd = diff.diffusion
        with torch.no_grad():
            if mt == "v":
                res = {"q_mean_variance": gd.q_mean_variance(hx, ht), "q_posterior_mean_variance": gd.q_posterior_mean_variance(hx, h1, ht),
                       "_predict_xstart_from_eps": (gd._predict_xstart_from_eps(h1, ht, h2),), "_predict_eps_from_start": (gd._predict_eps_from_start(h1, ht, hx),),
                       "_predict_v": (gd._predict_v(hx, ht, h2),), "_predict_start_from_v": (gd._predict_start_from_v(h1, ht, h2),)}
                for k, vs in res.items():
                    for i, v in enumerate(vs):
                        out["helper.%s.%d" % (k, i)] = (v * torch.ones_like(hx) if v.shape != hx.shape else v).numpy()
            for clip in (False, True):
                for rederive in (False, True):
                    mp = gd.model_predictions(diff._denoise, h1, ht, cond, None, clip_x_start=clip, rederive_pred_noise=rederive)
                    out["helper.model_predictions.%s.clip%d.rederive%d.pred_noise" % (mt, clip, rederive)] = mp.pred_noise.numpy()
                    out["helper.model_predictions.%s.clip%d.rederive%d.pred_x_start" % (mt, clip, rederive)] = mp.pred_x_start.numpy()
    # objectness_dim = 1 (no shipped YAML sets it; the reference's older encodings do: an extra channel between the class scores and the
    # latent code, its own encoder / decoder MLP in Unet1D -- denoise_net.py:513-516,580-583 -- and its own loss / IoU-mask branch in
    # p_losses -- diffusion_ddpm.py:578-583,591-595,613-616): forward, p_losses with the IoU term, three gradient slices
    kwo = dict(kw, objectness_dim=1, channels=kw["channels"] + 1)
    xo = torch.cat([x[:, :, :8 + kw["class_dim"]], torch.where(x[:, :, 8 + kw["class_dim"] - 1:8 + kw["class_dim"]] > 0, -1.0, 1.0),
                    x[:, :, 8 + kw["class_dim"]:]], dim=-1).contiguous()           # objectness = +1 on real objects, -1 on empty slots
    net, diff = build_ref(kwo, time_num=1000, model_mean_type="v", loss_separate=True, loss_iou=True, train_stats_file=stats_file)
    with torch.no_grad():
        out["objectness.forward"] = net(xo, t, cond, None).numpy()
    noise = W.synth_noise(tuple(xo.shape), 0, "train_noise_obj")
    with contextlib.redirect_stdout(io.StringIO()):
        losses, scal = diff.diffusion.p_losses(diff._denoise, xo, t, noise=noise, condition=cond, condition_cross=None)
    losses.mean().backward()
    out["objectness.losses"] = losses.detach().numpy()
    for k, v in scal.items():
        out["objectness." + k] = np.float32(v.item())
    out["objectness.grad_norms"] = np.array([float(p.grad.norm()) for _, p in net.named_parameters()], dtype=np.float32)
    out["objectness.grad.objectness_embedf.0"] = net.objectness_embedf[0].weight.grad.numpy()[:, :, 0].copy()
    out["objectness.grad.objectness_hidden2output.4"] = net.objectness_hidden2output[4].weight.grad.numpy()[:, :64, 0].copy()
    print("objectness p_losses", losses.detach().numpy(), {k: round(float(v), 5) for k, v in scal.items()})
    # loss_separate = False (every shipped YAML sets True): the plain mean over all channels (diffusion_ddpm.py:563-566 for the
    # re-arrangement layout, :597-600 for the full layout), logged terms unchanged; 'v', no IoU term
    net, diff = build_ref(kw, time_num=1000, model_mean_type="v", loss_separate=False, loss_iou=False)
    noise = W.synth_noise(tuple(x.shape), 0, "train_noise")
    with contextlib.redirect_stdout(io.StringIO()):
        losses, scal = diff.diffusion.p_losses(diff._denoise, x, t, noise=noise, condition=cond, condition_cross=None)
    losses.mean().backward()
    out["flat.losses"] = losses.detach().numpy()
    for k, v in scal.items():
        out["flat." + k] = np.float32(v.item())
    out["flat.grad_norms"] = np.array([float(p.grad.norm()) for _, p in net.named_parameters()], dtype=np.float32)
    print("loss_separate=False p_losses", losses.detach().numpy())
    kwr, xr, tr_, condr, _ = case_inputs("rearrange_living")
    for sep in (True, False):
        net, diff = build_ref(kwr, time_num=1000, model_mean_type="v", loss_separate=sep, loss_iou=False,
                              config_extra={"room_arrange_condition": True})
        noise = W.synth_noise(tuple(xr.shape), 0, "train_noise_arr")
        with contextlib.redirect_stdout(io.StringIO()):
            losses, scal = diff.diffusion.p_losses(diff._denoise, xr, tr_, noise=noise, condition=condr, condition_cross=None)
        losses.mean().backward()
        tag = "arrange_sep%d" % sep
        out[tag + ".losses"] = losses.detach().numpy()
        for k, v in scal.items():
            out[tag + "." + k] = np.float32(v.item())
        out[tag + ".grad_norms"] = np.array([float(p.grad.norm()) for _, p in net.named_parameters()], dtype=np.float32)
        print(tag, "p_losses", losses.detach().numpy())
    # objfeat_dim = 64 (the datasets carry a 32-d and a 64-d shape code, threed_front_dataset.py:481-513; every shipped YAML diffuses the 32-d one,
    # the wrapper reads sample_params["objfeats"] for any other width, diffusion_scene_layout_ddpm.py:139-143): 94 channels
    kw64 = dict(kw, objfeat_dim=64, channels=8 + kw["class_dim"] + 64)
    x64 = W.synth_scene_batch(B, N, kw["class_dim"], 64, seed=0)
    net, diff = build_ref(kw64, time_num=1000, model_mean_type="v", loss_separate=True, loss_iou=True, train_stats_file=stats_file)
    with torch.no_grad():
        out["objfeat64.forward"] = net(x64, t, cond, None).numpy()
    noise = W.synth_noise(tuple(x64.shape), 0, "train_noise_64")
    with contextlib.redirect_stdout(io.StringIO()):
        losses, scal = diff.diffusion.p_losses(diff._denoise, x64, t, noise=noise, condition=cond, condition_cross=None)
    losses.mean().backward()
    out["objfeat64.losses"] = losses.detach().numpy()
    for k, v in scal.items():
        out["objfeat64." + k] = np.float32(v.item())
    out["objfeat64.grad_norms"] = np.array([float(p.grad.norm()) for _, p in net.named_parameters()], dtype=np.float32)
    net, diff = build_ref(kw64, time_num=20, model_mean_type="v")
    seq = noise_list([(B, N, 94)] * 21, 14, "mt_64_")
    with torch.no_grad():
        s = diff.gen_samples((B, N, 94), "cpu", condition=cond, condition_cross=None, noise_fn=Replay(seq), clip_denoised=True)
    out["objfeat64.T20"] = s.numpy()
    print("objfeat64 p_losses", losses.detach().numpy(), "chain", float(s.abs().mean()))
    # the constructor DEFAULTS of the reference (Unet1D / GaussianDiffusion: objectness_dim 1, class_dim 21, angle_dim 1, objfeat_dim 0 --
    # the layout of the encodings without a shape code and with a raw angle, bbox_dim 7): 29 channels
    kwl = dict(kw, objectness_dim=1, class_dim=21, angle_dim=1, objfeat_dim=0, channels=7 + 21 + 1)
    base = W.synth_scene_batch(B, N, 21, 0, seed=0)                       # [trans 3 | size 3 | cos | sin | class 21]
    xl = torch.cat([base[:, :, :6], torch.atan2(base[:, :, 7:8], base[:, :, 6:7]) / np.pi, base[:, :, 8:29],
                    torch.where(base[:, :, 28:29] > 0, -1.0, 1.0)], dim=-1).contiguous()
    net, diff = build_ref(kwl, time_num=1000, model_mean_type="v", loss_separate=True, loss_iou=True, train_stats_file=stats_file)
    with torch.no_grad():
        out["legacy.forward"] = net(xl, t, cond, None).numpy()
    noise = W.synth_noise(tuple(xl.shape), 0, "train_noise_legacy")
    with contextlib.redirect_stdout(io.StringIO()):
        losses, scal = diff.diffusion.p_losses(diff._denoise, xl, t, noise=noise, condition=cond, condition_cross=None)
    losses.mean().backward()
    out["legacy.losses"] = losses.detach().numpy()
    for k, v in scal.items():
        out["legacy." + k] = np.float32(v.item())
    out["legacy.grad_norms"] = np.array([float(p.grad.norm()) for _, p in net.named_parameters()], dtype=np.float32)
    net, diff = build_ref(kwl, time_num=20, model_mean_type="v")
    seq = noise_list([(B, N, 29)] * 21, 15, "mt_legacy_")
    with torch.no_grad():
        s = diff.gen_samples((B, N, 29), "cpu", condition=cond, condition_cross=None, noise_fn=Replay(seq), clip_denoised=True)
    out["legacy.T20"] = s.numpy()
    print("legacy p_losses", losses.detach().numpy(), {k: round(float(v), 5) for k, v in scal.items()}, "chain", float(s.abs().mean()))
    np.savez_compressed(os.path.join(GOLDEN, "meantypes.npz"), **out)
    print("written", os.path.join(GOLDEN, "meantypes.npz"))


if __name__ == "__main__":
    sys.exit(main())
