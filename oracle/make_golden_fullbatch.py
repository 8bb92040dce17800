"""Golden outputs of the REAL reference for the OTHER BASELINE.json configurations at their full per-GPU batch (build container
only; TEST INFRASTRUCTURE -- imported by tests/ only).

Usage:  python -m oracle.make_golden_fullbatch   ->  tests/golden/fullbatch.npz     (about 2 minutes of CPU)

tests/golden/b256.npz pins the metric configuration (configs[2]: B=256, N=80).  The remaining configurations are benchmarked at
their own batch (bench.py --config ...), where the launches take other tiles than at B=2..4 (bedroom21: 4 scenes x 128 channels per
block; text: every launch below the split-bf16 block threshold, the exact-f32 kernel; B=128, N=80: the 4-wave 160 x 128 form):
  bedroom21  configs[1]  uncond bedrooms, B=256, N=21, C=62      p_losses (+IoU), 9 scalars, 16 gradient norms, one reverse step
  text       configs[3]  text bedrooms, B=128, N=12, L=32        the same with 32 cross-attention tokens per scene (+ d cross norm)
  arrange    configs[4]  re-arrangement, B=128, N=80, 5 channels p_losses of the arrange branch (diffusion_ddpm.py:558-570), gradient
                                                                 norms, one reverse step on the 5 diffused channels
  complete   configs[4]  completion, B=128, N=80, P=20 given     p_sample_loop_complete (:447-476) with T=10
All from the reference's own modules (oracle/ref_loader.py).  Weights and inputs are re-derived from seeds by the tests
(oracle/weights.py, fullbatch_inputs below); only outputs are stored: per-scene losses, scalars, gradient norms, every 16th scene of
the sampled tensors and f64 sums over the whole of them.
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
from .make_golden import GOLDEN, Replay, build_ref, noise_list
from .make_golden_b256 import grad_param_names

FULL = {
    # name: (net_kwargs, B, N, ctx_dim, L)
    "bedroom21": (W.UNCOND_BEDROOM, 256, 21, 128, 0),
    "text": (W.TEXT_BEDROOM, 128, 12, 128, 32),
    "arrange": (W.REARRANGE_LIVING, 128, 80, 512, 0),
    "complete": (W.UNCOND_LIVING, 128, 80, 128, 0),
}
COMPLETE_T, COMPLETE_P = 10, 20


def fullbatch_inputs(name, seed=0):
    kw, B, N, ctx_dim, L = FULL[name]
    C = kw["channels"]
    s = seed + 70 + sorted(FULL).index(name)
    if C == 5:
        x = W.synth_noise((B, N, 5), s, "x5f") * 0.5
    else:
        x = W.synth_scene_batch(B, N, kw["class_dim"], kw["objfeat_dim"], s)
    t = torch.tensor([(53 + 389 * i) % 1000 for i in range(B)], dtype=torch.int64)
    cond = W.synth_condition(B, N, ctx_dim, s, shared=(ctx_dim == 128)).contiguous()
    cross = W.synth_text_condition(B, L, kw.get("text_dim", 512), s) if L else None
    noise = W.synth_noise((B, N, C), s, "full_train_noise")
    step_noise = W.synth_noise((B, N, C), s + 100, "full_step_noise")
    return kw, x, t, cond, cross, noise, step_noise


def complete_noise(B, N, C, seed=0):
    """noise_fn call order of p_sample_loop_complete: x_T, then per step the partial-scene q_sample draw and the p_sample draw."""
    shapes = [(B, N, C)]
    for _ in range(COMPLETE_T):
        shapes += [(B, COMPLETE_P, C), (B, N, C)]
    return noise_list(shapes, seed + 75, "complete_full_")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    stats_file = os.path.join(tempfile.mkdtemp(), "dataset_stats.txt")
    with open(stats_file, "w") as f:
        json.dump(W.DATASET_STATS, f)
    out, names_out = {}, {}
    for name in ("bedroom21", "text", "arrange"):
        kw, x, t, cond, cross, noise, step_noise = fullbatch_inputs(name)
        arrange = name == "arrange"
        extra = {"room_arrange_condition": True} if arrange else {}
        net, diff = build_ref(kw, time_num=1000, model_mean_type="v", loss_separate=True, loss_iou=not arrange,
                              train_stats_file=stats_file, config_extra=extra)
        crossg = cross.clone().requires_grad_(True) if cross is not None else None
        losses, scal = diff.diffusion.p_losses(diff._denoise, x, t, noise=noise, condition=cond, condition_cross=crossg)
        losses.mean().backward()
        out[name + ".losses"] = losses.detach().numpy()
        for k, v in scal.items():
            out[name + "." + k] = np.float32(v.item())
        params = dict(net.named_parameters())
        names = grad_param_names([k for k, p in params.items() if p.grad is not None])
        names_out[name] = names
        out[name + ".grad_norms"] = np.array([float(params[k].grad.norm()) for k in names], dtype=np.float32)
        if crossg is not None:
            out[name + ".d_cross_norm"] = np.float32(float(crossg.grad.norm()))
        print("%s: losses mean %.6f, grad norms %s" % (name, float(losses.mean()), out[name + ".grad_norms"][:4]))
        net.zero_grad(set_to_none=True)
        with torch.no_grad():
            x_t = diff.diffusion.q_sample(x, t, noise=noise)
            y = diff.diffusion.p_sample(diff._denoise, x_t, t, cond, cross, noise_fn=Replay([step_noise]), clip_denoised=True)
        out[name + ".p_sample.scenes16"] = y[::16].numpy().copy()
        out[name + ".p_sample.sum"] = np.float64(y.double().sum().item())
        out[name + ".p_sample.abs_sum"] = np.float64(y.double().abs().sum().item())
        print("%s: p_sample sum %.6f abs-sum %.6f" % (name, out[name + ".p_sample.sum"], out[name + ".p_sample.abs_sum"]))

    # ---- completion at B=128, N=80, P=20, T=10 ------------------------------------------------------------------
    kw, x, t, cond, _, _, _ = fullbatch_inputs("complete")
    B, N, C = x.shape
    net, diff = build_ref(kw, time_num=COMPLETE_T, model_mean_type="v")
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        s = diff.complete_samples((B, N, C), "cpu", condition=cond, condition_cross=None, noise_fn=Replay(complete_noise(B, N, C)),
                                  clip_denoised=True, partial_boxes=x[:, :COMPLETE_P, :].contiguous())
    out["complete.scenes16"] = s[::16].numpy().copy()
    out["complete.sum"] = np.float64(s.double().sum().item())
    out["complete.abs_sum"] = np.float64(s.double().abs().sum().item())
    print("complete: sum %.6f abs-sum %.6f" % (out["complete.sum"], out["complete.abs_sum"]))
    with open(os.path.join(GOLDEN, "grad_names_fullbatch.json"), "w") as f:
        json.dump(names_out, f)
    np.savez_compressed(os.path.join(GOLDEN, "fullbatch.npz"), **out)
    print("written", os.path.join(GOLDEN, "fullbatch.npz"))


if __name__ == "__main__":
    sys.exit(main())
