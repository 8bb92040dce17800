"""Golden outputs of the REAL reference at the BASELINE.json shapes (build container only; TEST INFRASTRUCTURE).

Usage:  python -m oracle.make_golden_wide   ->  tests/golden/wide.npz (+ grad_names_living80.json)

oracle/make_golden.py pins the path at B=2, N=12/21; this file adds the shapes the headline numbers are quoted on:
  living80   Unet1D.forward, p_losses (+IoU) with the gradient norms of all 442 parameters, at N=80 (C=65)
  complete80 p_sample_loop_complete, N=80, P=20 given objects, T=50
  arrange80  p_sample_loop_arrange, N=80, T=50 (5 diffused channels, 512-d per-token condition)
  text32     Unet1D.forward and a T=20 chain with L=32 cross-attention tokens, B=4, N=12
  traj       p_sample_loop_trajectory (freq=10), T=50, B=2, N=12
Weights and inputs are re-derived from seeds by the tests (oracle/weights.py); only outputs are stored.
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

WIDE = {
    # name: (net_kwargs, B, N, ctx_dim, L)
    "living80": (W.UNCOND_LIVING, 2, 80, 128, 0),
    "text32": (W.TEXT_BEDROOM, 4, 12, 128, 32),
    "arrange80": (W.REARRANGE_LIVING, 2, 80, 512, 0),
}


def wide_inputs(name, seed=0):
    kw, B, N, ctx_dim, L = WIDE[name]
    C = kw["channels"]
    if C == 5:
        x = W.synth_noise((B, N, 5), seed + 40, "x5w") * 0.5
    else:
        x = W.synth_scene_batch(B, N, kw["class_dim"], kw["objfeat_dim"], seed + 40)
    t = torch.tensor([(91 + 377 * i) % 1000 for i in range(B)], dtype=torch.int64)
    cond = W.synth_condition(B, N, ctx_dim, seed + 40, shared=(ctx_dim == 128)).contiguous()
    cross = W.synth_text_condition(B, L, kw.get("text_dim", 512), seed + 40) if L else None
    return kw, x, t, cond, cross


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}
    stats_file = os.path.join(tempfile.mkdtemp(), "dataset_stats.txt")
    with open(stats_file, "w") as f:
        json.dump(W.DATASET_STATS, f)

    # ---- living80: forward + p_losses + gradients ------------------------------------------------------------
    kw, x, t, cond, _ = wide_inputs("living80")
    net, diff = build_ref(kw, time_num=1000, model_mean_type="v", loss_separate=True, loss_iou=True,
                          train_stats_file=stats_file)
    with torch.no_grad():
        out["living80.forward"] = net(x, t, cond, None).numpy()
    noise = W.synth_noise(tuple(x.shape), 40, "train_noise")
    losses, scal = diff.diffusion.p_losses(diff._denoise, x, t, noise=noise, condition=cond, condition_cross=None)
    losses.mean().backward()
    out["living80.losses"] = losses.detach().numpy()
    for k, v in scal.items():
        out["living80." + k] = np.float32(v.item())
    names = [k for k, _ in net.named_parameters()]
    out["living80.grad_norms"] = np.array([float(p.grad.norm()) for _, p in net.named_parameters()], dtype=np.float32)
    out["living80.grad.init_conv.bias"] = net.init_conv.bias.grad.numpy().copy()
    out["living80.grad.final.block2.proj"] = net.final_res_block.block2.proj.weight.grad.numpy()[:8, :16, 0].copy()
    with open(os.path.join(GOLDEN, "grad_names_living80.json"), "w") as f:
        json.dump(names, f)
    print("living80 losses", out["living80.losses"])

    # ---- complete80: N=80, P=20, T=50 ------------------------------------------------------------------------
    B, N, C = x.shape
    T, P = 50, 20
    net, diff = build_ref(kw, time_num=T, model_mean_type="v")
    shapes = [(B, N, C)]
    for _ in range(T):
        shapes += [(B, P, C), (B, N, C)]
    seq = noise_list(shapes, 41, "complete80_")
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        s = diff.complete_samples((B, N, C), "cpu", condition=cond, condition_cross=None, noise_fn=Replay(seq),
                                  clip_denoised=True, partial_boxes=x[:, :P, :].contiguous())
    out["complete80.T50"] = s.numpy()
    print("complete80", float(s.abs().mean()))

    # ---- arrange80: N=80, T=50 -------------------------------------------------------------------------------
    kwr, xr, _, condr, _ = wide_inputs("arrange80")
    Br, Nr = xr.shape[:2]
    full = W.synth_scene_batch(Br, Nr, 25, 32, 45)
    net, diff = build_ref(kwr, time_num=50, model_mean_type="v", config_extra={"room_arrange_condition": True})
    seq = noise_list([(Br, Nr, 5)] * 51, 42, "arrange80_")
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        s = diff.arrange_samples((Br, Nr, 65), "cpu", condition=condr, condition_cross=None, noise_fn=Replay(seq),
                                 clip_denoised=True, input_boxes=full)
    out["arrange80.T50"] = s.numpy()
    with torch.no_grad():
        out["arrange80.forward"] = net(xr, torch.tensor([91, 468]), condr, None).numpy()
    print("arrange80", float(s.abs().mean()))

    # ---- text32: forward + chain -----------------------------------------------------------------------------
    kwt, xt, tt, condt, crosst = wide_inputs("text32")
    net, diff = build_ref(kwt, time_num=20, model_mean_type="v")
    with torch.no_grad():
        out["text32.forward"] = net(xt, tt, condt, crosst).numpy()
    seq = noise_list([tuple(xt.shape)] * 21, 43, "text32_")
    with torch.no_grad():
        s = diff.gen_samples(tuple(xt.shape), "cpu", condition=condt, condition_cross=crosst, noise_fn=Replay(seq),
                             clip_denoised=True)
    out["text32.T20"] = s.numpy()
    print("text32", float(s.abs().mean()))

    # ---- trajectory (uncond bedroom B=2, N=12, T=50, freq=10) -------------------------------------------------
    kwb = W.UNCOND_BEDROOM
    Bb, Nb, Cb = 2, 12, 62
    condb = W.synth_condition(Bb, Nb, 128, 0).contiguous()
    net, diff = build_ref(kwb, time_num=50, model_mean_type="v")
    seq = noise_list([(Bb, Nb, Cb)] * 51, 44, "traj_")
    with torch.no_grad():
        imgs = diff.gen_sample_traj((Bb, Nb, Cb), "cpu", freq=10, condition=condb, condition_cross=None,
                                    noise_fn=Replay(seq), clip_denoised=True)
    out["traj.T50"] = np.stack([i.numpy() for i in imgs])
    print("traj", out["traj.T50"].shape)
    np.savez_compressed(os.path.join(GOLDEN, "wide.npz"), **out)
    print("written", os.path.join(GOLDEN, "wide.npz"))


if __name__ == "__main__":
    sys.exit(main())
