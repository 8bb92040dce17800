"""Golden outputs of the REAL reference at the HEADLINE batch: B=256 scenes of N=80 objects (build container only; TEST
INFRASTRUCTURE -- imported by tests/ only).

Usage:  python -m oracle.make_golden_b256   ->  tests/golden/b256.npz     (about 1 minute of CPU)

Every other reference golden is B=2 or 4; the headline numbers are quoted at B=256, where the kernels run different tile
configurations (the 160 x 256 split-bf16 tiles, grouped weight gradients cut 4..7 ways over the tokens).  Stored, all from the
reference's own modules (oracle/ref_loader.py), diffusion_ddpm.py:520-665 and :300-345:
  losses        p_losses (+IoU term) per scene, (256,)
  <9 scalars>   the logged loss terms
  grad_norms    gradient norm of 16 parameters spread over the network (names in grad_names_b256.json)
  p_sample.*    ONE reverse step (model call + posterior step) at B=256: 16 whole scenes, and f64 sums over the full tensor
Weights and inputs are re-derived from seeds by the tests (oracle/weights.py, b256_inputs below); only outputs are stored.
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

from . import weights as W
from .make_golden import GOLDEN, Replay, build_ref

B, N = 256, 80
GRAD_PARAMS = 16


def b256_inputs(seed=0):
    kw = W.UNCOND_LIVING
    x = W.synth_scene_batch(B, N, kw["class_dim"], kw["objfeat_dim"], seed + 60)
    t = torch.tensor([(91 + 377 * i) % 1000 for i in range(B)], dtype=torch.int64)
    cond = W.synth_condition(B, N, 128, seed + 60, shared=True).contiguous()
    noise = W.synth_noise((B, N, kw["channels"]), seed + 60, "b256_train_noise")
    step_noise = W.synth_noise((B, N, kw["channels"]), seed + 61, "b256_step_noise")
    return kw, x, t, cond, noise, step_noise


def grad_param_names(all_names):
    idx = np.linspace(0, len(all_names) - 1, GRAD_PARAMS).round().astype(int)
    return [all_names[i] for i in idx]


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    stats_file = os.path.join(tempfile.mkdtemp(), "dataset_stats.txt")
    with open(stats_file, "w") as f:
        json.dump(W.DATASET_STATS, f)
    kw, x, t, cond, noise, step_noise = b256_inputs()
    out = {}
    net, diff = build_ref(kw, time_num=1000, model_mean_type="v", loss_separate=True, loss_iou=True, train_stats_file=stats_file)
    losses, scal = diff.diffusion.p_losses(diff._denoise, x, t, noise=noise, condition=cond, condition_cross=None)
    losses.mean().backward()
    out["losses"] = losses.detach().numpy()
    for k, v in scal.items():
        out[k] = np.float32(v.item())
    params = dict(net.named_parameters())
    names = grad_param_names(list(params))
    out["grad_norms"] = np.array([float(params[k].grad.norm()) for k in names], dtype=np.float32)
    with open(os.path.join(GOLDEN, "grad_names_b256.json"), "w") as f:
        json.dump(names, f)
    print("b256 losses mean %.6f, grad norms %s" % (float(losses.mean()), out["grad_norms"][:4]))
    # one reverse step at B=256 (p_sample: model call + posterior mean + noise, :300-345), x_t = the q_sample of the batch
    net.zero_grad(set_to_none=True)
    with torch.no_grad():
        x_t = diff.diffusion.q_sample(x, t, noise=noise)
        y = diff.diffusion.p_sample(diff._denoise, x_t, t, cond, None, noise_fn=Replay([step_noise]), clip_denoised=True)
    out["p_sample.scenes16"] = y[::16].numpy().copy()
    out["p_sample.sum"] = np.float64(y.double().sum().item())
    out["p_sample.abs_sum"] = np.float64(y.double().abs().sum().item())
    print("p_sample sum %.6f abs-sum %.6f" % (out["p_sample.sum"], out["p_sample.abs_sum"]))
    np.savez_compressed(os.path.join(GOLDEN, "b256.npz"), **out)
    print("written", os.path.join(GOLDEN, "b256.npz"))


if __name__ == "__main__":
    sys.exit(main())
