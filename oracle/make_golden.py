"""Generate tests/golden/* by running the REAL reference modules (build container only).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Usage:  python -m oracle.make_golden

The reference holds no golden vectors for this path, so the fixtures are
outputs of the reference's own denoise_net.py / diffusion_ddpm.py / loss.py executed here on
seeded synthetic weights and inputs (oracle/weights.py).  Only outputs are stored; weights and
inputs are re-derived from the seeds by the tests.  /root/reference is never read at test time.
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
from .ref_loader import load_reference

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CASES = {
    # name: (net_kwargs, B, N, ctx_dim, text L)
    "uncond_bedroom": (W.UNCOND_BEDROOM, 2, 12, 128, 0),
    "uncond_living": (W.UNCOND_LIVING, 2, 21, 128, 0),
    "text_bedroom": (W.TEXT_BEDROOM, 2, 12, 128, 7),
    "rearrange_living": (W.REARRANGE_LIVING, 2, 21, 512, 0),
}


def build_ref(kw, seed=0, **diff_kwargs):
    _, dn, dd = load_reference()
    with contextlib.redirect_stdout(io.StringIO()):
        net = dn.Unet1D(**kw)
    sd = W.synth_state_dict(kw, seed)
    missing = net.load_state_dict(sd, strict=True)
    cfg = dict(objectness_dim=kw.get("objectness_dim", 1), class_dim=kw.get("class_dim", 21),
               angle_dim=kw.get("angle_dim", 1), objfeat_dim=kw.get("objfeat_dim", 0))
    cfg.update(diff_kwargs.pop("config_extra", {}))
    with contextlib.redirect_stdout(io.StringIO()):
        diff = dd.DiffusionPoint(net, cfg, **diff_kwargs)
    return net, diff


def case_inputs(name, seed=0):
    kw, B, N, ctx_dim, L = CASES[name]
    C = kw["channels"]
    if C == 5:
        x = W.synth_noise((B, N, 5), seed, "x5") * 0.5
    else:
        x = W.synth_scene_batch(B, N, kw["class_dim"], kw["objfeat_dim"], seed)
    t = torch.tensor([(37 + 411 * i) % 1000 for i in range(B)], dtype=torch.int64)
    cond = W.synth_condition(B, N, ctx_dim, seed, shared=(ctx_dim == 128)).contiguous()
    cross = W.synth_text_condition(B, L, kw.get("text_dim", 512), seed) if L else None
    return kw, x, t, cond, cross


class Replay:
    """noise_fn(size=, dtype=, device=) protocol of diffusion_ddpm.py:345,355-356 replaying a list."""

    def __init__(self, seq):
        self.seq, self.i = seq, 0

    def __call__(self, size=None, dtype=None, device=None):
        n = self.seq[self.i]
        self.i += 1
        assert tuple(n.shape) == tuple(size), (n.shape, size)
        return n.clone()


def noise_list(shapes, seed, tag):
    return [W.synth_noise(s, seed, "%s%d" % (tag, i)) for i, s in enumerate(shapes)]


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    _, dn, dd = load_reference()

    # 1. state_dict layout of the real module ------------------------------------------------
    keys = {}
    for name, (kw, *_rest) in CASES.items():
        with contextlib.redirect_stdout(io.StringIO()):
            net = dn.Unet1D(**kw)
        keys[name] = [[k, list(v.shape)] for k, v in net.state_dict().items()]
    with open(os.path.join(GOLDEN, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f)

    # 2. schedule tables ---------------------------------------------------------------------
    _, diff = build_ref(W.UNCOND_BEDROOM, time_num=1000, model_mean_type="v")
    g = diff.diffusion
    tabs = {k: getattr(g, k).numpy() for k in (
        "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
        "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
        "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
        "posterior_mean_coef1", "posterior_mean_coef2", "loss_weight")}
    np.savez_compressed(os.path.join(GOLDEN, "schedule_v_T1000.npz"), **tabs)

    # 3. Unet1D forward ----------------------------------------------------------------------
    fwd = {}
    for name in CASES:
        kw, x, t, cond, cross = case_inputs(name)
        net, _ = build_ref(kw)
        with torch.no_grad():
            fwd[name] = net(x, t, cond, cross).numpy()
        print(name, "forward", fwd[name].shape, float(np.abs(fwd[name]).mean()))
    np.savez_compressed(os.path.join(GOLDEN, "unet_forward.npz"), **fwd)

    # 4. p_losses (with IoU term) + gradients --------------------------------------------------
    stats_file = os.path.join(tempfile.mkdtemp(), "dataset_stats.txt")
    with open(stats_file, "w") as f:
        json.dump(W.DATASET_STATS, f)
    out = {}
    for name in ("uncond_bedroom", "uncond_living"):
        kw, x, t, cond, cross = case_inputs(name)
        net, diff = build_ref(kw, time_num=1000, model_mean_type="v", loss_separate=True, loss_iou=True,
                              train_stats_file=stats_file)
        noise = W.synth_noise(tuple(x.shape), 0, "train_noise")
        losses, scal = diff.diffusion.p_losses(diff._denoise, x, t, noise=noise, condition=cond, condition_cross=cross)
        loss = losses.mean()
        loss.backward()
        out[name + ".losses"] = losses.detach().numpy()
        out[name + ".loss"] = np.float32(loss.item())
        for k, v in scal.items():
            out[name + "." + k] = np.float32(v.item())
        names, gn = [], []
        for k, p in net.named_parameters():
            names.append(k)
            gn.append(float(p.grad.norm()))
        out[name + ".grad_norms"] = np.array(gn, dtype=np.float32)
        out[name + ".grad.init_conv.bias"] = net.init_conv.bias.grad.numpy().copy()
        out[name + ".grad.mid_attn.to_qkv"] = net.mid_attn.fn.fn.to_qkv.weight.grad.numpy()[:8, :16, 0].copy()
        out[name + ".grad.downs0.block1.proj"] = net.downs[0][0].block1.proj.weight.grad.numpy()[:8, :16, 0].copy()
        with open(os.path.join(GOLDEN, "grad_names_%s.json" % name), "w") as f:
            json.dump(names, f)
        print(name, "p_losses", losses.detach().numpy())
    np.savez_compressed(os.path.join(GOLDEN, "p_losses.npz"), **out)

    # 5. reverse chains ----------------------------------------------------------------------
    chains = {}
    kw, x, t, cond, cross = case_inputs("uncond_bedroom")
    B, N, C = x.shape
    for T, b in ((50, 2), (1000, 1)):
        net, diff = build_ref(kw, time_num=T, model_mean_type="v")
        seq = noise_list([(b, N, C)] * (T + 1), 1, "chain%d_" % T)
        with torch.no_grad():
            s = diff.gen_samples((b, N, C), "cpu", condition=cond[:b], condition_cross=None,
                                 noise_fn=Replay(seq), clip_denoised=True)
        chains["uncond_T%d" % T] = s.numpy()
        print("chain T=%d" % T, float(s.abs().mean()))
    # completion (first P rows given), T=50
    T, P = 50, 3
    net, diff = build_ref(kw, time_num=T, model_mean_type="v")
    shapes = [(B, N, C)]
    for _ in range(T):
        shapes += [(B, P, C), (B, N, C)]
    seq = noise_list(shapes, 2, "complete_")
    partial = x[:, :P, :].contiguous()
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        s = diff.complete_samples((B, N, C), "cpu", condition=cond, condition_cross=None,
                                  noise_fn=Replay(seq), clip_denoised=True, partial_boxes=partial)
    chains["complete_T50"] = s.numpy()
    # unclipped chain (generate_diffusion.py default clip_denoised=False), T=50
    seq = noise_list([(B, N, C)] * (T + 1), 3, "noclip_")
    with torch.no_grad():
        s = diff.gen_samples((B, N, C), "cpu", condition=cond, condition_cross=None,
                             noise_fn=Replay(seq), clip_denoised=False)
    chains["uncond_noclip_T50"] = s.numpy()
    # text chain T=20
    kwt, xt, tt, condt, crosst = case_inputs("text_bedroom")
    net, diff = build_ref(kwt, time_num=20, model_mean_type="v")
    seq = noise_list([tuple(xt.shape)] * 21, 4, "text_")
    with torch.no_grad():
        s = diff.gen_samples(tuple(xt.shape), "cpu", condition=condt, condition_cross=crosst,
                             noise_fn=Replay(seq), clip_denoised=True)
    chains["text_T20"] = s.numpy()
    # re-arrangement chain T=50 (5 diffused channels, config/rearrange/*)
    kwr, xr, tr_, condr, _ = case_inputs("rearrange_living")
    Br, Nr = xr.shape[:2]
    full = W.synth_scene_batch(Br, Nr, 25, 32, 5)
    net, diff = build_ref(kwr, time_num=50, model_mean_type="v",
                          config_extra={"room_arrange_condition": True})
    seq = noise_list([(Br, Nr, 5)] * 51, 5, "arrange_")
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        s = diff.arrange_samples((Br, Nr, 65), "cpu", condition=condr, condition_cross=None,
                                 noise_fn=Replay(seq), clip_denoised=True, input_boxes=full)
    chains["arrange_T50"] = s.numpy()
    np.savez_compressed(os.path.join(GOLDEN, "chains.npz"), **chains)
    print("golden written to", GOLDEN)


if __name__ == "__main__":
    sys.exit(main())
