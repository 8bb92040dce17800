"""Golden values for the variational-bound diagnostics (loss_type 'kl', prior_kl, all_kl) from the REAL reference
classes (diffusion_ddpm.py:94-99,511-518,657-660,679-717,734-745).  TEST INFRASTRUCTURE; build container only.

Usage:  python -m oracle.make_golden_bpd     -> tests/golden/bpd.npz
"""
import os
import sys

import numpy as np
import torch

from . import weights as W
from .make_golden import GOLDEN, build_ref, case_inputs, noise_list

T_LOOP = 20


def main():
    out = {}
    kw, x, t, cond, _ = case_inputs("uncond_bedroom")
    noise = W.synth_noise(tuple(x.shape), 11, "bpd_q")
    net, diff = build_ref(kw, time_num=1000, model_mean_type="v")
    gd = diff.diffusion
    with torch.no_grad():
        x_t = gd.q_sample(x, t, noise=noise)
        for clip in (True, False):
            kl, xr = gd._vb_terms_bpd(diff._denoise, data_start=x, data_t=x_t, t=t, condition=cond, condition_cross=None,
                                      clip_denoised=clip, return_pred_xstart=True)
            out["vb_kl_clip%d" % clip] = kl.numpy()
            out["vb_xstart_clip%d" % clip] = xr.numpy()
        out["prior_bpd"] = diff.prior_kl(x).numpy()
    # loss_type 'kl' through p_losses
    net, diff = build_ref(kw, time_num=1000, model_mean_type="v", loss_type="kl")
    with torch.no_grad():
        out["p_losses_kl"] = diff.diffusion.p_losses(diff._denoise, x, t, noise=noise, condition=cond).numpy()
    # all_kl over a T=20 process; q_sample draws replayed per timestep
    net, diff = build_ref(kw, time_num=T_LOOP, model_mean_type="v")
    seq = noise_list([tuple(x.shape)] * T_LOOP, 12, "bpd_loop")
    gd = diff.diffusion
    orig = gd.q_sample
    gd.q_sample = lambda x_start, t, noise=None: orig(x_start, t, noise=seq[int(t[0])] if noise is None else noise)
    with torch.no_grad():
        r = diff.all_kl(x, cond, None, clip_denoised=True)
    out["all_kl"] = np.array([float(r[k]) for k in ("total_bpd_b", "terms_bpd", "prior_bpd_b", "mse_bt")], dtype=np.float64)
    np.savez_compressed(os.path.join(GOLDEN, "bpd.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, float(np.abs(v).mean()))


if __name__ == "__main__":
    sys.exit(main())
