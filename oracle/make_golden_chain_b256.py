"""Golden reverse chain of the REAL reference at the HEADLINE batch (build container only; TEST INFRASTRUCTURE -- imported by tests/ only).

Usage:  python -m oracle.make_golden_chain_b256   ->  tests/golden/chain_b256.npz   (about 15 minutes on 8 CPU cores)

chain_split.npz (round 4) holds the long chains at B = 128, where every GroupNorm launch of the split-bf16 GEMM takes the FOUR-wave
tile <true,2,2,5> (csrc/gemm_split.hip select_tile: 128 eight-wave blocks < 192).  The benchmark's dominant kernel is the EIGHT-wave
tile <true,2,4,5> of B = 256, N = 80, which rounds 3-4 held to the reference for one training step and ONE reverse step only
(b256.npz).  Here: uncond living, B = 256, N = 80, C = 65 -- the reference's own p_sample_loop (diffusion_ddpm.py:355-371) through
DiffusionPoint.gen_samples, T = 200, clip_denoised=True, replayed noise.  Stored: every 32nd scene of the result (8 scenes), f64 sum /
abs-sum of the whole tensor, and the same of x_t at t = 149, 99, 49.  Weights, inputs and noise are re-derived from seeds by the
test (oracle/weights.py; chain_inputs / chain_noise below).
"""
import os
import sys
import time

import numpy as np
import torch

from . import weights as W
from .make_golden import GOLDEN, build_ref

B, N, T = 256, 80, 200
WATCH_T = (149, 99, 49)
SEED = 256


def chain_inputs():
    kw = W.UNCOND_LIVING
    x = W.synth_scene_batch(B, N, kw["class_dim"], kw["objfeat_dim"], SEED)
    cond = W.synth_condition(B, N, 128, SEED, shared=True).contiguous()
    return kw, x, cond


def chain_noise(i, shape):
    return W.synth_noise(shape, SEED, "chain_b256_%d" % i)


class LazyReplay:
    def __init__(self):
        self.i = 0

    def __call__(self, size=None, dtype=None, device=None):
        n = chain_noise(self.i, tuple(size))
        self.i += 1
        return n


def summarize(out, key, x):
    out[key + ".scenes32"] = x[::32].numpy().copy()
    out[key + ".sum"] = np.float64(x.double().sum().item())
    out[key + ".abs_sum"] = np.float64(x.double().abs().sum().item())


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    path = os.path.join(GOLDEN, "chain_b256.npz")
    out = {}
    kw, x, cond = chain_inputs()
    C = kw["channels"]
    net, diff = build_ref(kw, time_num=T, model_mean_type="v")
    inner = diff._denoise
    t0 = time.time()

    def watching(data, t, condition, condition_cross):
        ti = int(t[0])
        if ti in WATCH_T:
            summarize(out, "loop.t%d" % ti, data.detach().clone())
        if ti % 20 == 0:
            print("  loop t=%d  %.0f s" % (ti, time.time() - t0), flush=True)
        return inner(data, t, condition, condition_cross)

    diff._denoise = watching
    with torch.no_grad():
        s = diff.gen_samples((B, N, C), "cpu", condition=cond, condition_cross=None, noise_fn=LazyReplay(), clip_denoised=True)
    summarize(out, "loop.T%d" % T, s)
    print("loop T=%d: sum %.6f abs-sum %.6f (%.0f s)" % (T, out["loop.T%d.sum" % T], out["loop.T%d.abs_sum" % T], time.time() - t0))
    np.savez_compressed(path, **out)
    print("written", path)


if __name__ == "__main__":
    sys.exit(main())
