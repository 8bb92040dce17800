"""Golden LONG reverse chains of the REAL reference at a batch whose launches take the split-bf16 kernels (build container only;
TEST INFRASTRUCTURE -- imported by tests/ only).

Usage:  python -m oracle.make_golden_chain_split [--steps 1000] [--only loop|complete]   ->  tests/golden/chain_split.npz
        (about 30 + 4 minutes on 8 CPU cores)

Every multi-step golden of rounds 1-3 is B=2: its GEMM launches make < 160 blocks and stay on the exact-f32 MFMA kernel
(csrc/gemm_split.hip, MIN_BLOCKS), so the arithmetic the benchmark times (operands split 3 x bf16, six products) was held to the
reference for ONE reverse step (b256.npz) and a 10-step completion (fullbatch.npz) only.  Here:
  loop       uncond living, B=128, N=80, C=65: the reference's own p_sample_loop (diffusion_ddpm.py:355-371) through
             DiffusionPoint.gen_samples, T=1000, clip_denoised=True, replayed noise -- at this batch every GroupNorm launch makes
             256 four-wave blocks of the split kernel (the test asserts it through dsc_gemm_arithmetic)
  complete   p_sample_loop_complete (:447-476), same batch, T=100, 20 given objects
Stored: every 16th scene of the result (8 scenes), f64 sum / abs-sum over the whole tensor, and for `loop` the same three of x_t at
t = 749, 499, 249, 99 (taken from the x_t the reference hands to its denoise_fn) so a divergence can be located.
Weights and inputs are re-derived from seeds by the tests (oracle/weights.py, chain_inputs / chain_noise below).
"""
import argparse
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

from . import weights as W
from .make_golden import GOLDEN, build_ref

B, N = 128, 80
LOOP_T = 1000
COMPLETE_T, COMPLETE_P = 100, 20
WATCH_T = (749, 499, 249, 99)
SEED = 80


def chain_inputs():
    kw = W.UNCOND_LIVING
    x = W.synth_scene_batch(B, N, kw["class_dim"], kw["objfeat_dim"], SEED)
    cond = W.synth_condition(B, N, 128, SEED, shared=True).contiguous()
    return kw, x, cond


def chain_noise(i, shape, tag="chain_split_"):
    """The i-th draw of a chain, from its own generator: draws are made one at a time (1001 x 2.7 MB never sit in a list)."""
    return W.synth_noise(shape, SEED, "%s%d" % (tag, i))


class LazyReplay:
    """noise_fn(size=, dtype=, device=) protocol of diffusion_ddpm.py:345,355-356; the i-th call returns chain_noise(i) --
    completion alternates partial-scene and full draws, both counted by the same index (the order the reference calls in)."""

    def __init__(self, tag):
        self.tag, self.i = tag, 0

    def __call__(self, size=None, dtype=None, device=None):
        n = chain_noise(self.i, tuple(size), self.tag)
        self.i += 1
        return n


def summarize(out, key, x):
    out[key + ".scenes16"] = x[::16].numpy().copy()
    out[key + ".sum"] = np.float64(x.double().sum().item())
    out[key + ".abs_sum"] = np.float64(x.double().abs().sum().item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=LOOP_T)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    path = os.path.join(GOLDEN, "chain_split.npz")
    out = dict(np.load(path)) if os.path.isfile(path) else {}
    kw, x, cond = chain_inputs()
    C = kw["channels"]

    if a.only in ("", "loop"):
        T = a.steps
        net, diff = build_ref(kw, time_num=T, model_mean_type="v")
        inner = diff._denoise
        t0 = time.time()

        def watching(data, t, condition, condition_cross):
            ti = int(t[0])
            if ti in WATCH_T and T == LOOP_T:
                summarize(out, "loop.t%d" % ti, data.detach().clone())
            if ti % 50 == 0:
                print("  loop t=%d  %.0f s" % (ti, time.time() - t0), flush=True)
            return inner(data, t, condition, condition_cross)

        diff._denoise = watching
        with torch.no_grad():
            s = diff.gen_samples((B, N, C), "cpu", condition=cond, condition_cross=None, noise_fn=LazyReplay("chain_split_"),
                                 clip_denoised=True)
        summarize(out, "loop.T%d" % T, s)
        print("loop T=%d: sum %.6f abs-sum %.6f (%.0f s)" % (T, out["loop.T%d.sum" % T], out["loop.T%d.abs_sum" % T], time.time() - t0))
        np.savez_compressed(path, **out)

    if a.only in ("", "complete"):
        net, diff = build_ref(kw, time_num=COMPLETE_T, model_mean_type="v")
        t0 = time.time()
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            s = diff.complete_samples((B, N, C), "cpu", condition=cond, condition_cross=None, noise_fn=LazyReplay("chain_split_c_"),
                                      clip_denoised=True, partial_boxes=x[:, :COMPLETE_P, :].contiguous())
        summarize(out, "complete.T%d" % COMPLETE_T, s)
        print("complete T=%d: sum %.6f abs-sum %.6f (%.0f s)" % (COMPLETE_T, out["complete.T%d.sum" % COMPLETE_T],
                                                                 out["complete.T%d.abs_sum" % COMPLETE_T], time.time() - t0))
        np.savez_compressed(path, **out)
    print("written", path)


if __name__ == "__main__":
    sys.exit(main())
