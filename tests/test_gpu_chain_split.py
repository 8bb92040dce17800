"""GPU: LONG reverse chains at a batch whose launches run the split-bf16 kernels -- the arithmetic bench.py times -- against the REAL
reference (tests/golden/chain_split.npz, oracle/make_golden_chain_split.py: the reference's own p_sample_loop, T=1000, and
p_sample_loop_complete, T=100, at B=128, N=80 with replayed noise).

Every other multi-step golden is B=2, where each launch makes < 160 blocks and the dispatcher keeps the exact-f32 MFMA kernel; at
B=128, N=80 every GroupNorm launch makes 256 four-wave blocks and every plain 512-wide product 128 x 4 tiles of the split kernel
(asserted below through dsc_gemm_arithmetic on the plan's own argument structs).  Both arithmetics run in ONE pytest session
(per-call switch dsc_set_gemm_arithmetic), eagerly and from the captured hipGraph.  Tolerance: the north star's 1e-4 -- norm-relative
AND element-wise (test_gpu_wide.check) on every 16th scene, 1e-5 of the abs-sum on the f64 sums over the whole tensor; the watch
points at t = 749 / 499 / 249 / 99 locate a divergence along the chain."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import weights as W  # noqa: E402
from oracle.make_golden_chain_split import B, COMPLETE_P, COMPLETE_T, LOOP_T, N, WATCH_T, chain_inputs, chain_noise  # noqa: E402

from test_gpu_wide import check, dev  # noqa: E402


def _model(tmp_path, time_num):
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    kw = W.UNCOND_LIVING
    stats = tmp_path / "dataset_stats.txt"
    stats.write_text(json.dumps(W.DATASET_STATS))
    net = Unet1D(**kw)
    net.load_state_dict(W.synth_state_dict(kw))
    net.to(dev())
    diff = DiffusionPoint(net, dict(objectness_dim=0, class_dim=25, angle_dim=2, objfeat_dim=32), time_num=time_num, model_mean_type="v")
    return net, diff


def _arithmetic_of_plan(net, cond):
    """(split launches, all launches) of the GN and the 512-wide plain GEMMs of the sampling plan at this batch."""
    from diffuscene_amd import ops
    eng = net.engine(dev())
    plan = eng.prepare(B, N, cond, None)
    gn = [ops.gemm_uses_split(a, gn=True) for kind, a in plan.gemm_args() if kind == "gn"]
    plain = [(ops.gemm_uses_split(a), (a.m, a.n, a.k1 + a.k2, a.batch)) for kind, a in plan.gemm_args()
             if kind == "plain" and a.n >= 384 and a.m == B * N]
    return gn, plain


def _check_sums(y, g, key):
    s, a = float(y.double().sum()), float(y.double().abs().sum())
    want_s, want_a = float(g[key + ".sum"]), float(g[key + ".abs_sum"])
    assert abs(a - want_a) <= 1e-5 * want_a, (key, a, want_a)
    assert abs(s - want_s) <= 1e-5 * want_a, (key, s, want_s)


@pytest.mark.parametrize("gemm_arith", ["split", "f32"], indirect=True)
def test_thousand_step_chain_at_b128(golden_dir, tmp_path, gemm_arith):
    from diffuscene_amd.sampler import NoiseReplay
    g = np.load(os.path.join(golden_dir, "chain_split.npz"))
    kw, x, cond = chain_inputs()
    C = kw["channels"]
    net, diff = _model(tmp_path, LOOP_T)
    cond_d = cond[0].to(dev())[None].expand(B, -1, -1)          # the wrapper's stride-0 broadcast of the instance embedding
    gn, plain = _arithmetic_of_plan(net, cond_d)
    assert len(gn) == 56 and len(plain) >= 30, (len(gn), len(plain))
    if gemm_arith == "split":
        assert all(gn), "at B=128, N=80 every GroupNorm launch must run the split kernel"
        assert all(u for u, _ in plain), "512-wide products left on the f32 kernel: %s" % [s for u, s in plain if not u]
    else:
        assert not any(gn) and not any(u for u, _ in plain)
    noise = torch.empty((LOOP_T + 1, B, N, C), dtype=torch.float32, device=dev())
    for i in range(LOOP_T + 1):
        noise[i].copy_(chain_noise(i, (B, N, C)))
    # eager loop, with the reference's watch points: x_t as handed to the denoiser at t = 749 / 499 / 249 / 99
    seen = {}
    inner = diff._denoise

    def watching(data, t, condition, condition_cross):
        ti = int(t[0])
        if ti in WATCH_T:
            seen[ti] = data.detach().clone()
        return inner(data, t, condition, condition_cross)
    diff._denoise = watching
    with torch.no_grad():
        y = diff.gen_samples((B, N, C), dev(), condition=cond_d, noise_fn=NoiseReplay(noise), clip_denoised=True, graph=False)
    diff._denoise = inner
    for ti in WATCH_T:
        check(seen[ti][::16], g["loop.t%d.scenes16" % ti], "x_t at t=%d (%s, eager)" % (ti, gemm_arith))
        _check_sums(seen[ti], g, "loop.t%d" % ti)
    check(y[::16], g["loop.T%d.scenes16" % LOOP_T], "T=%d chain, every 16th scene (%s, eager)" % (LOOP_T, gemm_arith))
    _check_sums(y, g, "loop.T%d" % LOOP_T)
    with torch.no_grad():
        yg = diff.gen_samples((B, N, C), dev(), condition=cond_d, noise_fn=NoiseReplay(noise), clip_denoised=True, graph=True)
    check(yg[::16], g["loop.T%d.scenes16" % LOOP_T], "T=%d chain (%s, hipGraph)" % (LOOP_T, gemm_arith))
    _check_sums(yg, g, "loop.T%d" % LOOP_T)
    assert torch.equal(y, yg), "graph replay must reproduce the eager loop bit for bit"


@pytest.mark.parametrize("gemm_arith", ["split", "f32"], indirect=True)
def test_hundred_step_completion_at_b128(golden_dir, tmp_path, gemm_arith):
    from diffuscene_amd.sampler import NoiseReplay
    g = np.load(os.path.join(golden_dir, "chain_split.npz"))
    kw, x, cond = chain_inputs()
    C = kw["channels"]
    net, diff = _model(tmp_path, COMPLETE_T)
    # noise_fn call order of p_sample_loop_complete: x_T, then per step the partial-scene draw and the p_sample draw -- one counter
    main = [chain_noise(0, (B, N, C), "chain_split_c_")]
    part = []
    for s in range(COMPLETE_T):
        part.append(chain_noise(1 + 2 * s, (B, COMPLETE_P, C), "chain_split_c_"))
        main.append(chain_noise(2 + 2 * s, (B, N, C), "chain_split_c_"))
    main, part = torch.stack(main).to(dev()), torch.stack(part).to(dev())
    partial = x[:, :COMPLETE_P, :].contiguous().to(dev())
    res = []
    for graph in (False, True):
        with torch.no_grad():
            res.append(diff.complete_samples((B, N, C), dev(), condition=cond[0].to(dev())[None].expand(B, -1, -1), noise_fn=NoiseReplay(main, part),
                                             clip_denoised=True, partial_boxes=partial, graph=graph))
        check(res[-1][::16], g["complete.T%d.scenes16" % COMPLETE_T], "completion T=%d (%s, graph=%s)" % (COMPLETE_T, gemm_arith, graph))
        _check_sums(res[-1], g, "complete.T%d" % COMPLETE_T)
    assert torch.equal(res[0], res[1])
    assert torch.equal(res[1][:, :COMPLETE_P], partial)
