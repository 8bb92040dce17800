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


def test_two_hundred_step_chain_at_b256(golden_dir, tmp_path):
    """The HEADLINE batch: B = 256, N = 80 -- every GroupNorm launch on the EIGHT-wave tile <true,2,4,5> the benchmark's dominant kernel
    is (asserted through dsc_gemm_split_tile on the plan's own argument structs) -- against the real reference's own p_sample_loop
    (tests/golden/chain_b256.npz, oracle/make_golden_chain_b256.py: T = 200, replayed noise), eagerly and from the hipGraph, under
    BOTH arithmetics in one test.  And the distance BETWEEN the two arithmetics on the final scenes: the chain's sensitivity to
    rounding amplifies any difference, whichever kernel made it (the element-wise distance from the CPU reference is the chain's,
    not the kernel's: it is the same for split and exact f32); |split - f32| pins the split arithmetic itself -- two GPU runs that
    differ in nothing but the six-product split."""
    from diffuscene_amd import _lib
    from diffuscene_amd.sampler import NoiseReplay
    from oracle import make_golden_chain_b256 as G
    g = np.load(os.path.join(golden_dir, "chain_b256.npz"))
    kw, x, cond = G.chain_inputs()
    C, Bq, Nq, T = kw["channels"], G.B, G.N, G.T
    noise = torch.empty((T + 1, Bq, Nq, C), dtype=torch.float32, device=dev())
    for i in range(T + 1):
        noise[i].copy_(G.chain_noise(i, (Bq, Nq, C)))
    cond_d = cond[0].to(dev())[None].expand(Bq, -1, -1)
    results = {}
    prev = "split" if _lib.split_enabled() else "f32"
    try:
        for arith in ("split", "f32"):
            _lib.set_gemm_arithmetic(arith)
            net, diff = _model(tmp_path, T)
            eng = net.engine(dev())
            plan = eng.prepare(Bq, Nq, cond_d, None)
            tiles = [_lib.fn("dsc_gemm_split_tile")(a, 1) for kind, a in plan.gemm_args() if kind == "gn"]
            assert len(tiles) == 56
            if arith == "split":
                want = _lib.TILE_WAVE_GN if _lib.load().dsc_get_split_wave() == 1 else _lib.TILE_GN_80_W8     # (round 6: the wave-autonomous kernel by default)
                assert all(t == want for t in tiles), "B=256, N=80: every GroupNorm launch must run the benchmark's tile %d: %s" % (want, tiles)
            else:
                assert all(t == -1 for t in tiles)
            seen = {}
            inner = diff._denoise

            def watching(data, t, condition, condition_cross, seen=seen, inner=inner):
                ti = int(t[0])
                if ti in G.WATCH_T:
                    seen[ti] = data.detach().clone()
                return inner(data, t, condition, condition_cross)
            with torch.no_grad():
                yg = diff.gen_samples((Bq, Nq, C), dev(), condition=cond_d, noise_fn=NoiseReplay(noise), clip_denoised=True, graph=True)
                diff._denoise = watching
                y = diff.gen_samples((Bq, Nq, C), dev(), condition=cond_d, noise_fn=NoiseReplay(noise), clip_denoised=True, graph=False)
                diff._denoise = inner
            assert torch.equal(y, yg), "graph replay must reproduce the eager loop bit for bit (%s)" % arith
            for ti in G.WATCH_T:
                check(seen[ti][::32], g["loop.t%d.scenes32" % ti], "B=256 x_t at t=%d (%s)" % (ti, arith))
                _check_sums(seen[ti], g, "loop.t%d" % ti)
            check(y[::32], g["loop.T%d.scenes32" % T], "B=256 T=%d chain, every 32nd scene (%s)" % (T, arith))
            _check_sums(y, g, "loop.T%d" % T)
            results[arith] = y
            del net, diff, eng, plan
    finally:
        _lib.set_gemm_arithmetic(prev)
    # measured on MI355X (profiles/r05_parity.txt): 1.75e-7 norm-relative, 1.4e-6 max abs over the 200 steps -- an order of magnitude BELOW
    # either arithmetic's distance from the CPU reference (2.2e-6 / 2.9e-5 element-wise): the chain's drift from the reference is not the split's
    a, b = results["split"].double(), results["f32"].double()
    rel = float((a - b).norm() / b.norm())
    worst = float((a - b).abs().max())
    log = os.environ.get("DSC_PARITY_LOG")
    if log:
        with open(log, "a") as f:
            f.write("B=256 T=%d chain, |split - f32|: %.3e norm-relative, %.3e max abs\n" % (T, rel, worst))
    assert rel < 2e-6, "the two arithmetics drifted apart over the chain: %.3e norm-relative (max abs %.3e)" % (rel, worst)
