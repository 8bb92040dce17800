"""Memory safety of the hot path (reference loops served: diffusion_ddpm.py:355-371,447-506, train step
diffusion_scene_layout_ddpm.py:456-473).

The product bakes raw device pointers into launch plans and hipGraphs.  These tests run ALL BASELINE configurations -- reverse loops
(eager and graph), training steps (eager, capture, replay), the reverse loop again on the updated weights -- built and destroyed in
ONE process in bench.py's order (tools/guard_run.py):
  (a) under the guard allocator (tools/guard_alloc.cpp: one mapping per tensor, unmapped range behind it, unmap on free, canary in
      front, NaN-filled fresh memory) -- an access past the end of an operand or through a pointer to a freed tensor faults,
  (b) with PyTorch's caching switched off (every free is a real hipFree),
and require finite results equal to the run under the normal allocator.  Plus the two hazards found while building them: a sampling
graph that outlives the parameters' move into flat storage (round 4's bench.py did exactly that), and an out-of-range device timestep.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUN = os.path.join(ROOT, "tools", "guard_run.py")


def _guard_run(tmp_path, tag, args, env=None, timeout=900):
    out = os.path.join(str(tmp_path), tag + ".json")
    e = dict(os.environ)
    e.pop("PYTORCH_NO_CUDA_MEMORY_CACHING", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, RUN, "--out", out] + args, env=e, capture_output=True, text=True, timeout=timeout)
    log = os.environ.get("DSC_GUARD_LOG")
    if log:
        with open(log, "a") as f:
            f.write("==== %s %s (exit %d)\n%s\n" % (tag, " ".join(args), r.returncode, r.stderr[-6000:]))
    assert r.returncode == 0, "guard run %s failed (exit %d):\n%s" % (tag, r.returncode, r.stderr[-4000:])
    with open(out) as f:
        return json.load(f)


def _same(a, b, rel):
    return abs(a - b) <= rel * max(1.0, abs(a), abs(b))


def _compare(ref, got, what):
    """Results of two allocators: same kernels on the same values.  Only pointer ALIGNMENT differs (the guard allocator hands out
    16-byte aligned tensors that end their mapping; PyTorch's are 512-byte aligned), which may move ATen's vectorised reductions (the
    .mean() of the logged losses) by an ulp: 1e-6 relative on f64 checksums."""
    r0 = ref["loops"][0]
    for lp, row in enumerate(got["loops"]):
        for cfg, rec in row.items():
            for key in ("sample", "params", "sample_after_training"):
                assert _same(rec[key], r0[cfg][key], 1e-6), (what, lp, cfg, key, rec[key], r0[cfg][key])
            for x, y in zip(rec["losses"], r0[cfg]["losses"]):
                assert _same(x, y, 1e-5), (what, lp, cfg, rec["losses"], r0[cfg]["losses"])
            assert rec["clamped_timesteps"] == 0, (what, lp, cfg, "a kernel saw an out-of-range timestep")


def test_all_configs_in_one_process_under_the_guard_allocator(tmp_path):
    """bench order, twice in one process: the second pass re-uses whatever the first left behind (module-level caches, scratch
    buffers, the allocator's free lists)."""
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "_build", "libdsc_guard_alloc.so")
    if not os.path.exists(so):
        pytest.skip("the guard allocator was not built on this ROCm (tools/guard_alloc.cpp needs the HIP VMM APIs)")
    common = ["--T", "3", "--train-steps", "3"]
    ref = _guard_run(tmp_path, "normal", ["--mode", "normal", "--loops", "2"] + common)
    _compare(ref, ref, "normal allocator, second pass vs first")
    vmm = _guard_run(tmp_path, "vmm", ["--mode", "vmm", "--loops", "2"] + common)
    assert vmm["guard"]["corrupted_canaries"] == 0
    assert vmm["guard"]["allocations"] > 1000 and vmm["guard"]["frees"] > 500, vmm["guard"]
    _compare(ref, vmm, "guard allocator (vmm)")


def test_all_configs_with_allocator_caching_off(tmp_path):
    """PYTORCH_NO_CUDA_MEMORY_CACHING=1: every free returns the memory to the driver at once (no recycling hides a stale pointer);
    captures cannot allocate in this mode, so the loops and the training steps run eagerly -- same launches."""
    common = ["--T", "3", "--train-steps", "3", "--no-graphs"]
    ref = _guard_run(tmp_path, "normal_eager", ["--mode", "normal"] + common)
    got = _guard_run(tmp_path, "nocache", ["--mode", "normal"] + common, env={"PYTORCH_NO_CUDA_MEMORY_CACHING": "1"})
    assert got["no_caching"] is True
    _compare(ref, got, "caching off")


def test_canary_mode_sees_no_out_of_bounds_write(tmp_path):
    """hipMalloc + canary zones on both sides of every tensor, smaller batches (tile edges differ from the full-size runs)."""
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "_build", "libdsc_guard_alloc.so")
    if not os.path.exists(so):
        pytest.skip("the guard allocator was not built on this ROCm (tools/guard_alloc.cpp needs the HIP VMM APIs)")
    got = _guard_run(tmp_path, "canary", ["--mode", "canary", "--T", "2", "--train-steps", "2", "--batch-div", "4",
                                          "--configs", "living80,bedroom21,text,complete,arrange"])
    assert got["guard"]["corrupted_canaries"] == 0


# ------------------------------------------------------------------------------------------------ the two hazards themselves

def _small_model(tmp_path):
    import bench
    spec = dict(bench.CONFIGS["bedroom21"], batch=8)
    dev = torch.device("cuda", 0)
    import unittest.mock as mock
    with mock.patch("diffuscene_amd.flat.ensure_flat", lambda m: None):        # parameters still in their own storages (round 4's order)
        model, _ = bench.build_model(spec, dev, time_num=6)
    return bench, spec, model, dev


def test_sampling_graph_that_outlives_the_parameter_move_refuses_to_replay(tmp_path):
    """Round 1-4 bench.py: sampler._StepGraph captured, THEN the first train_on_batch re-homed every parameter into flat storage
    (the 442 old storages were freed), THEN the graph was replayed -- through dangling pointers.  Now: the plan owns the storages
    its pointers refer to (nothing dangles), a direct replay raises StalePlanError, and the public loop rebuilds its graph."""
    from diffuscene_amd import engine
    from diffuscene_amd.networks import optimizer_factory
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import train_on_batch
    from diffuscene_amd.sampler import NoiseReplay, _StepGraph
    bench, spec, model, dev = _small_model(tmp_path)
    shape, cond, cross, _, _ = bench.sampling_inputs(spec, model, dev, seed=0)
    dp = model.diffusion
    buf = torch.randn((7,) + tuple(shape), generator=torch.Generator().manual_seed(3)).to(dev)
    with torch.no_grad():
        g = _StepGraph(dp.diffusion, dp.model, shape, dev, cond, cross, True)
        first = dp.gen_samples(shape, dev, cond, cross, noise_fn=NoiseReplay(buf), graph=True)
    net = dp.model
    before = [p.data_ptr() for p in net.parameters()]
    g.replay_steps(1)                                              # current: runs
    _, batch = bench.synth_batch(spec, dev, seed=100)
    opt = optimizer_factory({"optimizer": "Adam", "lr": 0.0}, filter(lambda p: p.requires_grad, model.parameters()))
    train_on_batch(model, opt, batch, {"training": {"max_grad_norm": 10}})     # lr 0: the weights keep their values, only their storage moves
    assert sum(a != p.data_ptr() for a, p in zip(before, net.parameters())) > 300, "the training step did not re-home the parameters"
    # every pointer the stale plan holds still refers to memory the plan itself keeps alive
    kept = []

    def walk(k):
        if isinstance(k, (tuple, list)):
            for u in k:
                walk(u)
        elif isinstance(k, torch.Tensor) and k.is_cuda:
            kept.append((k.data_ptr(), k.data_ptr() + k.numel() * k.element_size()))
    walk(g.plan.keep)
    olds = set(before)
    checked = 0
    for _, a in g.plan.gemm_args():
        for f in ("w", "bias", "gamma", "beta"):
            v = getattr(a, f)
            if v in olds:
                assert any(lo <= v < hi for lo, hi in kept), "a parameter pointer of the plan is not owned by the plan"
                checked += 1
    assert checked > 100
    with pytest.raises(engine.StalePlanError):
        g.replay_steps(1)
    with pytest.raises(engine.StalePlanError):
        g.plan.run()
    with torch.no_grad():                                    # the public entry point notices, rebuilds, and (lr = 0) reproduces the result
        again = dp.gen_samples(shape, dev, cond, cross, noise_fn=NoiseReplay(buf), graph=True)
        eager = dp.gen_samples(shape, dev, cond, cross, noise_fn=NoiseReplay(buf), graph=False)
    assert torch.equal(again, first) and torch.equal(again, eager)


def test_out_of_range_device_timestep_is_clamped_and_counted():
    """The reference indexes its schedule tables with torch.gather, which raises on a bad t; our launches are asynchronous, so every
    kernel clamps the device value into the table and counts the event (dsc_device_error_count)."""
    from diffuscene_amd import _lib, ops
    dev = torch.device("cuda", 0)
    T, B, N, C = 10, 4, 5, 7
    tab = torch.linspace(0.1, 1.0, T, device=dev)
    x0, nz = torch.randn(B, N, C, device=dev), torch.randn(B, N, C, device=dev)
    _lib.device_error_count(reset=True)
    t_ok = torch.tensor([0, 3, 9, 5], device=dev)
    ref = ops.q_sample(x0, nz, t_ok, tab, tab)
    assert _lib.device_error_count() == 0
    t_bad = torch.tensor([-1, 3, 10 ** 12, 5], device=dev)      # -1: what a captured loop leaves behind; 1e12: float bits read as int64
    got = ops.q_sample(x0, nz, t_bad, tab, tab)
    t_clamped = torch.tensor([0, 3, 9, 5], device=dev)
    assert torch.equal(got, ops.q_sample(x0, nz, t_clamped, tab, tab)) and torch.equal(got, ref)
    out = ops.p_sample(x0, nz, nz, t_bad, tab, tab, tab, tab, tab, ops.MEAN_V, True)
    assert bool(torch.isfinite(out).all())
    xx = x0.clone()
    ops.complete_overwrite(xx, x0[:, :2].contiguous(), nz[:, :2].contiguous(), t_bad, tab, tab)
    assert bool(torch.isfinite(xx).all())
    assert _lib.device_error_count(reset=True) > 0
    assert _lib.device_error_count() == 0
    # the fused GroupNorm GEMM gathers its (scale, shift) row by the same vector (DSC_SS_BY_INDEX): clamped into the table
    from oracle import weights as W  # noqa: F401  (seeded helpers are not needed here; import keeps the checker importable)
    M, n, K, ntok, rows = 2 * 80 * 2, 512, 512, 80, 6
    a = torch.randn(M, K, device=dev)
    w = torch.randn(n, K, device=dev) * 0.05
    bias, gamma, beta = torch.randn(n, device=dev), torch.rand(n, device=dev) + 0.5, torch.randn(n, device=dev)
    ss = torch.randn(rows, 2 * n, device=dev)
    idx_bad = torch.tensor([-7, 2, 99, 5], device=dev)
    idx_ok = torch.tensor([0, 2, 5, 5], device=dev)
    y_bad = ops.gemm_gn_silu(a, w, bias, gamma, beta, ntok, scale_shift=ss, ss_mode=_lib.SS_BY_INDEX, ss_index=idx_bad)
    y_ok = ops.gemm_gn_silu(a, w, bias, gamma, beta, ntok, scale_shift=ss, ss_mode=_lib.SS_BY_INDEX, ss_index=idx_ok)
    assert torch.equal(y_bad, y_ok)
    # the same on the split-bf16 kernel (a launch large enough for it: 256 scenes of 80)
    if _lib.split_enabled():
        Bq = 256
        a = torch.randn(Bq * ntok, K, device=dev)
        (pl,) = ops.split_planes([(w, None, False)])
        idx_bad = torch.randint(0, rows, (Bq,), generator=torch.Generator().manual_seed(1)).to(dev)
        idx_ok = idx_bad.clone()
        idx_bad[0], idx_bad[100], idx_bad[255] = -1, rows, 2 ** 40
        idx_ok[0], idx_ok[100], idx_ok[255] = 0, rows - 1, rows - 1
        g = ops.make_gemm_args(a, w, torch.empty(Bq * ntok, n, device=dev), bias, gamma=gamma, beta=beta, tokens_per_scene=ntok,
                               scale_shift=ss, ss_mode=_lib.SS_BY_INDEX, ss_index=idx_bad, w_planes=pl)
        assert ops.gemm_uses_split(g, gn=True)
        y_bad = ops.gemm_gn_silu(a, w, bias, gamma, beta, ntok, scale_shift=ss, ss_mode=_lib.SS_BY_INDEX, ss_index=idx_bad, w_planes=pl)
        y_ok = ops.gemm_gn_silu(a, w, bias, gamma, beta, ntok, scale_shift=ss, ss_mode=_lib.SS_BY_INDEX, ss_index=idx_ok, w_planes=pl)
        assert torch.equal(y_bad, y_ok)
