#!/usr/bin/env python
"""Memory-safety run of the whole hot path (test infrastructure; tests/test_gpu_guard.py drives it, `tools/gpu_round.sh guard` records it).

Every BASELINE configuration, in bench.py's order, BUILT AND DESTROYED IN ONE PROCESS, `--loops` times: the reverse loop through the real
entry point of its kind (p_sample_loop / p_sample_loop_complete / p_sample_loop_arrange) eagerly AND from the captured hipGraph, then
train_on_batch steps (eager step, graph capture, replays), then the reverse loop again on the updated weights -- i.e. every transition
that moves or frees something a launch plan or graph holds a raw pointer to (flat storage of the first training step, plan caches,
graph teardown, empty_cache between configurations).

    --mode normal   PyTorch's caching allocator (the reference result; PYTORCH_NO_CUDA_MEMORY_CACHING=1 in the environment makes it the
                    "every free really frees" variant)
    --mode vmm      tools/guard_alloc.cpp: one virtual-memory mapping per tensor, unmapped range behind it, unmap on free (an access past
                    the end of an operand or through a pointer to a freed tensor is a GPU fault), canary in front, NaN-filled fresh memory
    --mode canary   hipMalloc + canary zones on both sides (out-of-bounds writes without a fault)

Writes one JSON record (--out): per configuration the f64 checksums of every result, which must be finite and equal between modes
(tests/test_gpu_guard.py compares them), the allocator statistics and the library's out-of-range-timestep counter (must be 0).
"""
import argparse
import ctypes
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GUARD_SO = os.path.join(ROOT, "tools", "_build", "libdsc_guard_alloc.so")
ORDER = ["living80", "bedroom21", "text", "complete", "arrange", "living80:f32"]      # bench.py's default run


def install_guard(mode):
    import torch
    if not os.path.exists(GUARD_SO):
        raise SystemExit("guard_run: %s missing -- python __graft_entry__.py builds it" % GUARD_SO)
    os.environ["DSC_GUARD_MODE"] = mode
    alloc = torch.cuda.memory.CUDAPluggableAllocator(GUARD_SO, "guard_malloc", "guard_free")
    torch.cuda.memory.change_current_allocator(alloc)
    lib = ctypes.CDLL(GUARD_SO)
    lib.guard_check.restype = ctypes.c_long
    return lib


def guard_stats(lib):
    if lib is None:
        return None
    arr = (ctypes.c_long * 8)()
    lib.guard_stats(arr)
    keys = ("allocations", "frees", "live_blocks", "live_MiB", "peak_MiB", "allocations_inside_captures", "deferred_frees", "corrupted_canaries")
    return dict(zip(keys, list(arr)))


def fsum(t):
    import torch
    assert bool(torch.isfinite(t).all()), "non-finite values"
    return float(t.double().sum().item())


def run_config(what, device, T, train_steps, batch_div, log, graphs=True):
    import torch
    import bench
    from diffuscene_amd import _lib
    from diffuscene_amd.networks import optimizer_factory
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import train_on_batch
    from diffuscene_amd.sampler import NoiseReplay
    name, _, arith = what.partition(":")
    prev = _lib.set_gemm_arithmetic(arith) if arith else None
    rec = {}
    try:
        spec = dict(bench.CONFIGS[name])
        spec["batch"] = max(spec["batch"] // batch_div, 2)
        model, _ = bench.build_model(spec, device, time_num=T)
        shape, cond, cross, partial, x = bench.sampling_inputs(spec, model, device, seed=0)
        dp = model.diffusion
        g = torch.Generator().manual_seed(1234)
        buf = torch.randn((T + 1,) + tuple(shape), generator=g).to(device)
        pbuf = torch.randn((T,) + tuple(partial.shape), generator=g).to(device) if partial is not None else None

        def loop(graph):
            nf = NoiseReplay(buf, pbuf)
            with torch.no_grad():
                if spec["kind"] == "complete":
                    return dp.complete_samples(shape, device, cond, cross, noise_fn=nf, clip_denoised=True, partial_boxes=partial, graph=graph)
                if spec["kind"] == "arrange":
                    full = (shape[0], shape[1], x.shape[-1])
                    return dp.arrange_samples(full, device, cond, cross, noise_fn=nf, clip_denoised=True, input_boxes=x, graph=graph)
                return dp.gen_samples(shape, device, cond, cross, noise_fn=nf, clip_denoised=True, graph=graph)

        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):          # the loops print 'last: ...' like the reference
            e1 = loop(False)
            g1 = loop(True) if graphs else e1
            g2 = loop(True) if graphs else e1                    # the cached graph again
        assert torch.equal(e1, g1) and torch.equal(g1, g2), "hipGraph loop != eager loop"
        rec["sample"] = fsum(e1)
        # ---- training: eager step, capture, replays
        _, batch = bench.synth_batch(spec, device, seed=100)
        opt = optimizer_factory({"optimizer": "Adam", "lr": 2e-4}, filter(lambda p: p.requires_grad, model.parameters()))
        tcfg = {"training": {"max_grad_norm": 10}}
        losses = []
        for i in range(train_steps):
            torch.manual_seed(100 + i)
            losses.append(float(train_on_batch(model, opt, batch, tcfg)))
        assert all(l == l and abs(l) < 1e30 for l in losses), losses
        rec["losses"] = losses
        rec["params"] = fsum(model._dsc_flat.P)
        # ---- the reverse loop on the updated weights: the engine re-derives its weights, the cached graph must not go stale
        with contextlib.redirect_stdout(io.StringIO()):
            e3 = loop(False)
            g3 = loop(True) if graphs else e3
        assert torch.equal(e3, g3), "hipGraph loop != eager loop after training"
        rec["sample_after_training"] = fsum(e3)
        assert rec["sample_after_training"] != rec["sample"], "the optimizer steps did not reach the sampling path"
        torch.cuda.synchronize()
        rec["clamped_timesteps"] = _lib.device_error_count(reset=True)
        log("%-13s sample %.6e  losses %s  after training %.6e" % (what, rec["sample"], " ".join("%.5f" % l for l in losses),
                                                                   rec["sample_after_training"]))
        del model, opt, dp, e1, g1, g2, e3, g3, buf, pbuf, cond, cross, partial, x, batch
    finally:
        gc.collect()
        torch.cuda.empty_cache()
        if prev is not None:
            _lib.set_gemm_arithmetic(prev)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="normal", choices=["normal", "vmm", "canary"])
    ap.add_argument("--out", default=None)
    ap.add_argument("--configs", default=",".join(ORDER))
    ap.add_argument("--loops", type=int, default=1)
    ap.add_argument("--T", type=int, default=4, help="diffusion steps of the reverse loops")
    ap.add_argument("--train-steps", type=int, default=4)
    ap.add_argument("--batch-div", type=int, default=1, help="divide every configuration's batch (quick runs)")
    ap.add_argument("--no-graphs", action="store_true",
                    help="eager loops and eager training steps only (PYTORCH_NO_CUDA_MEMORY_CACHING=1 cannot allocate inside a capture)")
    args = ap.parse_args()
    if args.no_graphs:
        os.environ["DSC_TRAIN_GRAPH"] = "0"
    import torch
    t0 = time.perf_counter()

    def log(msg):
        print("[guard_run %6.1fs %s] %s" % (time.perf_counter() - t0, args.mode, msg), file=sys.stderr, flush=True)

    lib = install_guard(args.mode) if args.mode != "normal" else None
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    out = {"mode": args.mode, "no_caching": os.environ.get("PYTORCH_NO_CUDA_MEMORY_CACHING") == "1", "T": args.T,
           "train_steps": args.train_steps, "batch_div": args.batch_div, "graphs": not args.no_graphs, "loops": []}
    for lp in range(args.loops):
        row = {}
        for what in args.configs.split(","):
            row[what] = run_config(what, device, args.T, args.train_steps, args.batch_div, log, graphs=not args.no_graphs)
        out["loops"].append(row)
        st = guard_stats(lib)
        log("loop %d of %d clean%s" % (lp + 1, args.loops, "" if st is None else ": %s" % st))
    if lib is not None:
        bad = lib.guard_check()
        out["guard"] = guard_stats(lib)
        assert bad == 0, "%d canary zones were overwritten (see stderr)" % bad
    out["seconds"] = round(time.perf_counter() - t0, 1)
    txt = json.dumps(out)
    if args.out:
        with open(args.out, "w") as f:
            f.write(txt + "\n")
    print(txt, flush=True)
    log("guard run finished: every result finite, graph == eager, no access outside an operand, no use of a freed tensor")


if __name__ == "__main__":
    main()
