#!/usr/bin/env python
"""Evidence for the root cause of round 4's `Memory access fault by GPU` -- without faulting.

Round 1-4 bench.py built its SampleRunner (sampler._StepGraph: ~140 launches with RAW parameter pointers baked into argument structs
and a hipGraph) BEFORE the first train_on_batch.  That first training step re-homes every parameter into the flat buffer P
(flat.FlatStorage: `p.data = view of P`), which frees the 442 original parameter storages; the plan kept the nn.Parameter OBJECTS, not
the storages.  From then on every sampling replay and every launch of roofline_dominant_kernel(sr.g.plan) read freed memory: recycled
by the training plan's activations (wrong weights, same timings) or -- after torch.cuda.graph's empty_cache() had handed fully free
segments back to the driver -- unmapped (a GPU page fault, depending on the allocator's layout at that moment).

This script rebuilds that sequence and classifies every parameter pointer of the stale plan against torch.cuda.memory_snapshot():
    live      inside an allocated block that still belongs to the parameter
    recycled  inside an allocated block that now belongs to something else
    cached    inside a free block of a mapped segment (readable garbage)
    unmapped  in no segment at all (an access faults)
once as round 4 ran it (DSC_REPRO_ROUND4=1: the plan keeps Parameter objects only, no flat storage up front) and once with this
round's fixes (plans own detached aliases of the storages they point into; stale plans refuse to run).
    python tools/stale_plan_repro.py [config]         (default: arrange, the configuration that faulted)
"""
import contextlib
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def classify(ptrs, owners):
    import torch
    segs = []
    for seg in torch.cuda.memory_snapshot():
        a = seg["address"]
        blocks = []
        for b in seg["blocks"]:
            ba = b.get("address", a)
            blocks.append((ba, ba + b["size"], b["state"]))
            a = ba + b["size"]
        segs.append((seg["address"], seg["address"] + seg["total_size"], blocks))
    out = {"live": 0, "recycled": 0, "cached": 0, "unmapped": 0}
    for p in ptrs:
        where = "unmapped"
        for s0, s1, blocks in segs:
            if s0 <= p < s1:
                where = "cached"
                for b0, b1, state in blocks:
                    if b0 <= p < b1:
                        if state == "active_allocated":
                            where = "live" if any(o0 <= p < o1 for o0, o1 in owners) else "recycled"
                        break
                break
        out[where] += 1
    return out


def run(config, round4, early_flat=None):
    import torch
    import bench
    from diffuscene_amd import engine
    from diffuscene_amd.networks import optimizer_factory
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import train_on_batch
    from diffuscene_amd.sampler import _StepGraph
    dev = torch.device("cuda", 0)
    spec = dict(bench.CONFIGS[config])
    saved_own = engine._own
    if early_flat is None:
        early_flat = not round4
    bench_flat = lambda m: None                          # noqa: E731 -- round 4's build_model did not create the flat storage up front
    if round4:
        engine._own = lambda t: t                        # round 4: the plan's keep-alive list held the Parameter objects themselves
    try:
        if not early_flat:
            import unittest.mock as mock
            with mock.patch("diffuscene_amd.flat.ensure_flat", bench_flat):
                model, _ = bench.build_model(spec, dev)
        else:
            model, _ = bench.build_model(spec, dev)
        shape, cond, cross, partial, _ = bench.sampling_inputs(spec, model, dev, seed=0)
        with torch.no_grad(), contextlib.redirect_stderr(io.StringIO()):
            g = _StepGraph(model.diffusion.diffusion, model.diffusion.model, shape, dev, cond, cross, True,
                           partial_shape=None if partial is None else tuple(partial.shape))
        net = model.diffusion.model
        before = {p.data_ptr() for p in net.parameters()}
        ptrs = set()
        for _, a in g.plan.gemm_args():
            for f in ("w", "bias", "gamma", "beta"):
                v = getattr(a, f)
                if v and v in before:
                    ptrs.add(v)
        g.replay_steps(2)
        torch.cuda.synchronize()
        _, batch = bench.synth_batch(spec, dev, seed=100)
        opt = optimizer_factory({"optimizer": "Adam", "lr": 2e-4}, filter(lambda p: p.requires_grad, model.parameters()))
        for i in range(3):                               # eager step (flat storage is created here in round 4), capture, replay
            train_on_batch(model, opt, batch, {"training": {"max_grad_norm": 10}})
        torch.cuda.synchronize()
        moved = sum(1 for p in net.parameters() if p.data_ptr() not in before)
        owners = []
        if not round4:
            def walk(k):
                if isinstance(k, (tuple, list)):
                    for u in k:
                        walk(u)
                elif isinstance(k, torch.Tensor) and k.is_cuda:
                    owners.append((k.data_ptr(), k.data_ptr() + max(k.numel() * k.element_size(), 1)))
            walk(g.plan.keep)
        res = {"config": config, "plan_keeps": "Parameter objects (round 4)" if round4 else "aliases of the storages (round 5)",
               "flat_storage_created": "before the sampling graph (round 5 bench.py)" if early_flat else "by the first training step, after the capture (round 4 bench.py)",
               "parameter_pointers_in_the_sampling_plan": len(ptrs),
               "parameters_moved_by_the_first_training_step": moved, "where_the_plan's_pointers_point_now": classify(sorted(ptrs), owners)}
        if round4:
            res["replay_of_the_stale_graph"] = "not attempted (round 4 had no check: it replayed, reading whatever the pointers reach)"
        else:
            try:
                g.replay_steps(1)
                torch.cuda.synchronize()
                res["replay_of_the_stale_graph"] = "ran: no parameter moved (flat storage exists before the capture)"
            except engine.StalePlanError:
                res["replay_of_the_stale_graph"] = "refused: StalePlanError"
        return res
    finally:
        engine._own = saved_own


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "arrange"
    # round-4 behaviour is only CLASSIFIED, never replayed: the classification says which pointers a replay reads garbage through
    # ("recycled" / "cached") and which would fault ("unmapped")
    for round4, early in ((True, False), (False, False), (False, True)):
        print(json.dumps(run(config, round4, early)), flush=True)
        import gc
        import torch
        gc.collect()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
