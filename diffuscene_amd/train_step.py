"""One DDPM training / validation step of ``DiffusionSceneLayout_DDPM`` on the static training plan (train_plan.py).

``train_on_batch`` (reference diffusion_scene_layout_ddpm.py:456-473) keeps its semantics -- zero_grad, loss, backward,
clip_grad_norm_(max_grad_norm), optimizer step, the 11 logged scalars -- but the work between the RNG draws and the optimizer
is ONE hipGraph replay on a single GPU:

    host: build target / conditioning (reference :131-226), draw t and the noise in the reference's RNG order
    graph: q_sample -> Unet1D forward -> p_losses (+ d loss / d out) -> backward of every layer -> gradients in flat G
    host: (only for conditioning computed by torch modules: fc_text_f, fc_arrange_condition ...) autograd through them
    data parallel: the same launch list cut into a few hipGraph SEGMENTS at the launches that finish a bucket of G; the bucket
        all-reduces (RCCL) are issued between the segment replays and run beside the rest of the backward
    FusedAdam: gradient norm + clip coefficient + Adam sweep on the device; one device->host copy for the logged scalars
"""
import os

import torch

from ._lib import SS_NONE, SS_PER_SLOT, SS_PER_TOKEN
from .flat import ensure_flat

_PART_KEYS = ('loss.bbox', 'loss.trans', 'loss.size', 'loss.angle', 'loss.class', 'loss.object', 'loss.objfeat',
              'loss.liou', 'loss.bbox_iou')


DDP_FLUSH_SCHEDULES = ("single", "block", "thirds")
_warned_end = False


def ddp_flush_schedule(world=1):
    """The data-parallel flush schedule named by DSC_DDP_FLUSH (PlanRunner.plan_for documents the three), or 'auto': the runner times
    all three on the first training steps and keeps the fastest (the default with more than one rank, round 6; a single rank --
    DSC_DDP_FORCE tests -- defaults to 'single').  'end' is a deprecated alias of 'single' (rounds 4 used it for what is now 'thirds')."""
    global _warned_end
    flush = os.environ.get("DSC_DDP_FLUSH", "auto" if world > 1 else "single")
    if flush == "end":
        if not _warned_end:
            import warnings
            warnings.warn("DSC_DDP_FLUSH=end is deprecated: it now means 'single' (no overlap; rounds 1-3) -- round 4 used the name for the "
                          "three-flush overlap schedule, which is 'thirds' now.  Say 'single', 'block', 'thirds' or 'auto'.", stacklevel=2)
            _warned_end = True
        flush = "single"
    if flush not in DDP_FLUSH_SCHEDULES + ("auto",):
        raise ValueError("DSC_DDP_FLUSH=%r: must be 'auto', 'single' (alias 'end'), 'block' or 'thirds'" % flush)
    return flush


class ScheduleTuner:
    """DSC_DDP_FLUSH=auto: the data-parallel runner tries every flush schedule on the first training steps -- WARM steps to build, capture
    and replay the plan, then TIMED steps on the wall clock (start of a step to the start of the next: exchange, clip and Adam
    included) --, the ranks agree on the per-schedule time with one all_reduce(MAX) each, and every rank keeps the same fastest
    schedule from then on.  Every step is a real training step under any schedule (same gradients); what the 18 steps cost is two
    extra plan builds.  ``interval`` (tests) replaces the measured seconds of a step by ``interval(schedule)``."""
    WARM, TIMED = 3, 3

    def __init__(self, device, interval=None, log=None):
        self.device, self.interval, self.log = device, interval, log
        self.order = list(DDP_FLUSH_SCHEDULES)
        self.i, self.n, self.acc, self.last = 0, 0, 0.0, None
        self.totals, self.choice = {}, None

    def current(self):
        return self.choice or self.order[self.i]

    def _now(self):
        import time
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        return time.perf_counter()

    def tick(self):
        """At the start of every training step while the choice is open."""
        if self.choice is not None:
            return
        import torch.distributed as dist
        now = self._now()
        if self.last is not None and self.n > self.WARM:
            self.acc += (now - self.last) if self.interval is None else self.interval(self.order[self.i])
        if self.n == self.WARM + self.TIMED:
            t = torch.tensor([self.acc / self.TIMED], dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)              # the slowest rank sets a schedule's step time
            self.totals[self.order[self.i]] = float(t.item())
            self.i, self.n, self.acc = self.i + 1, 0, 0.0
            if self.i == len(self.order):
                self.choice = min(self.order, key=lambda k: (self.totals[k], self.order.index(k)))
                if self.log is not None:
                    self.log("data-parallel flush schedule: %s (%s ms per step, max over ranks)" % (
                        self.choice, ", ".join("%s %.2f" % (k, 1e3 * self.totals[k]) for k in self.order)))
            now = self._now()                                     # the next schedule's first step starts its own clock
        self.n += 1
        self.last = now


def plan_supported(model):
    """The static plan covers every shipped training configuration; anything else keeps the autograd path (same kernels)."""
    if os.environ.get("DSC_TRAIN_PLAN", "1") == "0":
        return False
    d = model.diffusion.diffusion
    net = model.diffusion.model
    p = next(model.parameters())
    if not p.is_cuda or model.room_mask_condition:
        return False                    # the room-mask feature extractor is out of scope (networks/feature_extractors.py)
    # room_partial_condition (reference :193-199; no shipped YAML sets it): fc_partial_condition(target * mask) is one more torch module
    # in front of the plan, like fc_arrange_condition -- its rows enter as per-token context and its gradient leaves through d_ctx
    if d.loss_type != 'mse' or d.translation_dim != 3 or model.sample_num_points > 160:
        return False
    if not all(q.requires_grad for q in net.parameters()):
        return False                    # a frozen denoiser parameter has no slice of G: autograd path
    full = model.bbox_dim + model.class_dim + model.objectness_dim + model.objfeat_dim
    if model.room_arrange_condition:
        return net.channels == model.translation_dim + model.angle_dim
    return d.size_dim == 3 and model.config["point_dim"] == full and net.channels == full


class PlanRunner:
    """Plans + captured graphs of one model (one per batch signature)."""

    # what lowers a plan's ops to launches: HipBackend (libdiffuscene_hip.so).  tests/ substitute the torch backend of
    # tests/plan_sim.py to drive this runner -- plan cache, reducer, segments, broadcast -- under gloo on CPU; the product never does.
    backend_factory = None
    tuner_interval = None        # tests: seconds of a timed step per schedule instead of the wall clock (ScheduleTuner)

    def __init__(self, model):
        self.model = model
        self.flat = ensure_flat(model)
        self.plans = {}                   # insertion-ordered: least recently used first
        self.last_key = None
        self.synced = False
        self.tuner = None                 # DSC_DDP_FLUSH=auto (default with world > 1): ScheduleTuner

    def __deepcopy__(self, memo):         # plans hold ctypes tables and captured graphs: a copied model builds its own
        return None

    def __reduce__(self):
        return (type(None), ())

    def world(self):
        import torch.distributed as dist
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def distributed(self):
        """Reducer active: more than one rank (DSC_DDP_FORCE=1 also drives it with a single-rank process group, for tests)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size() > 1 or os.environ.get("DSC_DDP_FORCE", "0") == "1"

    def _tuner(self):
        if self.tuner is None:
            import torch.distributed as dist
            rank0 = dist.get_rank() == 0

            def say(msg, rank0=rank0):
                if rank0:
                    import sys
                    print("[diffuscene_amd] " + msg, file=sys.stderr, flush=True)
            self.tuner = ScheduleTuner(self.flat.device, PlanRunner.tuner_interval, say)
        return self.tuner

    def begin_step(self, backward):
        """Start of a training step: with DSC_DDP_FLUSH=auto and more than one rank the schedule tuner takes its clock reading here (and
        may move on to the next schedule, or settle), before the step's plan is looked up."""
        if backward and self.distributed() and ddp_flush_schedule(self.world()) == "auto":
            self._tuner().tick()

    def budget_bytes(self):
        """Activation memory all cached plans together may hold (a plan owns every activation and gradient of its signature:
        ~20 GB at B=256, N=80; tens of MB for a small last batch)."""
        if self.flat.device.type != "cuda":
            return 1 << 62
        total = torch.cuda.get_device_properties(self.flat.device).total_memory
        return int(float(os.environ.get("DSC_PLAN_CACHE_FRAC", "0.35")) * total)

    def plan_for(self, B, N, ctx_mode, ctx_dim, L, text_dim, ctx_param):
        from .train_plan import HipBackend, TrainPlan
        ws = self.world()
        distributed = self.distributed()
        # data parallel, DSC_DDP_FLUSH = when the grouped weight-gradient launches run (the single-GPU plan keeps ONE at the end of the
        # backward); measured on one GPU with a world-1 RCCL group (bench.py --ddp-selftest, profiles/r04_bench_ddp_selftest.json; B = 256 /
        # B = 32 scenes per GPU, over the single-GPU graph step):
        #   "single" (default) the single-GPU schedule unchanged: cheapest compute, NO overlap (every bucket finishes with the
        #                      last launch, the whole exchange is exposed)                                              +1.5 % /  +1.8 %
        #   "block"            a flush every few ResnetBlocks, 7 graph segments: buckets leave all through the backward   +5.2 % / +10.8 %
        #   "thirds"           whenever a third of G is pending: 3 launches, two thirds of the exchange can overlap       +9.2 % /  +8.7 %
        #   "end"              alias of "single" (what the name meant up to round 3: the single-GPU schedule, flushed at the end)
        # The default is the cheapest schedule MEASURED: no multi-GPU run exists yet that shows the overlap of "block" paying
        # for its extra launches; `bench.py --gpus N` times all three on the node and uses the fastest.
        flush = ddp_flush_schedule(ws)
        if flush == "auto":
            flush = self._tuner().current() if distributed else "single"
        per_block = distributed and flush == "block"
        thirds = distributed and flush == "thirds"
        from ._lib import gemm_mode
        arith = gemm_mode() if PlanRunner.backend_factory is None else None      # plans bake the arithmetic and kernel family in (planes, TN form)
        key = (B, N, ctx_mode, ctx_dim, L, text_dim, ctx_param is not None, ws, distributed, per_block, thirds, arith)
        ent = self.plans.pop(key, None)
        if ent is None:
            model = self.model
            dev = self.flat.device
            # least recently used signatures go first (text batches change L, the last batch of an epoch changes B);
            # at least the two most recent ones stay
            budget = self.budget_bytes()
            while self.plans and (len(self.plans) >= int(os.environ.get("DSC_PLAN_CACHE_MAX", "16")) or
                                  (len(self.plans) >= 2 and sum(e["plan"].bytes for e in self.plans.values()) > budget)):
                self.plans.pop(next(iter(self.plans)))
            backend = HipBackend(dev) if PlanRunner.backend_factory is None else PlanRunner.backend_factory(dev)
            plan = TrainPlan(model.diffusion.model, self.flat, model.diffusion.diffusion, B, N, ctx_mode, ctx_dim, L,
                             text_dim, backend, per_block_grads=per_block, ctx_param=ctx_param,
                             grad_scale=1.0 / (B * ws),
                             tn_flush_floats=self.flat.numel // 3 if thirds else None)
            ent = {"plan": plan, "graph": None, "reducer": None, "warm": 0}
            if distributed:
                from .ddp import FlatGradientReducer
                ent["reducer"] = FlatGradientReducer(self.flat, plan)
        self.plans[key] = ent                       # (re-)inserted last = most recently used
        if key != self.last_key:
            # parameters the previous signature wrote but this one never touches (cross-attention without text, ...) must not
            # keep a stale gradient for Adam: the reference leaves those .grad None
            if self.last_key is not None:
                self.flat.G.zero_()
            self.last_key = key
        return ent

    def run(self, ent, backward):
        plan = ent["plan"]
        if not backward:
            plan.run_forward()
            return
        red = ent["reducer"]
        eager = os.environ.get("DSC_TRAIN_GRAPH", "1") == "0"
        if ent["graph"] is None and not eager and not ent.get("eager_only") and ent["warm"] >= 1:
            ent["graph"] = _capture(plan, red, self.flat.device)
            if ent["graph"] is None:
                eager = ent["eager_only"] = True        # sticky: a failed capture is not retried on every step
        if eager or ent.get("eager_only") or ent["graph"] is None:
            # first step (and DSC_TRAIN_GRAPH=0) eagerly: loads every code object, surfaces launch errors with a Python stack
            ent["warm"] += 1
            plan.run_forward()
            plan.run_backward(on_progress=red.on_progress if red is not None else None)
            return
        ent["graph"].replay()


class _StepGraphs:
    """The captured training step.  Single GPU: ONE hipGraph (forward + loss + backward).  Data parallel: the same launch list
    cut after every launch that completes a gradient bucket -- segment 0 = forward + loss + backward up to the first cut -- each
    segment its own hipGraph; after replaying a segment the reducer issues the all-reduce of the buckets it finished (eager RCCL
    launches on the collective stream, ordered behind the segment by an event), so the exchange runs beside the remaining
    segments and no collective is ever inside a graph."""

    def __init__(self, plan, reducer):
        self.plan, self.reducer = plan, reducer
        cuts = sorted(reducer.at_launch) if reducer is not None else []
        self.segments, lo = [], 0                     # (first backward launch, one past the last, cut index or None)
        for c in cuts:
            self.segments.append((lo, c + 1, c))
            lo = c + 1
        if lo < len(plan.bwd) or not self.segments:
            self.segments.append((lo, len(plan.bwd), None))
        self.graphs = []

    def capture(self, make_graph):
        for i, (lo, hi, _) in enumerate(self.segments):
            def body(i=i, lo=lo, hi=hi):
                if i == 0:
                    self.plan.run_forward()
                self.plan.run_backward_range(lo, hi)
            self.graphs.append(make_graph(body))

    def replay(self):
        for g, (_, _, cut) in zip(self.graphs, self.segments):
            g.replay()
            if cut is not None:
                self.reducer.on_progress(cut)


class _EagerSegment:
    """Stand-in for a captured segment on devices without graphs (the CPU plan backend of the tests)."""

    def __init__(self, body):
        self.body = body

    def replay(self):
        self.body()


def _capture(plan, reducer, device):
    """-> _StepGraphs, or None when the capture failed (the step then stays eager: same launches, same results)."""
    sg = _StepGraphs(plan, reducer)
    if device.type != "cuda":
        sg.capture(_EagerSegment)
        return sg

    def make(body):
        g = torch.cuda.CUDAGraph()
        # thread_local: CUDA calls of other host threads (a DataLoader's pin-memory thread) do not invalidate the capture
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            body()
        return g
    try:
        sg.capture(make)
    except Exception as e:                                  # noqa: BLE001 -- any capture failure: keep training eagerly
        import warnings
        warnings.warn("diffuscene_amd: hipGraph capture of the training step failed (%s: %s); running it eagerly"
                      % (type(e).__name__, e))
        torch.cuda.synchronize(device)
        return None
    torch.cuda.current_stream(device).synchronize()
    return sg


def _runner(model):
    r = getattr(model, "_dsc_plan_runner", None)
    if r is None or not r.flat.valid() or r.flat is not getattr(model, "_dsc_flat", None):
        r = PlanRunner(model)
        object.__setattr__(model, "_dsc_plan_runner", r)
    r.flat.attach_grads()
    return r


def loss_step(model, sample_params, backward):
    """-> (loss 0-d device tensor, loss_dict of 0-d device tensors, runner entry).  With ``backward`` the gradients of
    every trainable parameter are in the flat buffer G afterwards (averaged over the ranks' batches when distributed)."""
    r = _runner(model)
    fl = r.flat
    ws = r.world()
    if ws > 1 and not r.synced:
        from .ddp import broadcast_parameters
        broadcast_parameters(model)
        r.synced = True
    grad_ctx = torch.enable_grad() if backward else torch.no_grad()
    with grad_ctx:
        target, condition, condition_cross = model._loss_inputs(sample_params)
    B, N, C = target.shape
    device = target.device
    # RNG draws of get_loss_iter / p_losses (diffusion_ddpm.py:758-772, :527): t first, then the noise
    T = model.diffusion.diffusion.num_timesteps
    t = torch.randint(0, T, size=(B,), device=device)
    noise = torch.randn(target.shape, dtype=target.dtype, device=device)
    # ---- conditioning signature
    ctx_mode, ctx_dim, ctx_param, ctx_rows = SS_NONE, 0, None, None
    if condition is not None:
        ctx_dim = condition.shape[-1]
        shared = condition.stride(0) == 0 or B == 1
        pe = getattr(model, "positional_embedding", None)
        if shared and pe is not None and condition.data_ptr() == pe.data_ptr() and tuple(condition.shape[1:]) == tuple(pe.shape):
            ctx_mode, ctx_param = SS_PER_SLOT, pe            # learnable instance embedding read in place, gradient -> G
        elif shared:
            ctx_mode, ctx_rows = SS_PER_SLOT, condition[0]
        else:
            ctx_mode, ctx_rows = SS_PER_TOKEN, condition.reshape(B * N, ctx_dim)
    L = text_dim = 0
    cross_rows = None
    if condition_cross is not None and model.diffusion.model.text_condition:
        L, text_dim = condition_cross.shape[1], condition_cross.shape[2]
        cross_rows = condition_cross.reshape(B * L, text_dim)
    r.begin_step(backward)
    ent = r.plan_for(B, N, ctx_mode, ctx_dim, L, text_dim, ctx_param)
    plan = ent["plan"]
    with torch.no_grad():
        plan.x0.copy_(target)
        plan.noise.copy_(noise)
        plan.t.copy_(t)
        if ctx_rows is not None:
            plan.ctx_in.t.copy_(ctx_rows)
        if cross_rows is not None:
            plan.cross_in.t.copy_(cross_rows)
    tail = [(x, g) for x, g in ((ctx_rows, plan.d_ctx), (cross_rows, plan.d_cross))
            if backward and x is not None and x.requires_grad]
    if backward:
        fl.zero_head()                      # autograd ACCUMULATES into the wrapper-level gradients (tiny region of G)
    r.run(ent, backward)
    if tail:
        torch.autograd.backward([x for x, _ in tail], [g.view_as(x) for x, g in tail])
    if backward and ent["reducer"] is not None:
        ent["reducer"].finish()
    loss = plan.losses.mean()
    means = plan.parts.mean(dim=0)
    if model.room_arrange_condition:
        loss_dict = {'loss.trans': means[1], 'loss.angle': means[3]}
    else:
        loss_dict = {k: means[i] for i, k in enumerate(_PART_KEYS)}
    return loss, loss_dict, ent
