"""Data-parallel gradient step for the DDPM training path (new work: the reference is single-device and never
calls torch.distributed).

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for tests).
Scenes are independent, so the batch is sharded over ranks with no data-path collective; the ONLY exchange is
one summed all-reduce of the 77.7 M fp32 gradients per step (the 1/world factor is folded into the loss gradient).

``FlatGradientReducer`` (default, with the static training plan): gradients live in ONE flat buffer G (flat.py); G is cut
into >= 8 contiguous buckets (~39 MB: xGMI is point-to-point and per-link bandwidth-bound, so few large messages beat many
small ones, but a bucket must be finished by the backward before it can leave) and each bucket is all-reduced IN PLACE -- no
pack, no copy-back -- as soon as the launch that finishes it has been enqueued, while the rest of the backward keeps
running on the compute stream.  The plan fixes the bucket -> launch schedule at build time, identically on every rank.

``OverlappedGradientReducer`` / ``average_gradients`` serve the autograd fallback path (per-parameter gradients): buckets are
packed from gradient hooks during backward.  The global-norm clip always runs on the reduced gradients, identical on every
rank, with no extra collective.  ``broadcast_parameters`` makes every replica start from rank 0's weights.
"""
import os

import torch
import torch.distributed as dist

BUCKET_BYTES = int(os.environ.get("DSC_DDP_BUCKET_MB", "128")) << 20


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _buckets(params, bucket_bytes):
    cur, size = [], 0
    for p in params:
        n = p.grad.numel() * p.grad.element_size()
        if cur and size + n > bucket_bytes:
            yield cur
            cur, size = [], 0
        cur.append(p)
        size += n
    if cur:
        yield cur


def average_gradients(model, bucket_bytes=None):
    """In-place mean of .grad over all ranks (no-op when not distributed).  Frozen sub-modules (BERT) are skipped."""
    ws = world()
    if ws == 1:
        return 0
    params = [p for p in model.parameters() if p.requires_grad and p.grad is not None]
    pending = []
    for bucket in _buckets(params, bucket_bytes or BUCKET_BYTES):
        flat = torch.cat([p.grad.reshape(-1) for p in bucket])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        pending.append((work, flat, bucket))
    inv = 1.0 / ws
    for work, flat, bucket in pending:
        work.wait()
        off = 0
        for p in bucket:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad)).mul_(inv)
            off += n
    return len(pending)


def clip_grad_norm_fused(parameters, max_norm):
    """torch.nn.utils.clip_grad_norm_ semantics (L2, clip coefficient max_norm / (norm + 1e-6) clamped to 1) with
    multi-tensor kernels and no host synchronisation; returns the total norm as a device tensor."""
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return torch.zeros(())
    norms = torch._foreach_norm(grads, 2.0)
    total = torch.linalg.vector_norm(torch.stack(norms), 2.0)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    torch._foreach_mul_(grads, coef)
    return total


class OverlappedGradientReducer:
    """All-reduce overlapped with backward: parameters are bucketed in reverse registration order (the order
    backward produces their gradients); a post-accumulate-grad hook per parameter counts a bucket's gradients in and, when
    the bucket is complete, packs it and launches its asynchronous all-reduce while backward keeps running on the compute
    stream.  ``finish()`` (after ``loss.backward()``) flushes incomplete buckets (parameters that received no gradient),
    waits for the collectives and writes the averaged gradients back.  Bucket composition is a pure function of the
    module structure, so every rank issues the same collectives in the same order."""

    def __init__(self, model, bucket_bytes=None):
        self.ws = world()
        self.bucket_bytes = bucket_bytes or BUCKET_BYTES
        params = [p for p in model.parameters() if p.requires_grad]
        self.buckets = []
        cur, size = [], 0
        for p in reversed(params):
            n = p.numel() * p.element_size()
            if cur and size + n > self.bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += n
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {id(p): bi for bi, b in enumerate(self.buckets) for p in b}
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._pending = []
        self.launched_during_backward = 0
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in params]

    def _hook(self, p):
        if self.ws == 1:
            return
        bi = self._bucket_of[id(p)]
        self._ready[bi] += 1
        if self._ready[bi] == len(self.buckets[bi]) and not self._launched[bi]:
            self._launch(bi)
            self.launched_during_backward += 1

    def _launch(self, bi):
        ps = [p for p in self.buckets[bi] if p.grad is not None]
        self._launched[bi] = True
        if not ps:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        self._pending.append((work, flat, ps))

    def finish(self):
        """-> number of bucket all-reduces of this step."""
        if self.ws == 1:
            return 0
        for bi in range(len(self.buckets)):
            if not self._launched[bi]:
                self._launch(bi)
        inv = 1.0 / self.ws
        n = len(self._pending)
        for work, flat, ps in self._pending:
            work.wait()
            flat.mul_(inv)
            torch._foreach_copy_([p.grad for p in ps], [t.view_as(p.grad) for t, p in zip(flat.split([p.grad.numel() for p in ps]), ps)])
        self._pending = []
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        return n

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def overlapped_reducer(model, bucket_bytes=None):
    """The reducer attached to ``model`` (created on first use; None when not distributed)."""
    if world() == 1 or os.environ.get("DSC_DDP_OVERLAP", "1") == "0":
        return None
    r = getattr(model, "_dsc_grad_reducer", None)
    if r is None:
        r = OverlappedGradientReducer(model, bucket_bytes)
        object.__setattr__(model, "_dsc_grad_reducer", r)
    return r


def shard_batch(sample_params, rank=None, ws=None):
    """Contiguous shard of every (B, ...) tensor of a collated batch for this rank (B must divide evenly so that the
    mean of per-rank mean losses equals the global-batch mean)."""
    ws = ws or world()
    if ws == 1:
        return sample_params
    rank = dist.get_rank() if rank is None else rank
    out = {}
    for k, v in sample_params.items():
        if torch.is_tensor(v):
            b = v.shape[0]
            if b % ws:
                raise ValueError("batch %d not divisible by world size %d" % (b, ws))
            out[k] = v[rank * (b // ws):(rank + 1) * (b // ws)]
        elif isinstance(v, (list, tuple)):
            b = len(v)
            out[k] = v[rank * (b // ws):(rank + 1) * (b // ws)]
        else:
            out[k] = v
    return out


def init_from_env(backend=None):
    """torchrun-style bootstrap: one process per GPU, LOCAL_RANK selects the device."""
    if not dist.is_available() or dist.is_initialized() or int(os.environ.get("WORLD_SIZE", "1")) < 2:
        return world()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend)
    return world()


# ---------------------------------------------------------------------------------------------------------------------
# flat-buffer path (static training plan)
# ---------------------------------------------------------------------------------------------------------------------
N_BUCKETS = int(os.environ.get("DSC_DDP_BUCKETS", "8"))


class FlatGradientReducer:
    """In-place bucketed all-reduce of the flat gradient buffer, driven by the training plan's launch schedule."""

    def __init__(self, flat, plan, n_buckets=None):
        self.flat = flat
        self.buckets = flat.buckets(n_buckets or N_BUCKETS)
        sched = plan.bucket_schedule(self.buckets)
        self.at_launch = {}
        self.at_finish = list(sched.get(None, []))
        for idx, bs in sched.items():
            if idx is None:
                continue
            for b in bs:
                # buckets that reach into the wrapper-level region also receive gradients from autograd AFTER the plan ran
                if self.buckets[b][0] < flat.head_floats:
                    self.at_finish.append(b)
                else:
                    self.at_launch.setdefault(idx, []).append(b)
        self.pending = []
        self.launched_during_backward = 0
        self.order = []                       # bucket ids in launch order of the last step (tests)

    def _launch(self, b):
        s, e = self.buckets[b]
        self.pending.append(dist.all_reduce(self.flat.G[s:e], op=dist.ReduceOp.SUM, async_op=True))
        self.order.append(b)

    def on_progress(self, i):
        bs = self.at_launch.get(i)
        if bs:
            for b in bs:
                self._launch(b)
                self.launched_during_backward += 1

    def finish(self):
        """-> number of bucket all-reduces of this step; returns when the current stream may read the reduced G."""
        for b in sorted(set(self.at_finish)):
            self._launch(b)
        n = len(self.pending)
        for w in self.pending:
            w.wait()
        self.pending = []
        self.last_order, self.order = self.order, []
        return n


def broadcast_parameters(module, src=0):
    """Every replica starts from rank ``src``'s parameters and buffers (one broadcast of the flat buffer when the module has
    been flattened)."""
    if world() == 1:
        return
    fs = getattr(module, "_dsc_flat", None)
    with torch.no_grad():
        if fs is not None and fs.valid():
            dist.broadcast(fs.P, src=src)
            done = {id(p) for p in fs.params}
        else:
            done = set()
        for p in module.parameters():
            if id(p) not in done:
                dist.broadcast(p.data, src=src)
        for b in module.buffers():
            dist.broadcast(b, src=src)


def measure_allreduce(module, reps=5):
    """The gradient exchange alone (no compute to hide behind): all buckets of G all-reduced back to back, per step."""
    fs = getattr(module, "_dsc_flat", None)
    if fs is None or world() == 1:
        return None
    buckets = fs.buckets(N_BUCKETS)
    g = torch.zeros_like(fs.G)

    def once():
        works = [dist.all_reduce(g[s:e], op=dist.ReduceOp.SUM, async_op=True) for s, e in buckets]
        for w in works:
            w.wait()
    once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = fs.numel * 4
    return {"backend": dist.get_backend(), "world": world(), "buckets": len(buckets), "bytes": nbytes,
            "ms_per_step": round(ms, 3), "algbw_GBps": round(nbytes / ms / 1e6, 1),
            "busbw_GBps": round(nbytes / ms / 1e6 * 2 * (world() - 1) / world(), 1)}
