"""Data-parallel gradient step for the DDPM training path (new work: the reference is single-device and never
calls torch.distributed).

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for tests).
Scenes are independent, so the batch is sharded over ranks with no data-path collective; the ONLY exchange is
one averaged all-reduce of the 77.7 M fp32 gradients per step.  Gradients are packed into a few large
contiguous buckets (default 128 MiB: xGMI is point-to-point, per-link bandwidth-bound, so few large
messages beat many small ones) and the bucket all-reduces are issued asynchronously so bucket i+1 is being
packed while bucket i is on the wire; the global-norm clip then runs on the reduced gradients, identical on
every rank, with no extra collective.
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
