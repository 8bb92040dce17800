"""CPU, world_size 2, gloo: the data-parallel gradient step (ddp.py) -- bucketed all-reduce mean, shard_batch and the
fused clip -- reproduces the single-process gradient on the concatenated batch."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.SiLU(), torch.nn.Linear(64, 64), torch.nn.SiLU(),
                               torch.nn.Linear(64, 8))


def _batch():
    g = torch.Generator().manual_seed(1)
    return {"x": torch.randn(12, 16, generator=g), "y": torch.randn(12, 8, generator=g), "description": list("abcdefghijkl")}


def _loss(m, b):
    return ((m(b["x"]) - b["y"]) ** 2).mean()


def _worker(rank, ws, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from diffuscene_amd import ddp
    m = _model()
    shard = ddp.shard_batch(_batch())
    assert shard["x"].shape[0] == 6 and len(shard["description"]) == 6
    _loss(m, shard).backward()
    nb = ddp.average_gradients(m, bucket_bytes=4096)          # tiny buckets: exercises the multi-bucket path
    assert nb >= 3
    total = ddp.clip_grad_norm_fused(m.parameters(), 0.05)
    flat = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(ws)]
    dist.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1])               # every rank clips / steps identically
    # overlapped form: bucket all-reduces launched from gradient hooks during backward give the same averaged gradients
    m2 = _model()
    red = ddp.overlapped_reducer(m2, bucket_bytes=4096)
    assert red is ddp.overlapped_reducer(m2) and len(red.buckets) >= 3
    for _ in range(2):                                          # two steps: state resets between steps
        m2.zero_grad(set_to_none=True)
        _loss(m2, shard).backward()
        assert red.launched_during_backward >= 3                # complete buckets left before backward ended
        assert red.finish() == len(red.buckets)
    flat2 = torch.cat([p.grad.reshape(-1) for p in m2.parameters()])
    ddp.clip_grad_norm_fused(m2.parameters(), 0.05)
    flat2 = torch.cat([p.grad.reshape(-1) for p in m2.parameters()])
    assert torch.allclose(flat2, flat, rtol=1e-6, atol=1e-9)
    # a parameter without gradient (frozen branch) must not dead-lock the bucket schedule
    m3 = _model()
    m3[4].weight.requires_grad_(False)
    red3 = ddp.OverlappedGradientReducer(m3, bucket_bytes=4096)
    _loss(m3, shard).backward()
    red3.finish()
    if rank == 0:
        torch.save({"grad": flat, "norm": total}, out)
    dist.destroy_process_group()


def test_ddp_mean_gradients_match_single_process(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    from diffuscene_amd import ddp
    m = _model()
    _loss(m, _batch()).backward()
    ref_norm = torch.nn.utils.clip_grad_norm_(m.parameters(), 0.05)
    ref = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    assert torch.allclose(got["norm"], ref_norm, rtol=1e-5)
    assert torch.allclose(got["grad"], ref, rtol=1e-5, atol=1e-8)
    assert ddp.world() == 1 and ddp.average_gradients(m) == 0   # no-op when not distributed
