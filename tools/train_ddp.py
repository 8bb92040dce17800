#!/usr/bin/env python
"""Data-parallel launcher for the UNCHANGED reference training script (scripts/train_diffusion.py), one process per GPU.

    python tools/train_ddp.py --reference /path/to/DiffuScene --gpus 8 -- <config.yaml> <output_dir> [script options]

The reference is single-device (train_diffusion.py:86-89 hard-codes cuda:0, :150-156 builds a plain shuffling DataLoader).
This wrapper leaves the script body alone and arranges the process around it:

* started from a plain shell it re-executes itself under ``torch.distributed.run`` (127.0.0.1, N ranks);
* every rank exposes only ITS GPU (HIP_VISIBLE_DEVICES = LOCAL_RANK, set before torch touches HIP), so the script's
  ``cuda:0`` is the local device, and joins the RCCL process group (``ddp.init_from_env``);
* ``scene_synthesis.networks`` / ``.stats_logger`` resolve to diffuscene_amd (``install_as_scene_synthesis``); with a live process
  group ``train_on_batch`` broadcasts rank 0's weights before the first step and all-reduces the flat gradient buffer over
  RCCL in buckets while the backward is still running (train_step.py, ddp.FlatGradientReducer);
* the script's ``DataLoader(..., shuffle=True)`` is given a ``DistributedSampler`` (same seed on every rank, a new epoch order on
  every pass) so the ranks draw disjoint shards -- the per-GPU batch stays the YAML's ``batch_size`` (weak scaling);
* ranks > 0 write their checkpoints / stats into a scratch directory (rank 0 keeps the requested output directory).

Nothing of the reference tree is modified or copied; ``--script`` selects another entry point with the same conventions."""
import argparse
import os
import runpy
import socket
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn(n, argv):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def shard_dataloaders(rank, world, seed):
    """torch.utils.data.DataLoader(..., shuffle=True) -> DistributedSampler over the ranks (a new permutation on every pass)."""
    import torch.utils.data as tud
    base = tud.DataLoader

    class _EpochSampler(tud.distributed.DistributedSampler):
        def __iter__(self):
            it = super().__iter__()
            self.set_epoch(self.epoch + 1)
            return it

    class ShardedDataLoader(base):
        def __init__(self, dataset, *args, **kw):
            if kw.get("shuffle") and kw.get("sampler") is None and kw.get("batch_sampler") is None and world > 1:
                kw["shuffle"] = False
                kw["sampler"] = _EpochSampler(dataset, num_replicas=world, rank=rank, shuffle=True, seed=seed, drop_last=False)
            super().__init__(dataset, *args, **kw)

    tud.DataLoader = ShardedDataLoader
    import torch.utils.data.dataloader as dl
    dl.DataLoader = ShardedDataLoader


def worker(args, script_argv):
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # the script says cuda:0: make that the local GPU (must happen before torch initialises HIP)
    os.environ["HIP_VISIBLE_DEVICES"] = os.environ.get("DSC_VISIBLE_DEVICE", str(local))
    os.environ["LOCAL_RANK"] = "0"                          # init_from_env picks cuda:LOCAL_RANK of the VISIBLE devices
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from diffuscene_amd import ddp, install_as_scene_synthesis
    ddp.init_from_env()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    install_as_scene_synthesis(reference_root=args.reference)
    seed = 27
    if "--seed" in script_argv:
        seed = int(script_argv[script_argv.index("--seed") + 1])
    shard_dataloaders(rank, world, seed)
    if rank > 0 and len(script_argv) >= 2:                  # positional: config_file output_directory
        pos = [i for i, a in enumerate(script_argv) if not a.startswith("-") and (i == 0 or not script_argv[i - 1].startswith("--"))]
        if len(pos) >= 2:
            script_argv[pos[1]] = tempfile.mkdtemp(prefix="dsc_rank%d_" % rank)
    script = args.script or os.path.join(args.reference, "scripts", "train_diffusion.py")
    sys.path.insert(0, os.path.dirname(script))
    sys.path.insert(0, args.reference)
    sys.argv = [script] + script_argv
    try:
        runpy.run_path(script, run_name="__main__")
    finally:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", required=True, help="DiffuScene checkout (its scripts/ and scene_synthesis/datasets are used)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--script", default=None, help="entry point to run per rank (default: <reference>/scripts/train_diffusion.py)")
    if "--" in sys.argv:
        i = sys.argv.index("--")
        own, script_argv = sys.argv[1:i], sys.argv[i + 1:]
    else:
        own, script_argv = sys.argv[1:], []
    args = ap.parse_args(own)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn(args.gpus, sys.argv[1:]))
    worker(args, script_argv)


if __name__ == "__main__":
    main()
