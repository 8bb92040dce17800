"""Input-pipeline throughput: device batch encoder vs the numpy port of the reference's decorator chain.
Usage (GPU box):  python tools/bench_dataset.py [--scenes 4096] [--batch 256] [--max-length 21]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dataset_ref as DR  # noqa: E402  (bench baseline leg only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--max-length", type=int, default=21)
    ap.add_argument("--epochs", type=int, default=5)
    a = ap.parse_args()
    from diffuscene_amd.datasets import CachedThreedFront, dataset_encoding_factory
    enc = "cached_diffusion_cosin_angle_objfeatsnorm_lat32_wocm"
    with tempfile.TemporaryDirectory() as tmp:
        ids = DR.write_synth_cached_dataset(tmp, a.scenes, seed=0, max_length=a.max_length)
        cfg = {"train_stats": "dataset_stats.txt", "room_layout_size": "64,64", "max_length": a.max_length}
        t0 = time.perf_counter()
        raw = CachedThreedFront(tmp, config=cfg, scene_ids=set(ids))
        load_s = time.perf_counter() - t0
    ds = dataset_encoding_factory(enc, raw, ["fixed_rotations"], None)
    loader = ds.loader(a.batch, shuffle=True, device="cuda:0")
    for _ in loader:            # warm-up epoch (uploads the store)
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nb = 0
    for _ in range(a.epochs):
        for s in loader:
            nb += 1
    torch.cuda.synchronize()
    dev_s = time.perf_counter() - t0
    # kernel-only time of one batch with the draws already made
    idx = list(range(a.batch))
    draws = ds.draw(idx)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ds.encode(idx, "cuda:0", draws)
    e0.record()
    for _ in range(50):
        ds.encode(idx, "cuda:0", draws)
    e1.record()
    torch.cuda.synchronize()
    enc_ms = e0.elapsed_time(e1) / 50
    # numpy port of the reference chain, one process
    rooms = [DR.synth_scene(i, 0, max_length=a.max_length) for i in range(a.batch)]
    st = DR.synth_stats()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        DR.encode_batch(rooms, st, a.max_length, augmentations=("fixed_rotations",))
    cpu_s = (time.perf_counter() - t0) / reps
    print(json.dumps({
        "metric": "training batches/s from the input pipeline (B=%d, N=%d)" % (a.batch, a.max_length),
        "value": round(nb / dev_s, 1), "ms_per_batch_incl_host_draws_and_h2d": round(1e3 * dev_s / nb, 3),
        "ms_per_batch_encode_only": round(enc_ms, 4),
        "cpu_numpy_port_batches_per_s_1proc": round(1.0 / cpu_s, 2), "cpu_ms_per_batch": round(1e3 * cpu_s, 2),
        "store_load_s": round(load_s, 2), "scenes": a.scenes}))


if __name__ == "__main__":
    main()
