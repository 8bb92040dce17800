#!/usr/bin/env python
"""Sweep the number of half-batch chains per captured reverse step and their phase offset (DSC_CHAINS, DSC_CHAIN_OFFSET_NS).
    python tools/chain_sweep.py [config] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "living80"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
spec = dict(bench.CONFIGS[name])
dev = torch.device("cuda:0")
model, _ = bench.build_model(spec, dev)
cases = [(1, 0)] + [(2, o) for o in (0, 15000, 30000, 45000, 60000, 80000, 110000)] + [(4, 30000), (1, 0)]
for nch, off in cases:
    os.environ["DSC_CHAINS"] = str(nch)
    os.environ["DSC_CHAIN_OFFSET_NS"] = str(off)
    model.diffusion.diffusion._graphs = {}
    sr = bench.SampleRunner(spec, model, dev, seed=0)
    sr.run(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sr.run(steps)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print("chains=%d offset=%6d ns : %.3f ms/step  %.2f steps/s" % (nch, off, ms, 1e3 / ms), flush=True)
    del sr
