#!/usr/bin/env python
"""Probe: do two independent half-batch denoiser chains on two HIP streams overlap (one chain's GEMM epilogue under the
other's main loop)?  Eager launches, no graph."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import argparse

args = argparse.Namespace(batch=256, objects=80)
dev = torch.device("cuda:0")
model, cfg = bench.build_model(args, dev)
net = model.diffusion.model
eng = net.engine(dev)
N = 80
with torch.no_grad():
    full = eng.prepare(256, N, model._instance_condition(256, dev), None, time_table=True)
    h0 = eng.prepare(128, N, model._instance_condition(128, dev), None, time_table=True, slot=0)
    h1 = eng.prepare(128, N, model._instance_condition(128, dev), None, time_table=True, slot=1)
    for p in (full, h0, h1):
        p.x_in.normal_(); p.t_in.fill_(500)

    def t(fn, n=10):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    s1 = torch.cuda.Stream()
    def seq_full(): full.run()
    def seq_halves(): h0.run(); h1.run()
    def par_halves():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        h0.run()
        with torch.cuda.stream(s1):
            h1.run()
        cur.wait_stream(s1)
    print("full batch, one chain      : %.3f ms" % t(seq_full))
    print("two half chains, one stream: %.3f ms" % t(seq_halves))
    print("two half chains, 2 streams : %.3f ms" % t(par_halves))
