#!/usr/bin/env python
"""Order-unbiased A/B timing of tools/gemm_tune.hip variants: the GPU takes milliseconds to ramp its clocks after an idle
sync, so variants are timed round-robin (rotating order) after a long warm-up and the median of the rounds is reported.
    python tools/gemm_tune.py --build && python tools/gemm_ab.py 23,25 [--gn 0,1] [--k 512,1024] [--nores] [--stamps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from diffuscene_amd import _lib, ops  # noqa: E402

lib = C.CDLL(os.path.join(ROOT, "tools", "libgemm_tune.so"))
lib.tune_launch.argtypes = [C.c_int, C.c_int, C.POINTER(_lib.GemmArgs), C.c_void_p]
lib.tune_name.restype = C.c_char_p
lib.tune_read_timing.argtypes = [C.c_void_p, C.c_int]


def opt(name, default):
    for i, a in enumerate(sys.argv):
        if a == name:
            return sys.argv[i + 1]
    return default


variants = [int(x) for x in sys.argv[1].split(",")]
gns = [int(x) for x in opt("--gn", "0,1").split(",")]
ks = [int(x) for x in opt("--k", "512,1024").split(",")]
B, N = int(opt("--batch", "256")), int(opt("--objects", "80"))
nores = "--nores" in sys.argv
if opt("--stagger", None) is not None:
    lib.tune_set_stagger(int(opt("--stagger", "0")))
    print("stagger: the second resident block of every CU starts %s cycles late" % opt("--stagger", "0"))
dev = torch.device("cuda:0")
M = B * N
torch.manual_seed(0)
for K in ks:
    a = torch.randn(M, 512, device=dev)
    a2 = torch.randn(M, 512, device=dev) if K == 1024 else None
    w = torch.randn(512, K, device=dev) * 0.05
    b = torch.randn(512, device=dev)
    gamma, beta = torch.rand(512, device=dev) + 0.5, torch.randn(512, device=dev) * 0.1
    ss = torch.randn(B, 1024, device=dev) * 0.1
    r = None if nores else torch.randn(M, 512, device=dev)
    for gn in gns:
        ys = {v: torch.zeros(M, 512, device=dev) for v in variants}
        gs = {}
        for v in variants:
            if gn:
                gs[v] = ops.make_gemm_args(a, w, ys[v], b, a2, r, gamma=gamma, beta=beta, tokens_per_scene=N, scale_shift=ss, ss_mode=2)
            else:
                gs[v] = ops.make_gemm_args(a, w, ys[v], b, a2, r)
        s = ops.stream_ptr()
        ok = [v for v in variants if lib.tune_launch(v, gn, C.byref(gs[v]), s) == 0]
        torch.cuda.synchronize()
        for _ in range(150):                                   # clock ramp
            lib.tune_launch(ok[0], gn, C.byref(gs[ok[0]]), s)
        times = {v: [] for v in ok}
        for rnd in range(9):
            order = ok[rnd % len(ok):] + ok[:rnd % len(ok)]
            evs = []
            for v in order:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    lib.tune_launch(v, gn, C.byref(gs[v]), s)
                e1.record()
                evs.append((v, e0, e1))
            torch.cuda.synchronize()
            for v, e0, e1 in evs:
                times[v].append(e0.elapsed_time(e1) * 100.0)
        ref = ys[ok[0]]
        for v in ok:
            us = float(np.median(times[v]))
            err = float((ys[v] - ref).abs().max() / ref.abs().max())
            print("K=%4d gn=%d  %-60s %7.1f us [%6.1f..%6.1f] %6.1f TF (%.3f)  err=%.1e" % (
                K, gn, lib.tune_name(v).decode()[:60], us, min(times[v]), max(times[v]), 2.0 * M * 512 * K / us / 1e6,
                2.0 * M * 512 * K / us / 1e6 / 157.3, err), flush=True)
        if "--stamps" in sys.argv:
            for v in ok:
                for _ in range(3):
                    lib.tune_launch(v, gn, C.byref(gs[v]), s)
                torch.cuda.synchronize()
                nb = 512
                buf = np.zeros((nb, 8), dtype=np.int64)
                lib.tune_read_timing(buf.ctypes.data, nb)
                st = buf[:, :5] - buf[:, 0:1]
                ph = [("prologue", st[:, 1]), ("main", st[:, 2] - st[:, 1]), ("stats", (st[:, 3] - st[:, 2]) if gn else st[:, 2] * 0),
                      ("store", st[:, 4] - (st[:, 3] if gn else st[:, 2])), ("total", st[:, 4])]
                if gn:
                    ph += [("partials", buf[:, 5] - buf[:, 2]), ("bar1", buf[:, 6] - buf[:, 5]), ("reduce+bar2", buf[:, 7] - buf[:, 6]),
                           ("rowtab+bar3", buf[:, 3] - buf[:, 7])]
                print("   stamps v=%d: " % v + "  ".join("%s %.0f" % (nm, c[c > 0].mean() if (c > 0).any() else 0) for nm, c in ph))
