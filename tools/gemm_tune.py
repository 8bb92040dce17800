#!/usr/bin/env python
"""Time the variants of tools/gemm_tune.hip on the GPU (build the .so locally first: `python tools/gemm_tune.py --build`)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "libgemm_tune.so")


def build():
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           os.path.join(ROOT, "tools", "gemm_tune.hip"), "-o", SO])


def main():
    if "--build" in sys.argv:
        build()
        return
    import torch
    from diffuscene_amd import _lib, ops
    lib = C.CDLL(SO)
    lib.tune_launch.argtypes = [C.c_int, C.c_int, C.POINTER(_lib.GemmArgs), C.c_void_p]
    lib.tune_name.restype = C.c_char_p
    dev = torch.device("cuda:0")
    only = None
    for a_ in sys.argv[1:]:
        if a_.startswith("--only="):
            only = [int(x) for x in a_[7:].split(",")]
    nores = "--nores" in sys.argv
    stag = 0
    for a_ in sys.argv[1:]:
        if a_.startswith("--stagger="):
            stag = int(a_[10:])
    if stag:
        lib.tune_set_stagger(stag)
        print("stagger: second resident block starts %d cycles late" % stag)
    NV = 27
    B, N = 256, 80
    M = B * N
    torch.manual_seed(0)
    res = {}
    for K in (512, 1024):
        a = torch.randn(M, 512, device=dev)
        a2 = torch.randn(M, 512, device=dev) if K == 1024 else None
        w = torch.randn(512, K, device=dev) * 0.05
        b = torch.randn(512, device=dev)
        gamma, beta = torch.rand(512, device=dev) + 0.5, torch.randn(512, device=dev) * 0.1
        ss = torch.randn(B, 1024, device=dev) * 0.1
        r = torch.randn(M, 512, device=dev)
        for gn in (0, 1):
            ref = None
            for v in (only if only is not None else range(NV)):
                y = torch.zeros(M, 512, device=dev)
                if gn and 18 <= v <= 22:
                    continue                     # small tiles cannot hold an 80-token scene
                if gn and v in (7, 24, 26):
                    continue                     # 8-wave BK64 tile is a plain-GEMM tile
                rr = None if nores else r
                if gn:
                    g = ops.make_gemm_args(a, w, y, b, a2, rr, gamma=gamma, beta=beta, tokens_per_scene=N,
                                           scale_shift=ss, ss_mode=2)
                else:
                    g = ops.make_gemm_args(a, w, y, b, a2, rr)
                s = ops.stream_ptr()
                rc = lib.tune_launch(v, gn, C.byref(g), s)
                torch.cuda.synchronize()
                if rc != 0:
                    print("variant", v, "launch error", rc)
                    continue
                if ref is None:
                    ref = y.clone()
                err = float((y - ref).abs().max() / ref.abs().max())
                for _ in range(3):
                    lib.tune_launch(v, gn, C.byref(g), s)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 20
                e0.record()
                for _ in range(reps):
                    lib.tune_launch(v, gn, C.byref(g), s)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / reps
                tf = 2.0 * M * 512 * K / us / 1e6
                res[(K, gn, v)] = us
                print("K=%4d gn=%d  %-58s %8.1f us  %6.1f TF  (%.0f%% of 157.3)  err=%.1e" % (
                    K, gn, lib.tune_name(v).decode(), us, tf, 100 * tf / 157.3, err), flush=True)
    # phase timeline of the baseline and the best 8-wave variant (shader-clock stamps per block)
    import numpy as np
    lib.tune_read_timing.argtypes = [C.c_void_p, C.c_int]
    for gn in (0, 1):
        for v in ((25, 23) if only is not None else (1, 13, 14)):
            a = torch.randn(M, 512, device=dev); w = torch.randn(512, 512, device=dev) * 0.05
            y = torch.zeros(M, 512, device=dev)
            rr = None if nores else r
            if gn:
                g = ops.make_gemm_args(a, w, y, b, None, rr, gamma=gamma, beta=beta, tokens_per_scene=N, scale_shift=ss, ss_mode=2)
            else:
                g = ops.make_gemm_args(a, w, y, b, None, rr)
            for _ in range(3):
                lib.tune_launch(v, gn, C.byref(g), ops.stream_ptr())
            torch.cuda.synchronize()
            nb = 256 if v == 14 else 512
            buf = np.zeros((nb, 8), dtype=np.int64)
            lib.tune_read_timing(buf.ctypes.data, nb)
            if False:
                hw, xcc = buf[:, 5], buf[:, 6]
                print("hw_id / xcc_id of blocks 0..23 and 256..263:")
                for bi in list(range(24)) + list(range(256, 264)):
                    print("   blk %3d hw=0x%08x xcc=0x%x" % (bi, hw[bi] & 0xffffffff, xcc[bi] & 0xffffffff))
                key = (xcc & 0xf) * 1000000 + (hw & 0xffffff00)
                import collections
                cnt = collections.Counter(key.tolist())
                print("distinct (xcc, hw_id[31:8]) slots:", len(cnt), " blocks per slot histogram:", collections.Counter(cnt.values()))
                pairs = collections.defaultdict(list)
                for bi in range(nb):
                    pairs[int(key[bi])].append(bi)
                print("sample co-resident block pairs:", [v_ for v_ in list(pairs.values())[:12]])
            t0 = buf[:, 0].min()
            st = buf[:, :5] - t0
            print("timeline gn=%d v=%d (cycles @ shader clock; per-block mean [min..max])" % (gn, v))
            for nm, col in (("start skew", st[:, 0]), ("prologue", st[:, 1] - st[:, 0]), ("main loop", st[:, 2] - st[:, 1]),
                            ("gn stats", (st[:, 3] - st[:, 2]) if gn else st[:, 2] * 0), ("store phase", st[:, 4] - (st[:, 3] if gn else st[:, 2])),
                            ("end time", st[:, 4])):
                print("   %-12s %9.0f [%9.0f .. %9.0f]" % (nm, col.mean(), col.min(), col.max()))
    print("\nfit T = a + b*K (us):")
    for gn in (0, 1):
        for v in range(NV):
            if (512, gn, v) in res and (1024, gn, v) in res:
                t1, t2 = res[(512, gn, v)], res[(1024, gn, v)]
                bb = (t2 - t1) / 512
                print("gn=%d v=%2d  fixed %.1f us, %.2f us per 32-K tile (ideal 4.27)" % (gn, v, t1 - bb * 512, bb * 32))


if __name__ == "__main__":
    main()
