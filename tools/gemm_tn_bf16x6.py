#!/usr/bin/env python
"""EXPERIMENT harness for tools/gemm_tn_bf16x6.hip (f32-accurate weight-gradient GEMM on the bf16 matrix cores; not in the product).

    python tools/gemm_tn_bf16x6.py --build      # build container: hipcc -> tools/libgemm_tn_bf16x6.so (travels to the GPU box)
    python tools/gemm_tn_bf16x6.py              # GPU box: error vs an f64 product beside dsc_gemm_tn_f32, and timing

dw[n][k] = sum_m dy[m][n] x[m][k] at the training shapes (M = 20480 tokens, 512 x 512 and 512 x 1024).  The experiment kernel writes
one [n][k] slab per token slice (its parallelism; the product's grouped launch takes it from the ~57 layers of a backward), so its
time is the main loop's; the slab sum (torch) is timed separately.  TF figures are ALGORITHMIC f32 flops (2*M*n*k)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "libgemm_tn_bf16x6.so")


def build():
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           os.path.join(ROOT, "tools", "gemm_tn_bf16x6.hip"), "-o", SO])
    print("built", SO)


def main():
    if "--build" in sys.argv:
        return build()
    import numpy as np
    import torch
    from diffuscene_amd import ops

    lib = C.CDLL(SO)
    lib.bf16x6_gemm_tn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_void_p]
    dev = torch.device("cuda:0")
    M = 256 * 80
    torch.manual_seed(0)
    s = ops.stream_ptr()
    for (n, k) in ((512, 512), (512, 1024), (1024, 512)):
        x = torch.nn.functional.silu(torch.randn(M, k, device=dev)) * 1.3
        dy = torch.randn(M, n, device=dev) * 0.01
        tiles = (n // 256) * (k // 128)
        slices = max(1, 256 // tiles)
        while M % (32 * slices):
            slices //= 2
        rps = M // slices
        slabs = torch.zeros(slices, n, k, device=dev)

        def run(products):
            rc = lib.bf16x6_gemm_tn(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), slabs.data_ptr(), M, n, k, slices, rps,
                                    products, s)
            assert rc == 0, rc

        ref = dy.double().t() @ x.double()
        rms = float(ref.pow(2).mean().sqrt())

        def err(y):
            d = (y.double() - ref).abs()
            return float(d.max()) / rms, float(d.pow(2).mean().sqrt()) / rms

        # identity check: dy = one-hot rows -> dw[n] = sum of the x rows whose one-hot is n (asymmetric, catches operand swaps)
        dy1 = torch.zeros(M, n, device=dev)
        dy1[torch.arange(M, device=dev), torch.arange(M, device=dev) % n] = 1.0
        keep = dy
        dy = dy1
        run(6)
        torch.cuda.synchronize()
        want = torch.zeros(n, k, device=dev, dtype=torch.float64).index_add_(0, torch.arange(M, device=dev) % n, x.double())
        print("n=%d k=%d  identity check: max |diff| / rms = %.2e" % (n, k, float((slabs.sum(0).double() - want).abs().max()) / float(want.pow(2).mean().sqrt())),
              flush=True)
        dy = keep
        y_prod = ops.gemm_tn(x, dy)
        torch.cuda.synchronize()
        print("n=%d k=%d  dsc_gemm_tn_f32 (f32 MFMA)   max %.2e  rms %.2e" % ((n, k) + err(y_prod)), flush=True)
        for pr in (6, 1):
            run(pr)
            torch.cuda.synchronize()
            print("n=%d k=%d  bf16 split products=%d        max %.2e  rms %.2e   (%d slices of %d tokens)" % ((n, k, pr) + err(slabs.sum(0)) + (slices, rps)),
                  flush=True)
        names = ["prod", 6, 1, "slabsum"]
        fns = {"prod": lambda: ops.gemm_tn(x, dy), 6: lambda: run(6), 1: lambda: run(1), "slabsum": lambda: slabs.sum(0)}
        for _ in range(60):
            fns["prod"]()
        times = {v: [] for v in names}
        for rnd in range(8):
            order = names[rnd % 4:] + names[:rnd % 4]
            evs = []
            for v in order:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fns[v]()
                e1.record()
                evs.append((v, e0, e1))
            torch.cuda.synchronize()
            for v, e0, e1 in evs:
                times[v].append(e0.elapsed_time(e1) * 100.0)
        flops = 2.0 * M * n * k
        for v in names:
            us = float(np.median(times[v]))
            label = {"prod": "dsc_gemm_tn_f32", 6: "bf16 split x6 (slabs)", 1: "bf16 x1 (slabs, pipe ceiling)", "slabsum": "torch sum of the slabs"}[v]
            print("n=%4d k=%4d  %-30s %7.1f us [%6.1f..%6.1f]  %6.1f TF f32-equivalent (%.3f of the f32-MFMA peak)" % (
                n, k, label, us, min(times[v]), max(times[v]), flops / us / 1e6, flops / us / 1e6 / 157.3), flush=True)


if __name__ == "__main__":
    main()
