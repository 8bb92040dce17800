#!/usr/bin/env python
"""Sustained launch time of the product GEMMs, split-bf16 path next to the exact-f32 MFMA path, same operands.

    python tools/gemm_split_bench.py [--batch 256] [--objects 80]

Every kernel runs alone for ~0.2 s from a captured graph of 50 launches (no host gaps; the power management averages over
milliseconds, so a short burst between other kernels can run above the sustained clock).  Shapes: the launches of the denoiser at
the given batch (plain 1x1 convs, the two-segment skip-connection form, the fused Block = WS-conv + GroupNorm + SiLU, K = 512 / 1024).
TF figures are ALGORITHMIC f32 flops (2 M n K); the split path executes 6 bf16 MFMA products per f32 product.
The experiment generations this kernel came from (pipelines 0-3, 1- / 3- / 6-product variants, attribution probes, the 4-wave /
weights-from-L2 second generation) are in the git history (tools/gemm_bf16x6*.hip, removed after round 3) and their measurements
under profiles/r03_bf16x6_*.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def opt(name, default):
    for i, a in enumerate(sys.argv):
        if a == name:
            return sys.argv[i + 1]
    return default


def main():
    import torch
    from diffuscene_amd import ops
    dev = torch.device("cuda:0")
    B, N = int(opt("--batch", "256")), int(opt("--objects", "80"))
    M = B * N
    torch.manual_seed(0)

    def sustained(fn):
        g = torch.cuda.CUDAGraph()
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(50):
                fn()
        for _ in range(30):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1000.0 / 2000

    print("# M = %d tokens (B = %d scenes x N = %d); us per launch, sustained" % (M, B, N))
    print("%-44s %10s %10s %8s %12s %12s" % ("launch", "f32 MFMA", "split-bf16", "ratio", "f32 TF", "split TF"))
    for name, n, k1, k2, gn, res in (("conv 512 -> 512", 512, 512, 0, False, False), ("conv 512 -> 1024", 1024, 512, 0, False, False),
                                     ("conv 512 -> 384 (qkv)", 384, 512, 0, False, False), ("conv [512|512] -> 512", 512, 512, 512, False, False),
                                     ("Block 512 -> 512", 512, 512, 0, True, False), ("Block 512 -> 512 + residual", 512, 512, 0, True, True),
                                     ("Block [512|512] -> 512", 512, 512, 512, True, False)):
        K = k1 + k2
        a1 = torch.nn.functional.silu(torch.randn(M, k1, device=dev)) * 1.3
        a2 = torch.nn.functional.silu(torch.randn(M, k2, device=dev)) * 1.3 if k2 else None
        w = torch.randn(n, K, device=dev) / K ** 0.5
        b = torch.randn(n, device=dev) * 0.1
        (pl,) = ops.split_planes([(w, None, False)])
        y = torch.empty(M, n, device=dev)
        r = torch.randn(M, n, device=dev) if res else None
        if gn:
            gamma, beta = torch.rand(n, device=dev) + 0.5, torch.randn(n, device=dev) * 0.1
            ss = torch.randn(B, 2 * n, device=dev) * 0.1
            args = [ops.make_gemm_args(a1, w, y, b, a2, r, gamma=gamma, beta=beta, tokens_per_scene=N, scale_shift=ss, ss_mode=2,
                                       w_planes=p) for p in (None, pl)]
        else:
            args = [ops.make_gemm_args(a1, w, y, b, a2, r, w_planes=p) for p in (None, pl)]
        t32, tsp = (sustained(lambda g=g: ops.run_gemm(g, gn=gn)) for g in args)
        fl = 2.0 * M * n * K
        print("%-44s %10.1f %10.1f %8.2f %12.1f %12.1f" % (name, t32, tsp, t32 / tsp, fl / t32 / 1e6, fl / tsp / 1e6), flush=True)


if __name__ == "__main__":
    main()
