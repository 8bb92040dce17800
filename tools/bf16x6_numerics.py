#!/usr/bin/env python
"""CPU study behind tools/gemm_bf16x6.hip: error of an f32 GEMM emulated with bf16 pieces (round-to-nearest split, exact
products, f32 accumulation) against an f64 product, next to a plain f32 GEMM.  Runs anywhere (torch CPU).
    python tools/bf16x6_numerics.py > profiles/r02_bf16x6_numerics.txt"""
import torch

torch.manual_seed(0)


def split(x, n, dt=torch.bfloat16):
    parts, r = [], x.clone()
    for _ in range(n):
        p = r.to(dt).float()
        parts.append(p)
        r = r - p
    return parts


def main():
    print("# rows: scheme, max|err|/max|ref|, max element-wise relative error (|ref| floored at 1e-3), rms err / rms ref")
    for (M, K, N) in [(2048, 512, 512), (2048, 1024, 512), (2048, 128, 512)]:
        A = torch.nn.functional.silu(torch.randn(M, K)) * 1.3
        W = torch.randn(K, N) / K ** 0.5
        ref = A.double() @ W.double()

        def err(y):
            d = (y.double() - ref).abs()
            return "%.2e  %.2e  %.2e" % ((d.max() / ref.abs().max()).item(), (d / ref.abs().clamp_min(1e-3)).max().item(),
                                         (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())

        print("M=%d K=%d N=%d" % (M, K, N))
        print("  f32 GEMM                               ", err(A @ W))
        a, w = split(A, 3), split(W, 3)
        exact = bool(torch.equal(a[0] + a[1] + a[2], A)) and bool(torch.equal(w[0] + w[1] + w[2], W))
        print("  3-piece split reconstructs the f32 operands exactly:", exact)
        for name, terms in [("bf16 x1 (1,1)", [(0, 0)]),
                            ("bf16 x3 (1,1)(1,2)(2,1)", [(0, 0), (0, 1), (1, 0)]),
                            ("bf16 x6 +(2,2)(1,3)(3,1)", [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]),
                            ("bf16 x9 all", [(i, j) for i in range(3) for j in range(3)])]:
            y = torch.zeros(M, N)
            for (i, j) in sorted(terms, key=lambda t: -(t[0] + t[1])):      # small terms first, as the kernel does
                y = y + a[i] @ w[j]
            print("  %-38s " % name, err(y))
        # f16 pieces (11-bit mantissas: two pieces = 22 bits) need only three products but are range-bound: the second piece of a
        # small operand is an f16 subnormal, so the error depends on the operands' scale -- not pursued
        for scale in (1.0, 1e-3):
            a, w = split(A * scale, 2, torch.float16), split(W, 2, torch.float16)
            y = (a[0] @ w[1] + a[1] @ w[0] + a[0] @ w[0]) / scale
            print("  %-38s " % ("f16 x3, activations scaled by %g" % scale), err(y))


if __name__ == "__main__":
    main()
