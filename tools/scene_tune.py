#!/usr/bin/env python
"""Stage-1 check of the scene-resident layer kernel (csrc/scene_core.h) against the product GEMM: same inputs, error and
time per launch.  Build first: python tools/gemm_tune.py --build"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "libgemm_tune.so")


def main():
    import torch
    from diffuscene_amd import _lib, ops
    lib = C.CDLL(SO)
    lib.tune_scene_launch.argtypes = [C.c_int, C.POINTER(_lib.GemmArgs), C.c_void_p]
    lib.tune_scene_direct_launch.argtypes = [C.c_int, C.POINTER(_lib.GemmArgs), C.c_void_p]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for (B, N) in ((256, 80),):
        M = B * N
        for K in (512, 1024):
            a = torch.randn(M, 512, device=dev)
            a2 = torch.randn(M, 512, device=dev) if K == 1024 else None
            w = torch.randn(512, K, device=dev) * 0.05
            b = torch.randn(512, device=dev)
            gamma, beta = torch.rand(512, device=dev) + 0.5, torch.randn(512, device=dev) * 0.1
            ss = torch.randn(B, 1024, device=dev) * 0.1
            r = torch.randn(M, 512, device=dev)
            for gn in (0, 1):
                y0 = torch.zeros(M, 512, device=dev)
                y1 = torch.zeros(M, 512, device=dev)
                kw = dict(gamma=gamma, beta=beta, tokens_per_scene=N, scale_shift=ss, ss_mode=2) if gn else {}
                g0 = ops.make_gemm_args(a, w, y0, b, a2, r, **kw)
                g1 = ops.make_gemm_args(a, w, y1, b, a2, r, **kw)
                g1.tokens_per_scene = N
                s = ops.stream_ptr()
                fn = _lib.fn("dsc_gemm_gn_silu_f32" if gn else "dsc_gemm_f32")
                _lib.check(fn(C.byref(g0), s), "product")
                rc = lib.tune_scene_launch(gn, C.byref(g1), s)
                torch.cuda.synchronize()
                assert rc == 0, rc
                err = float((y1 - y0).abs().max() / y0.abs().max())
                y2 = torch.zeros(M, 512, device=dev)
                g2 = ops.make_gemm_args(a, w, y2, b, a2, r, **kw)
                g2.tokens_per_scene = N
                rc = lib.tune_scene_direct_launch(gn, C.byref(g2), s)
                torch.cuda.synchronize()
                assert rc == 0, rc
                err2 = float((y2 - y0).abs().max() / y0.abs().max())

                def timeit(f):
                    for _ in range(3):
                        f()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        f()
                    e1.record()
                    torch.cuda.synchronize()
                    return e0.elapsed_time(e1) * 1e3 / 20
                t0 = timeit(lambda: fn(C.byref(g0), s))
                t1 = timeit(lambda: lib.tune_scene_launch(gn, C.byref(g1), s))
                t2 = timeit(lambda: lib.tune_scene_direct_launch(gn, C.byref(g2), s))
                y4 = torch.zeros(M, 512, device=dev)
                g4 = ops.make_gemm_args(a, w, y4, b, a2, r, **kw)
                g4.tokens_per_scene = N
                lib.tune_scene_g2_launch.argtypes = [C.c_int, C.POINTER(_lib.GemmArgs), C.c_void_p]
                rc = lib.tune_scene_g2_launch(gn, C.byref(g4), s)
                torch.cuda.synchronize()
                assert rc == 0, rc
                err4 = float((y4 - y0).abs().max() / y0.abs().max())
                t4 = timeit(lambda: lib.tune_scene_g2_launch(gn, C.byref(g4), s))
                print("      two 4-wave groups, LDS-counter barriers: %7.1f us (%5.1f TF) err %.1e" % (t4, 2.0 * M * 512 * K / 1e6 / t4, err4))
                # fragment-major operands (N == 80 only): swizzle on the host side of the test
                if N == 80:
                    def frag_x(x):          # [M, Kc] -> [scene][tt][kg][lg][li][4]
                        Kc = x.shape[1]
                        return x.view(B, 5, 16, Kc // 16, 4, 4).permute(0, 1, 3, 4, 2, 5).contiguous()
                    def frag_w(wm_):        # [n, Kt] -> [ct][kg][lg][li][4]
                        n_, Kt = wm_.shape
                        return wm_.view(n_ // 16, 16, Kt // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous()
                    def unfrag_y(yf, n_):   # [scene][tt][kgc][lg][li][4] -> [M, n]
                        return yf.view(B, 5, n_ // 16, 4, 16, 4).permute(0, 1, 4, 2, 3, 5).reshape(M, n_)
                    xa, xb = frag_x(a), (frag_x(a2) if a2 is not None else None)
                    wf_, rf = frag_w(w), frag_x(r)
                    y3 = torch.zeros(M * 512, device=dev)
                    g3 = ops.make_gemm_args(a, w, y0.clone(), b, a2, r, **kw)
                    g3.tokens_per_scene = N
                    g3.a1 = xa.data_ptr(); g3.w = wf_.data_ptr(); g3.residual = rf.data_ptr(); g3.y = y3.data_ptr()
                    if xb is not None:
                        g3.a2 = xb.data_ptr()
                    lib.tune_scene_frag_launch.argtypes = [C.c_int, C.c_int, C.POINTER(_lib.GemmArgs), C.c_void_p]
                    rc = lib.tune_scene_frag_launch(gn, 0, C.byref(g3), s)
                    torch.cuda.synchronize()
                    assert rc == 0, rc
                    err3 = float((unfrag_y(y3, 512) - y0).abs().max() / y0.abs().max())
                    t3 = timeit(lambda: lib.tune_scene_frag_launch(gn, 0, C.byref(g3), s))
                    pr = [timeit(lambda q=q: lib.tune_scene_frag_launch(1, q, C.byref(g3), s)) for q in (1, 2, 3, 4)] if gn else [0, 0, 0, 0]
                    print("      fragment-major operands: %7.1f us (%5.1f TF) err %.1e   [probes: no loads %.1f, L1-resident %.1f, X only %.1f, W only %.1f us]"
                          % (t3, 2.0 * M * 512 * K / 1e6 / t3, err3, pr[0], pr[1], pr[2], pr[3]))
                if gn and N == 80:
                    lib.tune_scene_probe_launch.argtypes = [C.c_int, C.POINTER(_lib.GemmArgs), C.c_void_p]
                    tp1 = timeit(lambda: lib.tune_scene_probe_launch(1, C.byref(g2), s))
                    tp2 = timeit(lambda: lib.tune_scene_probe_launch(2, C.byref(g2), s))
                    print("      probes (direct, GN): no loads in loop %7.1f us   k not advancing (L1-resident operands) %7.1f us" % (tp1, tp2))
                tf = 2.0 * M * 512 * K / 1e6
                print("B=%3d N=%2d K=%4d gn=%d  product %7.1f us (%5.1f TF)  scene/LDS %7.1f us (%5.1f TF) err %.1e  scene/direct %7.1f us (%5.1f TF) err %.1e"
                      % (B, N, K, gn, t0, tf / t0, t1, tf / t1, err, t2, tf / t2, err2), flush=True)


if __name__ == "__main__":
    main()
