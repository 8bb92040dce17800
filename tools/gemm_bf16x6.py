#!/usr/bin/env python
"""EXPERIMENT harness for tools/gemm_bf16x6.hip (f32-accurate GEMM on the bf16 matrix cores; not part of the product).

    python tools/gemm_bf16x6.py --build            # build container: hipcc -> tools/libgemm_bf16x6.so (travels to the GPU box)
    python tools/gemm_bf16x6.py [--k 512,1024] [--noslp]   # GPU box: error vs an f64 product and order-unbiased timing

For every K: the error of each variant and of the product's dsc_gemm_f32 against torch's f64 matmul (max / rms, relative to
the rms of the result), an identity check with an asymmetric weight (catches operand / row-column swaps), and round-robin
timing after a clock warm-up (same method as tools/gemm_ab.py).  TF figures are ALGORITHMIC f32 flops (2*M*N*K)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "libgemm_bf16x6.so")
SO_NOSLP = os.path.join(ROOT, "tools", "libgemm_bf16x6_noslp.so")   # scalar f32 subtractions in the split instead of v_pk_add_f32


def build():
    variants = ((SO, [], "gemm_bf16x6.hip"), (SO_NOSLP, ["-fno-slp-vectorize"], "gemm_bf16x6.hip"),
                (SO.replace(".so", "_v5.so"), [], "gemm_bf16x6_v5.hip"))
    for so, extra, src in variants:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + extra +
                              [os.path.join(ROOT, "tools", src), "-o", so])
        print("built", so)


def opt(name, default):
    for i, a in enumerate(sys.argv):
        if a == name:
            return sys.argv[i + 1]
    return default


def main():
    if "--build" in sys.argv:
        return build()
    import numpy as np
    import torch
    from diffuscene_amd import ops

    lib = C.CDLL(opt("--so", SO_NOSLP if "--noslp" in sys.argv else SO))
    print("so:", opt("--so", "default"))
    print("library:", "no-SLP build (scalar subtractions)" if "--noslp" in sys.argv else "default build (v_pk_add_f32 in the split)")
    FM = hasattr(lib, "bf16x6_layout_fragment_major")       # second generation: fragment-major weight planes
    if FM:
        lib.bf16x6_split_planes_nk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    else:
        lib.bf16x6_split_planes.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]

    def split_into(wt, planes_t, s_):
        if FM:
            assert lib.bf16x6_split_planes_nk(wt.data_ptr(), wt.shape[0], wt.shape[1], planes_t.data_ptr(), s_) == 0
        else:
            assert lib.bf16x6_split_planes(wt.data_ptr(), wt.numel(), planes_t.data_ptr(), s_) == 0

    def planes_nk(planes_t, n_, k_):
        """planes as [3][n][k] whatever the storage order"""
        if not FM:
            return planes_t
        return planes_t.view(3, n_ // 16, k_ // 32, 4, 16, 8).permute(0, 1, 4, 2, 3, 5).reshape(3, n_, k_)
    lib.bf16x6_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,   # x x2 k1 lda planes bias out ldc
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,                                # m n k act residual ldr
                                  C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int,                        # gn gamma beta eps ss ld_ss
                                  C.c_int, C.c_void_p, C.c_void_p, C.c_int,                                               # ss_mode ss_index preact ld_pre
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]                                         # ntok products pipe tile stream

    def ptr(t):
        return t.data_ptr() if t is not None else None

    def flops_(K_):
        return 2.0 * M * NOUT * K_

    def launch6(xin, planes_t, bias, out, m, n, k, products, pipe, x2=None, k1=None, act=0, residual=None, gn=None, tile=0, stream=None,
                ntok=0, ss_mode=2, ss_index=None, preact=None):
        gamma, beta, ss = gn if gn is not None else (None, None, None)
        rc = lib.bf16x6_launch(ptr(xin), ptr(x2), k if k1 is None else k1, xin.stride(0), ptr(planes_t), ptr(bias), ptr(out), out.stride(0),
                               m, n, k, act, ptr(residual), residual.stride(0) if residual is not None else 0,
                               1 if gn is not None else 0, ptr(gamma), ptr(beta), 1e-5, ptr(ss), ss.stride(0) if ss is not None else 0,
                               ss_mode if ss is not None else 0, ptr(ss_index), ptr(preact), preact.stride(0) if preact is not None else 0,
                               ntok if ntok else (80 if gn is not None else 0), products, pipe, tile, stream)
        assert rc == 0, ("bf16x6_launch", rc, products, pipe)

    dev = torch.device("cuda:0")
    B, N = int(opt("--batch", "256")), int(opt("--objects", "80"))
    M, NOUT = B * N, int(opt("--n", "512"))
    variants = [(6, 0), (6, 1), (6, 2), (6, 3), (3, 2), (1, 0), (1, 3)]          # (products, pipeline); pipeline 3 = split once per block at staging
    if opt("--variants", None):                 # e.g. --variants 6:1,6:3 : one process per risky variant, a GPU fault only loses that run
        variants = [tuple(int(v) for v in t.split(":")) for t in opt("--variants", "").split(",")]
    gn_pipes = [pp for pp in ((0, 1, 2, 3) if FM else (1, 2, 3)) if (6, pp) in variants]
    extras = "--no-extras" not in sys.argv      # sections 6 and 7 (two segments, GELU + residual, n = 384, N = 21)
    torch.manual_seed(0)
    for K in [int(x) for x in opt("--k", "512,1024").split(",")]:
        x = torch.nn.functional.silu(torch.randn(M, K, device=dev)) * 1.3
        w = torch.randn(NOUT, K, device=dev) / K ** 0.5
        b = torch.randn(NOUT, device=dev) * 0.1
        planes = torch.empty(3, NOUT, K, device=dev, dtype=torch.int16)
        s = ops.stream_ptr()

        def split(wt):
            split_into(wt, planes, s)

        def run(v, out, xin=x, bias=b):
            launch6(xin, planes, bias, out, M, NOUT, K, v[0], v[1], stream=s)

        # 1. plane split is exact: w1 + w2 + w3 == w bit for bit
        split(w)
        pf = (planes_nk(planes, NOUT, K).contiguous().to(torch.int32) << 16).view(torch.float32)
        exact = bool(torch.equal(pf[0] + pf[1] + pf[2], w))
        print("K=%d  plane split exact: %s" % (K, exact), flush=True)

        # 2. identity check with an asymmetric operand: x = [I | 0] rows, out[m][n] must equal w[n][m % K] (+ bias)
        eye = torch.zeros(M, K, device=dev)
        eye[torch.arange(M, device=dev), torch.arange(M, device=dev) % K] = 1.0
        want = w.t()[torch.arange(M, device=dev) % K] + b
        for v in [v for v in variants if v[0] == 6]:
            out = torch.zeros(M, NOUT, device=dev)
            run(v, out, eye)
            torch.cuda.synchronize()
            print("K=%d  identity check products=%d pipe=%d: max |diff| = %.3e" % (K, v[0], v[1], float((out - want).abs().max())),
                  flush=True)

        # 3. error against f64
        ref = x.double() @ w.double().t() + b.double()
        rms = float(ref.pow(2).mean().sqrt())

        def err(y):
            d = (y.double() - ref).abs()
            return float(d.max()) / rms, float(d.pow(2).mean().sqrt()) / rms

        y_prod = ops.gemm(x, w, b)
        torch.cuda.synchronize()
        print("K=%d  dsc_gemm_f32 (f32 MFMA)        max %.2e  rms %.2e" % ((K,) + err(y_prod)), flush=True)
        print("K=%d  torch f32 matmul               max %.2e  rms %.2e" % ((K,) + err(x @ w.t() + b)), flush=True)
        outs = {}
        for v in variants:
            outs[v] = torch.zeros(M, NOUT, device=dev)
            run(v, outs[v])
            torch.cuda.synchronize()
            print("K=%d  bf16 split products=%d pipe=%d   max %.2e  rms %.2e" % ((K, v[0], v[1]) + err(outs[v])), flush=True)

        # 3a. attribution probes of the second generation (wrong results on purpose): which stream's latency the K loop waits for
        if "--probes" in sys.argv:
            o_ = torch.zeros(M, NOUT, device=dev)
            plist = ((("product form", 0), ("token rows from K tile 0 only", 64), ("weight fragments from K tile 0 only", 256), ("both", 320))
                     if FM else
                     (("product form", 0), ("no split, no plane writes", 64), ("token fragments read once", 128), ("no weight DMA", 256),
                      ("weight fragments read once", 512), ("no block barrier", 1024), ("no token-row loads", 2048),
                      ("no LDS fragment reads at all", 128 + 512), ("no staging at all (DMA, loads, split)", 64 + 256 + 2048),
                      ("MFMAs + barrier only", 64 + 128 + 256 + 512 + 2048), ("MFMAs only", 64 + 128 + 256 + 512 + 1024 + 2048)))
            for name, act in plist:
                for v in [v for v in variants if v[0] == 6]:
                    g_ = torch.cuda.CUDAGraph()                  # graph of 50 launches: no host gaps, sustained clocks
                    launch6(x, planes, b, o_, M, NOUT, K, 6, v[1], act=act, stream=ops.stream_ptr())
                    torch.cuda.synchronize()
                    with torch.cuda.graph(g_):
                        for _ in range(50):
                            launch6(x, planes, b, o_, M, NOUT, K, 6, v[1], act=act, stream=ops.stream_ptr())
                    for _ in range(20):
                        g_.replay()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        g_.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    print("K=%d probe pipe=%d %-44s %.1f us" % (K, v[1], name, e0.elapsed_time(e1)), flush=True)

        # 3c. sustained rate: every kernel alone for ~0.3 s (the power management averages over milliseconds: a 10-launch burst
        #     between other kernels can run above the sustained clock).  Launches go through a captured graph of 50 (no host gaps).
        if "--sustained" in sys.argv:
            o_ = torch.zeros(M, NOUT, device=dev)
            gp_ = ops.make_gemm_args(x, w, o_, b)
            cands = [("dsc_gemm_f32", lambda: ops.run_gemm(gp_))] + [
                ("bf16 split products=%d pipe=%d" % v, (lambda v=v: launch6(x, planes, b, o_, M, NOUT, K, v[0], v[1], stream=ops.stream_ptr())))
                for v in variants if v[0] == 6]
            for rep in range(2):
                for name, fn in cands:
                    g_ = torch.cuda.CUDAGraph()
                    fn()
                    torch.cuda.synchronize()
                    with torch.cuda.graph(g_):
                        for _ in range(50):
                            fn()
                    for _ in range(40):
                        g_.replay()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(60):
                        g_.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) * 1000.0 / 3000
                    print("K=%d sustained (rep %d) %-34s %7.1f us  %6.1f TF f32-equivalent" % (K, rep, name, us, flops_(K) / us / 1e6), flush=True)

        # 3b. phase stamps of one warm launch (second generation only): where the launch's time goes, per block
        if FM and "--stamps" in sys.argv:
            lib.bf16x6_set_stamps.argtypes = [C.c_void_p]
            gamma_, beta_ = torch.rand(NOUT, device=dev) + 0.5, torch.randn(NOUT, device=dev) * 0.1
            ss_, res_ = torch.randn(B, 2 * NOUT, device=dev) * 0.1, torch.randn(M, NOUT, device=dev)
            for v in [v for v in variants if v[0] == 6]:
                for gn_ in (False, True):
                    nblk = (M // 160) * (NOUT // 128)
                    st = torch.zeros(nblk, 8, device=dev, dtype=torch.int64)
                    o_ = torch.zeros(M, NOUT, device=dev)

                    def go():
                        if gn_:
                            launch6(x, planes, b, o_, M, NOUT, K, 6, v[1], residual=res_, gn=(gamma_, beta_, ss_), stream=s)
                        else:
                            run(v, o_)
                    for _ in range(60):
                        go()
                    torch.cuda.synchronize()
                    lib.bf16x6_set_stamps(st.data_ptr())
                    go()
                    torch.cuda.synchronize()
                    lib.bf16x6_set_stamps(None)
                    w_ = st[:, :4].double() * 10.0          # ns (100 MHz wall clock)
                    c_ = st[:, 4:].double()
                    t0 = float(w_[:, 0].min())
                    first, second = slice(0, nblk // 2), slice(nblk // 2, nblk)
                    for name, sl in (("first-half blocks", first), ("second-half blocks", second)):
                        ww, cc = w_[sl], c_[sl]
                        print("K=%d stamps pipe=%d %s %s: start %.1f..%.1f us | prologue %.1f us (%.0f cyc) | K loop %.1f us (%.0f cyc, %.2f GHz) | "
                              "epilogue %.1f us (%.0f cyc) | end %.1f..%.1f us" % (
                                  K, v[1], "GN" if gn_ else "plain", name, (float(ww[:, 0].min()) - t0) / 1e3, (float(ww[:, 0].max()) - t0) / 1e3,
                                  float((ww[:, 1] - ww[:, 0]).mean()) / 1e3, float((cc[:, 1] - cc[:, 0]).mean()),
                                  float((ww[:, 2] - ww[:, 1]).mean()) / 1e3, float((cc[:, 2] - cc[:, 1]).mean()),
                                  float((cc[:, 2] - cc[:, 1]).mean()) / max(float((ww[:, 2] - ww[:, 1]).mean()), 1.0),
                                  float((ww[:, 3] - ww[:, 2]).mean()) / 1e3, float((cc[:, 3] - cc[:, 2]).mean()),
                                  (float(ww[:, 3].min()) - t0) / 1e3, (float(ww[:, 3].max()) - t0) / 1e3), flush=True)

        # 4. timing, round-robin after a clock warm-up; "prod" = the product kernel on the same operands
        yp = torch.empty(M, NOUT, device=dev)
        gp = ops.make_gemm_args(x, w, yp, b)
        names = ["prod"] + variants

        def launch(v):
            if v == "prod":
                ops.run_gemm(gp)
            else:
                run(v, outs[v])

        for _ in range(150):
            launch("prod")
        times = {v: [] for v in names}
        for rnd in range(9):
            order = names[rnd % len(names):] + names[:rnd % len(names)]
            evs = []
            for v in order:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    launch(v)
                e1.record()
                evs.append((v, e0, e1))
            torch.cuda.synchronize()
            for v, e0, e1 in evs:
                times[v].append(e0.elapsed_time(e1) * 100.0)
        flops = 2.0 * M * NOUT * K
        for v in names:
            us = float(np.median(times[v]))
            label = "dsc_gemm_f32" if v == "prod" else "bf16 split products=%d pipe=%d" % v
            extra = "" if v == "prod" else "  bf16 pipe: %.0f TF = %.3f of 2500" % (flops * v[0] / us / 1e6, flops * v[0] / us / 1e6 / 2500)
            print("K=%4d  %-34s %7.1f us [%6.1f..%6.1f]  %6.1f TF f32-equivalent (%.3f of the f32-MFMA peak)%s" % (
                K, label, us, min(times[v]), max(times[v]), flops / us / 1e6, flops / us / 1e6 / 157.3, extra), flush=True)


        # 5. the fused Block epilogue (wave-local GroupNorm): error vs an f64 evaluation next to dsc_gemm_gn_silu_f32, and timing
        if N == 80 and gn_pipes:
            gamma, beta = torch.rand(NOUT, device=dev) + 0.5, torch.randn(NOUT, device=dev) * 0.1
            ss = torch.randn(B, 2 * NOUT, device=dev) * 0.1
            res = torch.randn(M, NOUT, device=dev)
            z = ref.view(B, N, NOUT // 64, 64)
            mu = z.mean(dim=(1, 3), keepdim=True)
            var = z.var(dim=(1, 3), unbiased=False, keepdim=True)
            zn = ((z - mu) / (var + 1e-5).sqrt()).view(B, N, NOUT) * gamma.double() + beta.double()
            zn = zn * (ss[:, None, :NOUT].double() + 1) + ss[:, None, NOUT:].double()
            ref_gn = (zn * torch.sigmoid(zn)).view(M, NOUT) + res.double()
            rms_gn = float(ref_gn.pow(2).mean().sqrt())

            def err_gn(y):
                d = (y.double() - ref_gn).abs()
                return float(d.max()) / rms_gn, float(d.pow(2).mean().sqrt()) / rms_gn

            yp = torch.empty(M, NOUT, device=dev)
            ggn = ops.make_gemm_args(x, w, yp, b, None, res, gamma=gamma, beta=beta, tokens_per_scene=N, scale_shift=ss, ss_mode=2)
            ops.run_gemm(ggn, gn=True)
            torch.cuda.synchronize()
            print("K=%d  GN  dsc_gemm_gn_silu_f32          max %.2e  rms %.2e" % ((K,) + err_gn(yp)), flush=True)
            ygn = {pp: torch.zeros(M, NOUT, device=dev) for pp in gn_pipes}

            def run_gn(pp):
                launch6(x, planes, b, ygn[pp], M, NOUT, K, 6, pp, residual=res, gn=(gamma, beta, ss), stream=s)

            for pp in gn_pipes:
                run_gn(pp)
                torch.cuda.synchronize()
                print("K=%d  GN  bf16 split x6 pipe=%d          max %.2e  rms %.2e" % ((K, pp) + err_gn(ygn[pp])), flush=True)
            # the other conditioning modes of the product epilogue + the saved pre-activation, against dsc_gemm_gn_silu_f32
            pp = gn_pipes[-1]
            ss_slot = torch.randn(N, 2 * NOUT, device=dev) * 0.1                       # DSC_SS_PER_SLOT: row = token % N
            ss_tab = torch.randn(1000, 2 * NOUT, device=dev) * 0.1                     # DSC_SS_BY_INDEX: row = t[scene]
            tvec = torch.randint(0, 1000, (B,), device=dev, dtype=torch.int64)
            ss_tok = torch.randn(M, 2 * NOUT, device=dev) * 0.1                        # DSC_SS_PER_TOKEN
            for mode, sst, idx in ((3, ss_slot, None), (4, ss_tab, tvec), (1, ss_tok, None)):
                ya, yb = torch.empty(M, NOUT, device=dev), torch.zeros(M, NOUT, device=dev)
                pa, pb = torch.empty(M, NOUT, device=dev), torch.zeros(M, NOUT, device=dev)
                gm = ops.make_gemm_args(x, w, ya, b, None, res, gamma=gamma, beta=beta, tokens_per_scene=N, scale_shift=sst, ss_mode=mode,
                                        preact=pa, ss_index=idx)
                ops.run_gemm(gm, gn=True)
                launch6(x, planes, b, yb, M, NOUT, K, 6, pp, residual=res, gn=(gamma, beta, sst), stream=s, ss_mode=mode, ss_index=idx, preact=pb)
                torch.cuda.synchronize()
                print("K=%d  GN  ss_mode=%d pipe=%d vs dsc_gemm_gn_silu_f32: max |diff| / rms = %.2e, preact %.2e" % (
                    K, mode, pp, float((ya - yb).abs().max()) / float(ya.pow(2).mean().sqrt()),
                    float((pa - pb).abs().max()) / float(pa.pow(2).mean().sqrt())), flush=True)
            names = ["prod"] + gn_pipes
            for _ in range(150):
                ops.run_gemm(ggn, gn=True)
            times = {v: [] for v in names}
            for rnd in range(9):
                order = names[rnd % len(names):] + names[:rnd % len(names)]
                evs = []
                for v in order:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        if v == "prod":
                            ops.run_gemm(ggn, gn=True)
                        else:
                            run_gn(v)
                    e1.record()
                    evs.append((v, e0, e1))
                torch.cuda.synchronize()
                for v, e0, e1 in evs:
                    times[v].append(e0.elapsed_time(e1) * 100.0)
            for v in names:
                us = float(np.median(times[v]))
                print("K=%4d  GN  %-30s %7.1f us [%6.1f..%6.1f]  %6.1f TF f32-equivalent (%.3f of the f32-MFMA peak)" % (
                    K, "dsc_gemm_gn_silu_f32" if v == "prod" else "bf16 split x6 pipe=%d" % v, us, min(times[v]), max(times[v]),
                    flops / us / 1e6, flops / us / 1e6 / 157.3), flush=True)


        # 6. the other forms the product needs: two K segments (torch.cat of a skip connection), GELU + residual epilogue, and the
        #    320 x 128 tile for n = 384 -- each against dsc_gemm_f32 on the same operands
        if not extras:
            continue
        if K >= 128:
            h = K // 2
            xa, xb_ = x[:, :h].contiguous(), x[:, h:].contiguous()
            y2 = torch.zeros(M, NOUT, device=dev)
            launch6(xa, planes, b, y2, M, NOUT, K, 6, 1 if FM else (2 if K % 64 == 0 else 1), x2=xb_, k1=h, stream=s)
            torch.cuda.synchronize()
            print("K=%d  two segments (%d+%d)           max %.2e  rms %.2e" % ((K, h, h) + err(y2)), flush=True)
        res2 = torch.randn(M, NOUT, device=dev)
        y3 = torch.zeros(M, NOUT, device=dev)
        launch6(x, planes, b, y3, M, NOUT, K, 6, 1 if FM else 2, act=1, residual=res2, stream=s)
        yp3 = ops.gemm(x, w, b, residual=res2, act_out=1)
        torch.cuda.synchronize()
        print("K=%d  GELU + residual vs dsc_gemm_f32: max |diff| / rms = %.2e" % (K, float((y3 - yp3).abs().max()) / float(yp3.pow(2).mean().sqrt())),
              flush=True)
        n3 = 384
        w3 = torch.randn(n3, K, device=dev) / K ** 0.5
        planes3 = torch.empty(3, n3, K, device=dev, dtype=torch.int16)
        split_into(w3, planes3, s)
        y4 = torch.zeros(M, n3, device=dev)
        ref3 = x.double() @ w3.double().t()
        for pp in (1, 2):
            launch6(x, planes3, None, y4, M, n3, K, 6, pp, stream=s)
            torch.cuda.synchronize()
            d = (y4.double() - ref3).abs()
            print("K=%d  n=384 (320x128 tile) pipe=%d      max %.2e  rms %.2e" % (K, pp, float(d.max()) / float(ref3.pow(2).mean().sqrt()),
                                                                                float(d.pow(2).mean().sqrt()) / float(ref3.pow(2).mean().sqrt())), flush=True)
        yp4 = torch.empty(M, n3, device=dev)
        g4 = ops.make_gemm_args(x, w3, yp4, None)
        tt = {}
        for name, fn in (("dsc_gemm_f32", lambda: ops.run_gemm(g4)), ("bf16 split x6 pipe=2", lambda: launch6(x, planes3, None, y4, M, n3, K, 6, 2, stream=s))):
            for _ in range(30):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                fn()
            e1.record()
            torch.cuda.synchronize()
            tt[name] = e0.elapsed_time(e1) * 1000.0 / 30
            print("K=%4d  n=384  %-24s %7.1f us  %6.1f TF f32-equivalent" % (K, name, tt[name], 2.0 * M * n3 * K / tt[name] / 1e6), flush=True)


    if not extras:
        return
    # 7. scenes of 21 tokens (BASELINE config 2: B = 256, N = 21): 4 scenes x 128 channels per block, scenes padded to 32 rows in LDS
    B2, N2, K = 256, 21, 512
    M2 = B2 * N2
    x = torch.nn.functional.silu(torch.randn(M2, K, device=dev)) * 1.3
    w = torch.randn(NOUT, K, device=dev) / K ** 0.5
    b = torch.randn(NOUT, device=dev) * 0.1
    planes = torch.empty(3, NOUT, K, device=dev, dtype=torch.int16)
    split_into(w, planes, s)
    gamma, beta = torch.rand(NOUT, device=dev) + 0.5, torch.randn(NOUT, device=dev) * 0.1
    ss = torch.randn(B2, 2 * NOUT, device=dev) * 0.1
    res = torch.randn(M2, NOUT, device=dev)
    ref = x.double() @ w.double().t() + b.double()
    z = ref.view(B2, N2, NOUT // 64, 64)
    mu, var = z.mean(dim=(1, 3), keepdim=True), z.var(dim=(1, 3), unbiased=False, keepdim=True)
    zn = ((z - mu) / (var + 1e-5).sqrt()).view(B2, N2, NOUT) * gamma.double() + beta.double()
    zn = zn * (ss[:, None, :NOUT].double() + 1) + ss[:, None, NOUT:].double()
    ref_gn = (zn * torch.sigmoid(zn)).view(M2, NOUT) + res.double()
    rms_gn = float(ref_gn.pow(2).mean().sqrt())
    yp = torch.empty(M2, NOUT, device=dev)
    ggn = ops.make_gemm_args(x, w, yp, b, None, res, gamma=gamma, beta=beta, tokens_per_scene=N2, scale_shift=ss, ss_mode=2)
    ops.run_gemm(ggn, gn=True)
    y21 = torch.zeros(M2, NOUT, device=dev)
    torch.cuda.synchronize()
    print("N=21  GN  dsc_gemm_gn_silu_f32          max %.2e" % (float((yp.double() - ref_gn).abs().max()) / rms_gn), flush=True)
    for pp in ((0, 1) if FM else (1, 2, 3)):
        y21.zero_()
        launch6(x, planes, b, y21, M2, NOUT, K, 6, pp, residual=res, gn=(gamma, beta, ss), stream=s, ntok=N2)
        torch.cuda.synchronize()
        print("N=21  GN  bf16 split x6 pipe=%d          max %.2e" % (pp, float((y21.double() - ref_gn).abs().max()) / rms_gn), flush=True)
    for name, fn in (("dsc_gemm_gn_silu_f32", lambda: ops.run_gemm(ggn, gn=True)),
                     ("bf16 split x6 pipe=1", lambda: launch6(x, planes, b, y21, M2, NOUT, K, 6, 1, residual=res, gn=(gamma, beta, ss), stream=s, ntok=N2)),
                     ("bf16 split x6 pipe=%d" % (0 if FM else 2), lambda: launch6(x, planes, b, y21, M2, NOUT, K, 6, 0 if FM else 2, residual=res, gn=(gamma, beta, ss), stream=s, ntok=N2)),
                     ("bf16 split x6 pipe=%d" % (1 if FM else 3), lambda: launch6(x, planes, b, y21, M2, NOUT, K, 6, 1 if FM else 3, residual=res, gn=(gamma, beta, ss), stream=s, ntok=N2))):
        for _ in range(50):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000.0 / 50
        print("N=21  GN  %-24s %7.1f us  %6.1f TF f32-equivalent (%.3f of the f32-MFMA peak)" % (
            name, us, 2.0 * M2 * NOUT * K / us / 1e6, 2.0 * M2 * NOUT * K / us / 1e6 / 157.3), flush=True)


if __name__ == "__main__":
    main()
