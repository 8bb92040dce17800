"""What would the forward / input-gradient GEMMs gain if their row operand arrived ALREADY split (DESIGN 7, next-list item 1)?

Upper-bound probe: the product's split-bf16 kernel source (csrc/gemm_split.hip, copied at build time -- the product file is not touched)
with the three-way operand split replaced by raw bit moves: the f32 loads, the three LDS plane writes, the fragment reads and all MFMAs
stay, the ~40 VALU instructions per 8 elements of `split8` go.  Results are garbage, the time is what a producer-side split could
reach at best (it would ALSO change the operand traffic: three bf16 planes = 6 B per element instead of 4 B -- not modelled here).

    python tools/split_probe.py --build     # tools/_build/libsplit_probe_{ctl,nosplit}.so (cross-compiles without a GPU)
    python tools/split_probe.py             # on the GPU box: chains of launches in hipGraphs, forms in turn -> stdout (profiles/r04_split_probe.txt)
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BUILD = os.path.join(ROOT, "tools", "_build")
CSRC = os.path.join(ROOT, "diffuscene_amd", "csrc")

NOSPLIT = '''__device__ __forceinline__ void split8(const f32x4 lo, const f32x4 hi, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
    p1 = __builtin_bit_cast(bf16x8, lo);      // PROBE: raw bits instead of the three bf16 pieces
    p2 = __builtin_bit_cast(bf16x8, hi);
    p3 = p1;
}
'''
# Producer side of the same idea (round 5, the GO / NO-GO of the plan-wide "activations as three bf16 planes" format): the epilogue ALSO
# splits its four outputs exactly into three bf16 pieces and stores them into planes [3][m][n] (the pointer rides in the unused
# actgrad_x field) -- "both": next to the f32 result (what a training step needs: residual paths, GroupNorm / LayerNorm backward and the
# weight-gradient operand read f32), "planes": instead of it (a sampling-only intermediate such as block1's output).
PROD = '''{
                        uint16_t* const pb__ = reinterpret_cast<uint16_t*>(const_cast<float*>(p.actgrad_x));
                        const int64_t eo__ = (int64_t)(row0 + srow + l15 + i * 16) * p.n + cbase + j * 16, pl__ = (int64_t)p.m * p.n;
                        const unsigned a0__ = cvt_pk_bf16(y[0], y[1]), a1__ = cvt_pk_bf16(y[2], y[3]);
                        const float r0__ = y[0] - bf_lo(a0__), r1__ = y[1] - bf_hi(a0__), r2__ = y[2] - bf_lo(a1__), r3__ = y[3] - bf_hi(a1__);
                        const unsigned b0__ = cvt_pk_bf16(r0__, r1__), b1__ = cvt_pk_bf16(r2__, r3__);
                        const unsigned c0__ = cvt_pk_bf16(r0__ - bf_lo(b0__), r1__ - bf_hi(b0__)), c1__ = cvt_pk_bf16(r2__ - bf_lo(b1__), r3__ - bf_hi(b1__));
                        *reinterpret_cast<uint2*>(pb__ + eo__) = uint2{a0__, a1__};
                        *reinterpret_cast<uint2*>(pb__ + pl__ + eo__) = uint2{b0__, b1__};
                        *reinterpret_cast<uint2*>(pb__ + 2 * pl__ + eo__) = uint2{c0__, c1__};
                    }'''
STORE = "*reinterpret_cast<f32x4*>(ob + (int64_t)i * 16 * p.ldy + j * 16) = y;"

TAIL = '''
extern "C" int probe_gn(const dsc_gemm_args* a, void* stream) { return dsc_split::launch<true, 2, 4, 5>(a, a->tokens_per_scene, static_cast<hipStream_t>(stream)); }
extern "C" int probe_plain(const dsc_gemm_args* a, void* stream) { return dsc_split::launch<false, 2, 4, 5>(a, 80, static_cast<hipStream_t>(stream)); }
'''


def build():
    os.makedirs(BUILD, exist_ok=True)
    src = open(os.path.join(CSRC, "gemm_split.hip")).read()
    a = src.index("__device__ __forceinline__ void split8(")
    b = src.index("// ----", a)
    nosplit = src[:a] + NOSPLIT + "\n" + src[b:]
    assert src.count(STORE) == 2                      # the plain and the GroupNorm epilogue
    gn_at = src.rindex(STORE)                         # the GroupNorm epilogue's store (the second one)

    def producer(text, keep_f32):
        at = text.rindex(STORE)
        return text[:at] + (STORE + " " if keep_f32 else "") + PROD + text[at + len(STORE):]
    variants = (("ctl", src), ("nosplit", nosplit), ("prod_both", producer(src, True)), ("prod_planes", producer(src, False)),
                ("nosplit_prod_both", producer(nosplit, True)))
    del gn_at
    for name, text in variants:
        path = os.path.join(BUILD, "gemm_split_%s.hip" % name)
        open(path, "w").write(text + TAIL)
        so = os.path.join(BUILD, "libsplit_probe_%s.so" % name)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize",
                               "-I", CSRC, "-I", os.path.join(ROOT, "include"), path, "-o", so])
        print("built", so)


def planes_probe(a):
    """GO / NO-GO of the plane format on a ResnetBlock forward chain (block1: GN, no residual -> block2: GN + residual), B = 256, N = 80.
    Every form is a chain of `layers` ResnetBlocks (2 launches each) in a hipGraph; ms per ResnetBlock, median of `rounds`, forms in turn.
      product          the product kernel
      bound: sampling  block1 writes PLANES ONLY (its output has one consumer), block2 reads them (consumer = the no-split upper bound:
                       it still loads 4 B per element, a real one loads 6 B) and writes f32 + planes for the next block
      bound: training  every launch writes f32 AND planes (the backward reads f32), every consumer is the no-split upper bound"""
    import torch
    from diffuscene_amd import _lib, ops
    from diffuscene_amd._lib import SS_PER_SCENE
    names = ("ctl", "nosplit", "prod_both", "prod_planes", "nosplit_prod_both")
    libs = {}
    for name in names:
        L = C.CDLL(os.path.join(BUILD, "libsplit_probe_%s.so" % name))
        L.probe_gn.restype = C.c_int
        L.probe_gn.argtypes = [C.POINTER(_lib.GemmArgs), C.c_void_p]
        libs[name] = L
    dev = torch.device("cuda:0")
    B, N, D = a.batch, 80, 512
    M = B * N
    g = torch.Generator().manual_seed(0)

    def rnd(*shape, scale=1.0):
        return ((torch.rand(*shape, generator=g) * 2 - 1) * scale).to(dev)
    nl = a.layers
    W = [rnd(D, D, scale=0.06) for _ in range(2 * nl)]
    P = ops.split_planes([(w, None, False) for w in W])
    bias = [rnd(D, scale=0.05) for _ in range(2 * nl)]
    gamma, beta = [rnd(D, scale=0.1) + 1 for _ in range(2 * nl)], [rnd(D, scale=0.1) for _ in range(2 * nl)]
    ss = [rnd(B, 2 * D, scale=0.2) for _ in range(nl)]
    x = [torch.empty(M, D, device=dev) for _ in range(3)]          # block input / h / block output rotate
    planes = [torch.empty(3, M, D, device=dev, dtype=torch.int16) for _ in range(2)]

    def block_args(i):
        xin, h, out = x[(2 * i) % 3], x[(2 * i + 1) % 3], x[(2 * i + 2) % 3]
        a1 = ops.make_gemm_args(xin, W[2 * i], h, bias[2 * i], None, None, gamma=gamma[2 * i], beta=beta[2 * i], eps=1e-5, tokens_per_scene=N,
                                scale_shift=ss[i], ss_mode=SS_PER_SCENE, w_planes=P[2 * i])
        a2 = ops.make_gemm_args(h, W[2 * i + 1], out, bias[2 * i + 1], None, xin, gamma=gamma[2 * i + 1], beta=beta[2 * i + 1], eps=1e-5,
                                tokens_per_scene=N, w_planes=P[2 * i + 1])
        a1.actgrad_x, a2.actgrad_x = planes[0].data_ptr(), planes[1].data_ptr()
        return a1, a2
    args = [block_args(i) for i in range(nl)]
    forms = [("product", "ctl", "ctl"),
             ("bound: sampling (h as planes only)", "prod_planes", "nosplit_prod_both"),
             ("bound: training (f32 + planes everywhere)", "nosplit_prod_both", "nosplit_prod_both"),
             ("(consumer bound alone: no split, no planes written)", "nosplit", "nosplit"),
             ("(producer cost alone: f32 + planes, split kept)", "prod_both", "prod_both")]
    graphs = {}
    for title, l1, l2 in forms:
        f1, f2 = libs[l1].probe_gn, libs[l2].probe_gn

        def body(st, f1=f1, f2=f2):
            for a1, a2 in args:
                assert f1(C.byref(a1), st) == 0 and f2(C.byref(a2), st) == 0
        for t in x:
            t.uniform_(-1, 1)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            body(side.cuda_stream)
        torch.cuda.current_stream().wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            body(torch.cuda.current_stream().cuda_stream)
        graphs[title] = gr
    for gr in graphs.values():
        for _ in range(10):
            gr.replay()
    torch.cuda.synchronize()
    times = {k: [] for k in graphs}
    reps = 10
    for r in range(a.rounds):
        order = list(graphs)
        if r & 1:
            order.reverse()
        for k in order:
            for t in x:
                t.uniform_(-1, 1)                         # (the no-split forms leave garbage behind)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                graphs[k].replay()
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) * 1e3 / (reps * nl))
    med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    print("# ResnetBlock forward chain (block1: GN + scale/shift + SiLU -> block2: GN + SiLU + residual), M = %d tokens (B = %d x N = 80), n = K = 512," % (M, B))
    print("# tile <2,4,5>; %d ResnetBlocks per hipGraph, forms replayed in turn, median of %d rounds; us per ResnetBlock (two launches)" % (nl, a.rounds))
    base = med["product"]
    for title, _, _ in forms:
        print("%-58s %8.2f us  %+6.1f %%" % (title, med[title], 100.0 * (med[title] / base - 1.0)))
    print("# GO needed the bound forms >= 8 %% FASTER than the product (VERDICT r4 item 2); a real consumer also loads 6 B per element")
    print("# instead of 4 B and pays its LDS-DMA issue slots, so the bounds are optimistic.")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--layers", type=int, default=16)
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--planes", action="store_true", help="round 5: the producer-side cost and the ResnetBlock chain bound (profiles/r05_planes.txt)")
    a = ap.parse_args()
    if a.build:
        return build()
    import torch
    from diffuscene_amd import _lib, ops
    from diffuscene_amd._lib import SS_PER_SCENE
    _lib.load()
    if a.planes:
        return planes_probe(a)
    libs = {}
    for name in ("ctl", "nosplit"):
        L = C.CDLL(os.path.join(BUILD, "libsplit_probe_%s.so" % name))
        for f in (L.probe_gn, L.probe_plain):
            f.restype = C.c_int
            f.argtypes = [C.POINTER(_lib.GemmArgs), C.c_void_p]
        libs[name] = L
    dev = torch.device("cuda:0")
    B, N, D = a.batch, 80, 512
    M = B * N
    g = torch.Generator().manual_seed(0)

    def rnd(*shape, scale=1.0):
        return ((torch.rand(*shape, generator=g) * 2 - 1) * scale).to(dev)
    nl = a.layers
    W = [rnd(D, D, scale=0.06) for _ in range(nl)]
    P = ops.split_planes([(w, None, False) for w in W])
    bias = [rnd(D, scale=0.05) for _ in range(nl)]
    gamma, beta = [rnd(D, scale=0.1) + 1 for _ in range(nl)], [rnd(D, scale=0.1) for _ in range(nl)]
    ss = [rnd(B, 2 * D, scale=0.2) for _ in range(nl)]
    x0 = rnd(M, D)
    bufs = [torch.empty(M, D, device=dev) for _ in range(3)]

    def chain(gn, residual):
        out = []
        for i in range(nl):
            xin = x0 if i == 0 else bufs[(i - 1) % 3]
            res = bufs[(i + 1) % 3] if residual else None        # a third buffer: not the input, not the output
            if gn:
                out.append(ops.make_gemm_args(xin, W[i], bufs[i % 3], bias[i], None, res, gamma=gamma[i], beta=beta[i], eps=1e-5,
                                              tokens_per_scene=N, scale_shift=ss[i], ss_mode=SS_PER_SCENE, w_planes=P[i]))
            else:
                out.append(ops.make_gemm_args(xin, W[i], bufs[i % 3], bias[i], None, res, w_planes=P[i]))
        return out
    bufs[1].copy_(x0); bufs[2].copy_(x0)
    forms = [("Block launch (GN + scale/shift + SiLU)", True, False, "probe_gn"), ("Block launch + residual", True, True, "probe_gn"),
             ("plain 512 -> 512 (input gradient shape)", False, False, "probe_plain"), ("plain + residual", False, True, "probe_plain")]
    print("# M = %d tokens (B = %d x N = 80), n = K = 512, tile <2,4,5> (160 x 256, 8 waves); chains of %d dependent launches in hipGraphs," % (M, B, nl))
    print("# forms replayed in turn, median of %d rounds; ctl = the product source compiled by this tool, nosplit = split8 replaced by bit moves" % a.rounds)
    print("%-44s %10s %10s %8s" % ("launch", "ctl us", "nosplit us", "gain"))
    for title, gn, residual, fname in forms:
        args = chain(gn, residual)
        graphs = {}
        for name, L in libs.items():
            fn = getattr(L, fname)

            def body(st, fn=fn):
                for s_ in args:
                    rc = fn(C.byref(s_), st)
                    assert rc == 0, rc
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                body(side.cuda_stream)
            torch.cuda.current_stream().wait_stream(side)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                body(torch.cuda.current_stream().cuda_stream)
            graphs[name] = gr
        for gr in graphs.values():
            for _ in range(10):
                gr.replay()
        torch.cuda.synchronize()
        times = {k: [] for k in graphs}
        reps = 10
        for r in range(a.rounds):
            names = list(graphs)
            if r & 1:
                names.reverse()
            for name in names:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    graphs[name].replay()
                e1.record()
                torch.cuda.synchronize()
                times[name].append(e0.elapsed_time(e1) * 1e3 / (reps * nl))
        med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
        print("%-44s %10.2f %10.2f %+7.1f %%" % (title, med["ctl"], med["nosplit"], 100.0 * (med["nosplit"] / med["ctl"] - 1.0)))
        x0.normal_()                                            # the nosplit form leaves garbage (possibly non-finite) in the buffers
        for t in bufs:
            t.copy_(x0)


if __name__ == "__main__":
    sys.exit(main())
