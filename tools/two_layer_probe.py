"""GO / NO-GO harness of the two-layer ResnetBlock launch (tools/two_layer_probe.hip).

    python tools/two_layer_probe.py --build          # cross-compile tools/_build/libtwo_layer_probe.so (no GPU needed)
    python tools/two_layer_probe.py [--batch 256]    # on the GPU box: correctness, then sustained timing -> stdout (profiles/r04_two_layer.txt)

Workload = a chain of ResnetBlocks of the headline shape (B scenes x 80 tokens, 512 -> 512 -> 512 channels, time scale/shift on block1,
residual on block2), every block reading the previous block's output (ping-pong buffers), exactly the dependency structure of the
denoiser.  Two forms of the same chain are captured in hipGraphs and replayed in turn:
  product   2 launches per ResnetBlock (dsc_gemm_gn_silu_f32 twice)
  pair      1 persistent launch per ResnetBlock (probe_gn_pair): block2's tiles wait on an L2 flag set by the column blocks of their row tile
GO (VERDICT): K=512 GN pair <= 108 us (product: ~58.6 + 62.5).  The pair form must reproduce the product bit for bit.
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "_build", "libtwo_layer_probe.so")


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize",
                           os.path.join(ROOT, "tools", "two_layer_probe.hip"), "-o", SO])
    print("built", SO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--objects", type=int, default=80, help="tokens per scene: 80 (living; B=256 eight-wave tile, B=128 four-wave tile) or 21 (bedroom)")
    ap.add_argument("--blocks", type=int, default=8, help="ResnetBlocks per captured chain")
    ap.add_argument("--rounds", type=int, default=9)
    a = ap.parse_args()
    if a.build:
        return build()
    import torch
    from diffuscene_amd import _lib, ops
    from diffuscene_amd._lib import SS_PER_SCENE
    lib = _lib.load()
    probe = C.CDLL(SO)
    probe.probe_gn_pair.restype = C.c_int
    probe.probe_gn_pair.argtypes = [C.POINTER(_lib.GemmArgs), C.POINTER(_lib.GemmArgs), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    dev = torch.device("cuda:0")
    B, N, D = a.batch, a.objects, 512
    M = B * N
    g = torch.Generator().manual_seed(0)

    def rnd(*shape, scale=1.0):
        return ((torch.rand(*shape, generator=g) * 2 - 1) * scale).to(dev)
    nb = a.blocks
    W1 = [rnd(D, D, scale=0.06) for _ in range(nb)]
    W2 = [rnd(D, D, scale=0.06) for _ in range(nb)]
    P1 = ops.split_planes([(w, None, False) for w in W1])
    P2 = ops.split_planes([(w, None, False) for w in W2])
    b1, b2 = [rnd(D, scale=0.05) for _ in range(nb)], [rnd(D, scale=0.05) for _ in range(nb)]
    ga1, ga2 = [rnd(D, scale=0.1) + 1 for _ in range(nb)], [rnd(D, scale=0.1) + 1 for _ in range(nb)]
    be1, be2 = [rnd(D, scale=0.1) for _ in range(nb)], [rnd(D, scale=0.1) for _ in range(nb)]
    ss = [rnd(B, 2 * D, scale=0.2) for _ in range(nb)]
    x0 = rnd(M, D)
    bufs = [torch.empty(M, D, device=dev) for _ in range(2)]        # block outputs, ping-pong
    h = torch.empty(M, D, device=dev)                                # block1 output of the current ResnetBlock

    def args_of(i, xin, yout):
        a1 = ops.make_gemm_args(xin, W1[i], h, b1[i], None, None, gamma=ga1[i], beta=be1[i], eps=1e-5, tokens_per_scene=N,
                                scale_shift=ss[i], ss_mode=SS_PER_SCENE, w_planes=P1[i])
        a2 = ops.make_gemm_args(h, W2[i], yout, b2[i], None, xin, gamma=ga2[i], beta=be2[i], eps=1e-5, tokens_per_scene=N,
                                w_planes=P2[i])
        return a1, a2
    chain = []
    for i in range(nb):
        xin = x0 if i == 0 else bufs[(i - 1) & 1]
        chain.append(args_of(i, xin, bufs[i & 1]))
    assert all(ops.gemm_uses_split(a1, gn=True) and ops.gemm_uses_split(a2, gn=True) for a1, a2 in chain), "launches must run the split kernel"
    flags = torch.zeros(1024, dtype=torch.int32, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    fn = lib.dsc_gemm_gn_silu_f32

    def run_product(s):
        for a1, a2 in chain:
            _lib.check(fn(C.byref(a1), s), "gn1")
            _lib.check(fn(C.byref(a2), s), "gn2")

    def run_pair(s, fence):
        for a1, a2 in chain:
            rc = probe.probe_gn_pair(C.byref(a1), C.byref(a2), flags.data_ptr(), err.data_ptr(), fence, s)
            assert rc == 0, rc

    # ---- correctness: the pair form reproduces the product bit for bit
    s = ops.stream_ptr()
    run_product(s)
    torch.cuda.synchronize()
    want = bufs[(nb - 1) & 1].clone()
    want_h = h.clone()
    for fence in (1, 0):
        for t in bufs + [h]:
            t.fill_(float("nan"))
        run_pair(s, fence)
        torch.cuda.synchronize()
        ok = torch.equal(bufs[(nb - 1) & 1], want) and torch.equal(h, want_h)
        print("pair launch, fence=%d: bit-identical to the two product launches over a chain of %d ResnetBlocks: %s   (spin timeout flag %d)"
              % (fence, nb, ok, int(err.item())))
        if not ok:
            d = (bufs[(nb - 1) & 1] - want).abs()
            print("   max |diff| %.3g, mismatching elements %d of %d" % (float(d.nan_to_num(1e9).max()), int((bufs[(nb - 1) & 1] != want).sum()), want.numel()))

    # ---- sustained timing: hipGraph replay, forms in turn, median of the rounds
    graphs = {}
    for name, body in (("product (2 launches / ResnetBlock)", lambda st: run_product(st)), ("pair, fence=0", lambda st: run_pair(st, 0)),
                       ("pair, fence=1 (agent-scope release / acquire)", lambda st: run_pair(st, 1))):
        gr = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            body(side.cuda_stream)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(gr):
            body(torch.cuda.current_stream().cuda_stream)
        graphs[name] = gr
    for gr in graphs.values():                        # clocks ramp for milliseconds after an idle sync: warm up every form
        for _ in range(20):
            gr.replay()
    torch.cuda.synchronize()
    times = {k: [] for k in graphs}
    reps = 20
    for r in range(a.rounds):
        names = list(graphs)
        names = names[r % len(names):] + names[:r % len(names)]            # rotating order
        for name in names:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                graphs[name].replay()
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) * 1e3 / (reps * nb))
    print("\nus per ResnetBlock (block1 + block2, M=%d, n=K=512), sustained graph replay, median of %d rounds [min .. max]:" % (M, a.rounds))
    base = None
    for name, ts in times.items():
        ts = sorted(ts)
        med = ts[len(ts) // 2]
        base = base or med
        print("  %-50s %7.2f  [%6.2f .. %6.2f]   %+5.1f %% vs product" % (name, med, ts[0], ts[-1], 100.0 * (med / base - 1.0)))
    print("spin timeout flag after timing: %d" % int(err.item()))
    print("GO criterion (VERDICT r3): pair <= 108 us per ResnetBlock")


if __name__ == "__main__":
    sys.exit(main())
