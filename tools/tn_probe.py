"""Where does the token step of the split-bf16 weight-gradient kernel spend its time?  (tools/tn_probe.hip: the product's block body with
one ingredient removed per variant.)

    python tools/tn_probe.py --build        # cross-compile tools/_build/libtn_probe.so
    python tools/tn_probe.py                # GPU: 32 layers of dW[512][512] over M = 20480 tokens = 256 blocks, one per CU, 640 steps each

Per variant: us per launch (median of rounds, rotating order) and cycles per 32-token step per block at the 2.0 GHz the chip sustains
under bf16-MFMA load.  MFMA issue floor per step and SIMD: 2 waves x 96 MFMAs x 16 cycles = 3072 cycles."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "_build", "libtn_probe.so")
NAMES = {0: "product step", 1: "no operand split (raw bits as pieces)", 2: "no global loads", 3: "no fragment reads",
         4: "no loads, no split, no plane writes", 5: "MFMAs + barrier only"}


def main():
    if "--build" in sys.argv:
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize",
                               os.path.join(ROOT, "tools", "tn_probe.hip"), "-o", SO])
        return
    import torch
    lib = C.CDLL(SO)
    lib.tn_probe.restype = C.c_int
    lib.tn_probe.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    dev = torch.device("cuda:0")
    m, n, k, layers = 20480, 512, 512, 32
    a = torch.randn(layers, m, k, device=dev)
    dy = torch.randn(layers, m, n, device=dev)
    out = torch.empty(layers, n, k, device=dev)
    s = torch.cuda.current_stream().cuda_stream

    def run(p, map_=0):
        rc = lib.tn_probe(p, a.data_ptr(), dy.data_ptr(), out.data_ptr(), m, n, k, layers, map_, s)
        assert rc == 0, rc
    run(0)
    torch.cuda.synchronize()
    ref = torch.einsum("lmn,lmk->lnk", dy[:2].double(), a[:2].double())
    err = float((out[:2].double() - ref).abs().max() / ref.abs().max())
    print("product variant vs f64 on 2 layers: max rel err %.3g" % err)
    assert err < 5e-6
    for p in NAMES:                                     # warm the clocks
        for _ in range(3):
            run(p)
    torch.cuda.synchronize()
    times = {p: [] for p in NAMES}
    for r in range(7):
        order = list(NAMES)[r % 6:] + list(NAMES)[:r % 6]
        for p in order:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                run(p)
            e1.record()
            torch.cuda.synchronize()
            times[p].append(e0.elapsed_time(e1) * 250.0)
    steps = m // 32
    print("\n%-42s %10s %16s" % ("variant", "us/launch", "cycles/step@2GHz"))
    for p, ts in times.items():
        med = sorted(ts)[len(ts) // 2]
        print("%-42s %10.1f %16.0f" % ("%d: %s" % (p, NAMES[p]), med, med * 2000.0 / steps))
    # block -> XCD placement of the PRODUCT step (tools/tn_probe.hip): a layer's 8 tiles on 8 different L2s / on one L2 / all strips hot
    maps = {0: "consecutive ids (product order): a layer's tiles on 8 XCDs", 1: "XCD-grouped: a layer's tiles share one L2",
            2: "all blocks on layer 0 (cache-hit bound)"}
    run(0, 1)
    torch.cuda.synchronize()
    assert float((out[:2].double() - ref).abs().max() / ref.abs().max()) < 5e-6
    mt = {k_: [] for k_ in maps}
    for r in range(7):
        for k_ in list(maps)[r % 3:] + list(maps)[:r % 3]:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                run(0, k_)
            e1.record()
            torch.cuda.synchronize()
            mt[k_].append(e0.elapsed_time(e1) * 250.0)
    print("\nblock placement (product step):")
    for k_, ts in mt.items():
        med = sorted(ts)[len(ts) // 2]
        print("  map %d %-62s %8.1f us %8.0f cycles/step" % (k_, maps[k_], med, med * 2000.0 / steps))
    fl = 2.0 * m * n * k * layers
    print("\nproduct: %.1f TFLOP/s algorithmic (%.0f executed); MFMA issue floor 3072 cycles per step" % (
        fl / sorted(times[0])[3] / 1e6, 6 * fl / sorted(times[0])[3] / 1e6))


if __name__ == "__main__":
    main()
