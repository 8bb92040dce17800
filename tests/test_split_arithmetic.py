"""CPU: the claims behind the split-bf16 GEMM arithmetic (csrc/gemm_split.hip), checked with torch on the CPU, and the library's
dispatch decision (dsc_gemm_arithmetic, a host function: no GPU needed).

* the 3-way round-to-nearest bf16 split is EXACT for every finite f32 of normal magnitude (pieces carry 8 + 8 + 8 mantissa bits and
  bf16 has the f32 exponent range, so no scaling is involved);
* the six products kept (x1w1, x1w2, x2w1, x2w2, x1w3, x3w1) reproduce an f32-accurate product: error against f64 not above a plain
  f32 GEMM's; three products are NOT enough (an order of magnitude worse) -- the reason the 6-product form is the one shipped;
* dispatch: launches that fill the chip take the split kernel, small ones and unsupported shapes stay on the exact-f32 kernel."""
import ctypes as C
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def split3(x):
    parts, r = [], x.clone()
    for _ in range(3):
        p = r.to(torch.bfloat16).float()          # round to nearest even, as v_cvt_pk_bf16_f32
        parts.append(p)
        r = r - p                                 # exact in f32: the residual has at most 16 / 8 significant bits
    return parts


def test_three_piece_split_is_exact():
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(200000, generator=g) * s for s in (1.0, 1e-6, 1e-20, 3e4, 1e30)])
    edge = torch.tensor([0.0, -0.0, 1.0, -1.0, 1.0 + 2 ** -23, 1.0 - 2 ** -24, 65504.0, 3.0e38, -3.0e38, 2.0 ** -100, 1.17549435e-38 * 2 ** 24])
    x = torch.cat([x, edge])
    a, b, c = split3(x)
    assert torch.equal((a + b) + c, x)
    # every piece is a bf16 value, and the pieces shrink by >= 2^-8 each (what makes the dropped products <= 2^-24 relative)
    for p in (a, b, c):
        assert torch.equal(p.to(torch.bfloat16).float(), p)
    nz = x != 0
    assert float((b[nz].abs() / x[nz].abs()).max()) <= 2.0 ** -8 and float((c[nz].abs() / x[nz].abs()).max()) <= 2.0 ** -16


def test_six_products_are_f32_accurate_three_are_not():
    torch.manual_seed(1)
    for K in (128, 512, 1024):
        A = torch.nn.functional.silu(torch.randn(512, K)) * 1.3
        W = torch.randn(K, 256) / K ** 0.5
        ref = A.double() @ W.double()

        def rms(y):
            return float(((y.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))

        a, w = split3(A), split3(W)
        six = [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]
        y6 = torch.zeros(512, 256)
        for i, j in sorted(six, key=lambda t: -(t[0] + t[1])):       # small terms first, as the kernel does
            y6 = y6 + a[i] @ w[j]
        y3 = a[0] @ w[1] + a[1] @ w[0] + a[0] @ w[0]
        e32, e6, e3 = rms(A @ W), rms(y6), rms(y3)
        assert e6 <= 1.1 * e32 + 2e-8, (K, e6, e32)
        assert e3 > 5 * e6 and e3 > 2e-6, (K, e3, e6)


def _decide(env=None):
    """The library's dispatch decision for a few launches (subprocess: DSC_GEMM is read once per process)."""
    code = r'''
import ctypes as C, sys
sys.path.insert(0, %r)
from diffuscene_amd import _lib
lib = _lib.load()
def ask(m, n, k, gn=0, N=0, planes=True, batch=1, y_off=0, k2=0):
    g = _lib.GemmArgs()
    g.a1, g.lda1, g.k1 = 0x1000000, k, k
    if k2:
        g.a2, g.lda2, g.k2 = 0x2000000, k2, k2
    g.w, g.ldw = 0x3000000, k + k2
    g.y, g.ldy = 0x4000000 + y_off, n
    g.m, g.n, g.batch = m, n, batch
    g.sw, g.sy = n * (k + k2), m * n
    g.tokens_per_scene = N
    if planes:
        g.w_planes = 0x5000000
    return lib.dsc_gemm_arithmetic(C.byref(g), gn)
print(ask(20480, 512, 512), ask(20480, 512, 512, gn=1, N=80), ask(10240, 512, 512, gn=1, N=80), ask(5376, 512, 512, gn=1, N=21),
      ask(1536, 512, 512), ask(1536, 512, 512, gn=1, N=12), ask(20480, 512, 512, planes=False), ask(20480, 520, 512),
      ask(20480, 512, 512, y_off=4), ask(80 * 96, 512, 512, gn=1, N=96), ask(20480, 1024, 512, batch=3), ask(20480, 512, 512, k2=512))
''' % ROOT
    e = dict(os.environ)
    e.pop("DSC_GEMM", None)
    if env:
        e.update(env)
    out = subprocess.check_output([sys.executable, "-c", code], env=e, text=True).strip().splitlines()[-1]
    return [int(v) for v in out.split()]


def test_dispatch_decision_of_the_library():
    #        dense  GN80   GN80/B128  GN21   small  GN12  no planes  n%128  unaligned y  N=96  grouped  two segments
    assert _decide() == [1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1]
    assert _decide({"DSC_GEMM": "f32"}) == [0] * 12


def test_arithmetic_switch_is_one_source_of_truth():
    """DSC_GEMM was parsed in three places with different rules.  Now the library owns the switch
    (dsc_get / dsc_set_gemm_arithmetic): the environment is read once, strictly ('split' | 'f32'), unknown values are REJECTED at load,
    and the per-call switch moves the library's dispatch and the host code's view (engine, training plan, bench) together."""
    code = r'''
import sys
sys.path.insert(0, %r)
from diffuscene_amd import _lib
print("OK", _lib.split_enabled())
''' % ROOT
    for val, want in ((None, "OK True"), ("split", "OK True"), ("f32", "OK False")):
        e = dict(os.environ)
        e.pop("DSC_GEMM", None)
        if val is not None:
            e["DSC_GEMM"] = val
        assert subprocess.check_output([sys.executable, "-c", code], env=e, text=True).strip().splitlines()[-1] == want
    for bad in ("fp32", "float", "F32", "bf16"):
        e = dict(os.environ, DSC_GEMM=bad)
        r = subprocess.run([sys.executable, "-c", code], env=e, text=True, capture_output=True)
        assert r.returncode != 0 and "must be 'split' (default) or 'f32'" in r.stderr, (bad, r.stderr[-300:])
    # per-call switch, in this process
    from diffuscene_amd import _lib
    lib = _lib.load()
    start = "split" if _lib.split_enabled() else "f32"
    try:
        assert _lib.set_gemm_arithmetic("f32") == start and not _lib.split_enabled() and lib.dsc_get_gemm_arithmetic() == 0
        g = _lib.GemmArgs()
        g.a1, g.lda1, g.k1, g.w, g.ldw, g.y, g.ldy, g.m, g.n, g.batch = 0x1000000, 512, 512, 0x3000000, 512, 0x4000000, 512, 20480, 512, 1
        g.w_planes = 0x5000000
        import ctypes as C
        assert lib.dsc_gemm_arithmetic(C.byref(g), 0) == 0
        assert _lib.set_gemm_arithmetic("split") == "f32" and _lib.split_enabled()
        assert lib.dsc_gemm_arithmetic(C.byref(g), 0) == 1
        assert lib.dsc_set_gemm_arithmetic(7) != 0 and _lib.split_enabled()
        with pytest.raises(ValueError):
            _lib.set_gemm_arithmetic("fp32")
    finally:
        _lib.set_gemm_arithmetic(start)
