"""GPU: the K-parallel exact-f32 GEMM for launches too small to fill the chip (csrc/gemm_skinny.h, round 6) against the tile kernel of
gemm_core.h it replaces there and against an fp64 evaluation of the same epilogue.

Both kernels multiply the same f32 operands with v_mfma_f32_32x32x2_f32; the K-parallel kernel associates the K sum as eight slice sums,
so the two agree to f32 rounding of a K-long sum (held here to 5e-6 of the output scale, and each to 1e-5 of fp64), not bit for bit.
Shapes: the one-scene generation call (B = 1, N = 12 / 21), a handful of scenes, ragged rows, launches it must leave alone,
two K segments, every (scale, shift) mode, residual (also aliasing the output), saved pre-activation, grouped launches."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + 1000 * len(shape) + sum(shape))
    return ((torch.rand(*shape, generator=g) * 2 - 1) * scale).to(dev())


@pytest.fixture(autouse=True)
def _skinny_on():
    from diffuscene_amd import _lib
    lib = _lib.load()
    prev = lib.dsc_get_skinny()
    lib.dsc_set_skinny(1)
    yield
    lib.dsc_set_skinny(prev)


def both(run):
    """run() with the K-parallel kernel switched off (tile kernel), then on."""
    from diffuscene_amd import _lib
    lib = _lib.load()
    out = []
    for on in (0, 1):
        lib.dsc_set_skinny(on)
        out.append(run())
    lib.dsc_set_skinny(1)
    return out


def close(a, b, tol, what):
    a, b = a.double(), b.double()
    e = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert e < tol, "%s: %.3g >= %.3g" % (what, e, tol)
    return e


def expected_form(m, n, K, unit=1, batch=1):
    """Mirror of gemm_mfma.hip: skinny_plan -- 0: tile kernels, 1: K-parallel with blocks of <= 32 rows, 2: of <= 16 rows (K % 512 == 0)."""
    ncb = n // 64
    for rows in ((16, 32) if (K % 512 == 0 and unit <= 16) else (32,)):
        if unit > rows:
            continue
        r = min((rows // unit) * unit, -(-m // unit) * unit)
        if -(-m // r) * ncb * batch <= 256:
            return 2 if rows == 16 else 1
    return 0


def _act64(x, act):
    from diffuscene_amd import _lib
    if act == _lib.ACT_GELU:
        return torch.nn.functional.gelu(x)
    if act == _lib.ACT_SILU:
        return torch.nn.functional.silu(x)
    return x


@pytest.mark.parametrize("m", [12, 21, 32, 33, 48, 64, 100, 1536])
def test_plain_forms(m):
    """dsc_gemm_f32: bias / GELU / SiLU / residual / in-place accumulation / two K segments, n = 512, 384, 1024, 3072."""
    from diffuscene_amd import _lib, ops
    d = dev()
    for n, k1, k2 in ((512, 512, 0), (512, 512, 512), (384, 512, 0), (512, 128, 0), (1024, 512, 0), (512, 3072, 0), (3072, 512, 0), (512, 384, 0)):
        a, a2 = rnd(m, k1, seed=11), (rnd(m, k2, seed=12) if k2 else None)
        w, b, r = rnd(n, k1 + k2, seed=13, scale=0.06), rnd(n, seed=14), rnd(m, n, seed=15)
        forms = [dict(), dict(bias=b, residual=r), dict(bias=b, act_out=_lib.ACT_GELU), dict(bias=b, act_out=_lib.ACT_SILU, residual=r),
                 dict(inplace=True)]
        for f in forms:
            def run():
                y = r.clone() if f.get("inplace") else torch.empty(m, n, device=d)
                g = ops.make_gemm_args(a, w, y, f.get("bias"), a2, y if f.get("inplace") else f.get("residual"), act_out=f.get("act_out", 0))
                sk = _lib.fn("dsc_gemm_skinny")(g, 0)
                ops.run_gemm(g)
                return sk, y
            (s0, y0), (s1, y1) = both(run)
            assert s0 == 0
            want = expected_form(m, n, k1 + k2)
            assert s1 == want, (m, n, k1, k2, s1, want)
            if not want:                                     # more than one round of blocks: the tile kernels keep it
                continue
            A = torch.cat([a, a2], 1) if k2 else a
            ref = A.double() @ w.double().t()
            if f.get("bias") is not None:
                ref = ref + b.double()
            ref = _act64(ref, f.get("act_out", 0))
            if f.get("inplace") or f.get("residual") is not None:
                ref = ref + r.double()
            assert torch.isfinite(y1).all()
            close(y1, ref, 1e-5, "K-parallel vs fp64 m=%d n=%d K=%d+%d %s" % (m, n, k1, k2, sorted(f)))
            close(y1, y0, 5e-6, "K-parallel vs tile kernel m=%d n=%d K=%d+%d %s" % (m, n, k1, k2, sorted(f)))


def _gn64(z, N, gamma, beta, eps, ss_rows, res):
    M, n = z.shape
    zz = z.view(M // N, N, n // 64, 64)
    mu = zz.mean(dim=(1, 3), keepdim=True)
    var = ((zz - mu) ** 2).mean(dim=(1, 3), keepdim=True)
    h = ((zz - mu) / torch.sqrt(var + eps)).view(M, n) * gamma + beta
    if ss_rows is not None:
        h = h * (ss_rows[:, :n] + 1.0) + ss_rows[:, n:]
    y = torch.nn.functional.silu(h)
    return y + res if res is not None else y


@pytest.mark.parametrize("N,scenes", [(12, 1), (21, 1), (12, 4), (21, 3), (12, 64), (12, 128), (33, 2), (32, 3), (5, 7), (4, 9)])
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_groupnorm_forms(N, scenes, mode):
    """dsc_gemm_gn_silu_f32 (Block.forward as one launch): every (scale, shift) mode, residual, saved pre-activation, two K segments."""
    from diffuscene_amd import _lib, ops
    n, d = 512, dev()
    M = scenes * N
    for k1, k2, res, pre in ((512, 0, False, False), (512, 512, True, True), (256, 0, True, False)):
        a, a2 = rnd(M, k1, seed=N + 1), (rnd(M, k2, seed=N + 2) if k2 else None)
        w, b = rnd(n, k1 + k2, seed=3, scale=0.06), rnd(n, seed=4)
        gamma, beta = rnd(n, seed=5) + 1.5, rnd(n, seed=6)
        r = rnd(M, n, seed=7) if res else None
        rows = {0: 0, 1: M, 2: scenes, 3: N, 4: 1000}[mode]
        ss = rnd(rows, 2 * n, seed=8, scale=0.3) if rows else None
        kw = dict(scale_shift=ss, ss_mode=mode)
        idx = None
        if mode == 4:
            idx = torch.randint(0, 1000, (scenes,), generator=torch.Generator().manual_seed(5)).to(d)
            kw["ss_index"] = idx

        def run():
            y, z = torch.empty(M, n, device=d), (torch.empty(M, n, device=d) if pre else None)
            g = ops.make_gemm_args(a, w, y, b, a2, r, gamma=gamma, beta=beta, tokens_per_scene=N, preact=z, **kw)
            sk = _lib.fn("dsc_gemm_skinny")(g, 1)
            ops.run_gemm(g, gn=True)
            return sk, y, z
        (s0, y0, z0), (s1, y1, z1) = both(run)
        assert s0 == 0
        want = expected_form(M, n, k1 + k2, unit=N) if N <= 32 else 0
        assert s1 == want, (N, scenes, k1, k2, s1, want)
        if not want:                                       # scenes of more than 32 tokens / more than one round of blocks
            continue
        A = torch.cat([a, a2], 1) if k2 else a
        z64 = A.double() @ w.double().t() + b.double()
        tok = torch.arange(M, device=d)
        ssr = None
        if mode:
            rowsel = {1: tok, 2: tok // N, 3: tok % N, 4: None}[mode]
            if mode == 4:
                rowsel = idx[tok // N]
            ssr = ss.double()[rowsel]
        ref = _gn64(z64, N, gamma.double(), beta.double(), 1e-5, ssr, r.double() if res else None)
        assert torch.isfinite(y1).all()
        close(y1, ref, 1e-5, "K-parallel GN vs fp64 N=%d scenes=%d mode=%d K=%d+%d" % (N, scenes, mode, k1, k2))
        close(y1, y0, 5e-6, "K-parallel GN vs tile kernel N=%d scenes=%d mode=%d K=%d+%d" % (N, scenes, mode, k1, k2))
        if pre:
            close(z1, z64, 1e-5, "saved pre-activation vs fp64")
            close(z1, z0, 2e-6, "saved pre-activation vs tile kernel")


def test_grouped_launch_and_refusals():
    """batch = 3 (the hoisted q / k / v style launches), and what the K-parallel kernel leaves to the tile kernels."""
    from diffuscene_amd import _lib, ops
    d = dev()
    m, n, k = 21, 512, 512
    a, w, b = rnd(3, m, k, seed=1), rnd(3, n, k, seed=2, scale=0.06), rnd(3, n, seed=3)

    def run():
        y = torch.empty(3, m, n, device=d)
        g = ops.make_gemm_args(a[0], w[0], y[0], b[0], act_out=_lib.ACT_GELU)
        g.batch, g.sa1, g.sw, g.sbias, g.sy = 3, m * k, n * k, n, m * n
        sk = _lib.fn("dsc_gemm_skinny")(g, 0)
        ops.run_gemm(g)
        return sk, y
    (s0, y0), (s1, y1) = both(run)
    assert (s0, s1) == (0, 2)                # m = 21 plain, K = 512: two blocks of <= 16 rows per channel group
    ref = torch.nn.functional.gelu(torch.einsum("zmk,znk->zmn", a.double(), w.double()) + b.double()[:, None, :])
    close(y1, ref, 1e-5, "grouped K-parallel vs fp64")
    close(y1, y0, 2e-6, "grouped K-parallel vs tile kernel")
    # refusals: K not a multiple of 64, n not a multiple of 64, too many blocks for one round, scenes of more than 32 tokens
    fn = _lib.fn("dsc_gemm_skinny")
    y = torch.empty(m, n, device=d)
    assert fn(ops.make_gemm_args(rnd(m, 96), rnd(n, 96), y), 0) == 0
    assert fn(ops.make_gemm_args(rnd(m, k), rnd(32, k), torch.empty(m, 32, device=d)), 0) == 0
    big = 32 * 40
    assert fn(ops.make_gemm_args(rnd(big, k), rnd(n, k), torch.empty(big, n, device=d)), 0) == 0
    g = ops.make_gemm_args(rnd(40, k), rnd(n, k), torch.empty(40, n, device=d), rnd(n), gamma=rnd(n), beta=rnd(n), tokens_per_scene=40)
    assert fn(g, 1) == 0
    # an unaligned output (a head written at a column offset of the (M, C) tensor) stays on the tile kernel
    wide = torch.empty(m, n + 3, device=d)
    assert fn(ops.make_gemm_args(rnd(m, k), rnd(n, k), wide[:, 3:]), 0) == 0


def test_one_scene_forward_uses_it_and_matches_the_tile_kernels():
    """The whole denoiser forward at B = 1 (the reference's generation call shape): K-parallel on vs off."""
    from diffuscene_amd import _lib
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.workloads import UNCOND_BEDROOM, synth_scene_batch
    torch.manual_seed(0)
    net = Unet1D(**UNCOND_BEDROOM).to(dev())
    x = synth_scene_batch(1, 12, 22, 32, seed=4).to(dev())
    t = torch.tensor([417], device=dev())
    cond = torch.randn(1, 12, 128, device=dev())

    def run():
        with torch.no_grad():
            return net(x, t, cond, None)
    y0, y1 = both(run)
    close(y1, y0, 1e-5, "B=1 forward, K-parallel vs tile kernels")


def test_row_invariant_launches_stay_on_the_tile_kernels():
    """DSC_GEMM_ROW_INVARIANT: the per-step time MLP (m = B rows) and the table built for the captured loops (m = T rows) must give a row
    the same bits -- the flagged launch is refused by the K-parallel kernel and equals the row of a 1000-row launch."""
    from diffuscene_amd import _lib, ops
    d = dev()
    a, w, b = rnd(1000, 512, seed=21), rnd(2048, 512, seed=22, scale=0.06), rnd(2048, seed=23)
    y_all = ops.gemm(a, w, b, act_out=_lib.ACT_GELU, row_invariant=True)
    for m in (1, 4, 128):
        y = torch.empty(m, 2048, device=d)
        g = ops.make_gemm_args(a[:m], w, y, b, act_out=_lib.ACT_GELU, row_invariant=True)
        assert _lib.fn("dsc_gemm_skinny")(g, 0) == 0
        ops.run_gemm(g)
        assert torch.equal(y, y_all[:m]), m
        g2 = ops.make_gemm_args(a[:m], w, y, b, act_out=_lib.ACT_GELU)
        assert _lib.fn("dsc_gemm_skinny")(g2, 0) in (1, 2)
