"""GPU: the wave-autonomous form of the split-bf16 GEMM (csrc/gemm_split_wave.h, round 6) against the block-staged kernel
(csrc/gemm_split.hip) it replaces where a launch qualifies.  Both run the same per-element sequence of MFMAs and epilogue operations:
every form must agree BIT FOR BIT -- so every golden comparison made with one family holds for the other -- and the plane layouts the two
families read must hold the same values."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + 1000 * len(shape) + sum(shape))
    return ((torch.rand(*shape, generator=g) * 2 - 1) * scale).to(dev())


@pytest.fixture(autouse=True)
def _needs_split():
    from diffuscene_amd import _lib
    if not _lib.split_enabled():
        pytest.skip("exact-f32 arithmetic selected")
    prev = _lib.set_split_wave(True)
    yield
    _lib.set_split_wave(prev)


def both_families(run):
    """run() under the block-staged kernels, then under the wave-autonomous ones -> (tile, result) of each."""
    from diffuscene_amd import _lib
    out = []
    for wave in (False, True):
        _lib.set_split_wave(wave)
        out.append(run())
    _lib.set_split_wave(True)
    return out


def test_fragment_major_planes_hold_the_same_values():
    """dsc_split_bf16x3_f32 with the fragment flag == the row-major planes permuted (ops.fragment_major), plain and transposed."""
    from diffuscene_amd import ops
    w = rnd(384, 160, seed=1)
    (row, frag, rowt, fragt) = ops.split_planes([(w, None, 0), (w, None, 2), (w, None, 1), (w, None, 3)])
    assert torch.equal(ops.fragment_major(row), frag)
    assert torch.equal(ops.fragment_major(rowt), fragt)
    assert tuple(fragt.shape) == (3, 160, 384)


@pytest.mark.parametrize("N,scenes", [(80, 256), (70, 256), (48, 256), (33, 256), (21, 256), (80, 512), (80, 210), (80, 128), (70, 120)])
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_groupnorm_forms_agree_bit_for_bit(N, scenes, mode):
    """Block.forward as one launch: no / per-token / per-scene / per-slot / per-timestep (scale, shift); residual; saved pre-activation;
    one and two K segments; whole and ragged last MFMA block (N = 70, 33, 21); 1, 2 and 0.82 rounds of waves."""
    from diffuscene_amd import _lib, ops
    n, d = 512, dev()
    M = scenes * N
    for k1, k2, res, pre in ((512, 0, False, False), (512, 512, True, True), (256, 0, True, False)):
        a, a2 = rnd(M, k1, seed=N + 1), (rnd(M, k2, seed=N + 2) if k2 else None)
        w, b = rnd(n, k1 + k2, seed=3, scale=0.06), rnd(n, seed=4)
        gamma, beta = rnd(n, seed=5) + 1.5, rnd(n, seed=6)
        r = rnd(M, n, seed=7) if res else None
        (pl,) = ops.split_planes([(w, None, False)])
        rows = {0: 0, 1: M, 2: scenes, 3: N, 4: 1000}[mode]
        ss = rnd(rows, 2 * n, seed=8, scale=0.3) if rows else None
        kw = dict(scale_shift=ss, ss_mode=mode)
        if mode == 4:
            kw["ss_index"] = torch.randint(0, 1000, (scenes,), generator=torch.Generator().manual_seed(5)).to(d)

        def run():
            y, z = torch.empty(M, n, device=d), (torch.empty(M, n, device=d) if pre else None)
            g = ops.make_gemm_args(a, w, y, b, a2, r, gamma=gamma, beta=beta, tokens_per_scene=N, preact=z, w_planes=pl, **kw)
            tile = _lib.fn("dsc_gemm_split_tile")(g, 1)
            ops.run_gemm(g, gn=True)
            return tile, y, z
        (tb, yb, zb), (tw, yw, zw) = both_families(run)
        waves = scenes * n // 128
        if waves * 5 >= -(-waves // 1024) * 1024 * 4:
            assert tw == _lib.TILE_WAVE_GN and tb != tw, (tb, tw)
        elif N > 64 and 2 * waves <= 1024 and 2 * waves * 5 >= 1024 * 4:
            assert tw == _lib.TILE_WAVE_GN_64 and tb != tw, (tb, tw)          # half-size launches: waves of 80 x 64
        assert torch.isfinite(yw).all()
        assert torch.equal(yb, yw), "GroupNorm GEMM N=%d mode=%d K=%d+%d: wave kernel != block kernel" % (N, mode, k1, k2)
        if pre:
            assert torch.equal(zb, zw), "saved pre-activation differs"


@pytest.mark.parametrize("M", [20480, 20480 - 37, 40960])
def test_dense_forms_agree_bit_for_bit(M):
    """dsc_gemm_f32 on dense rows: bias / activation / residual, the training-step epilogues (pre-activation also stored; result times
    act'(saved pre-activation)), two K segments, residual == output (in-place accumulation), a ragged last group of rows."""
    from diffuscene_amd import _lib, ops
    d = dev()
    for n, k1, k2 in ((512, 512, 0), (512, 512, 512), (1024, 512, 0), (512, 1024, 0)):
        a, a2 = rnd(M, k1, seed=11), (rnd(M, k2, seed=12) if k2 else None)
        w, b, r, u = rnd(n, k1 + k2, seed=13, scale=0.06), rnd(n, seed=14), rnd(M, n, seed=15), rnd(M, n, seed=16, scale=2.0)
        (pl,) = ops.split_planes([(w, None, False)])
        forms = [dict(), dict(bias=b, residual=r), dict(bias=b, act_out=_lib.ACT_GELU), dict(bias=b, act_out=_lib.ACT_SILU, preact=True),
                 dict(act_out=_lib.ACT_GELU, actgrad_x=u, residual=r), dict(act_out=_lib.ACT_SILU, actgrad_x=u), dict(inplace=True)]
        for f in forms:
            def run():
                y = r.clone() if f.get("inplace") else torch.empty(M, n, device=d)
                z = torch.empty(M, n, device=d) if f.get("preact") else None
                g = ops.make_gemm_args(a, w, y, f.get("bias"), a2, y if f.get("inplace") else f.get("residual"), act_out=f.get("act_out", 0),
                                       preact=z, actgrad_x=f.get("actgrad_x"), w_planes=pl)
                tile = _lib.fn("dsc_gemm_split_tile")(g, 0)
                ops.run_gemm(g)
                return tile, y, z
            (tb, yb, zb), (tw, yw, zw) = both_families(run)
            assert tw == _lib.TILE_WAVE_DENSE and tb not in (tw, -1), (tb, tw, n, k1, k2)
            assert torch.equal(yb, yw), "dense GEMM n=%d K=%d+%d %s: wave kernel != block kernel" % (n, k1, k2, sorted(f))
            if zb is not None:
                assert torch.equal(zb, zw)


def test_grouped_dense_launch_agrees():
    """batch > 1: the problems' weights are the row blocks of one stacked matrix (the encoder / decoder MLP layers)."""
    from diffuscene_amd import _lib, ops
    d, M, n, K, Z = dev(), 20480, 512, 256, 3
    a, w, b = rnd(Z, M, K, seed=21), rnd(Z * n, K, seed=22, scale=0.08), rnd(Z * n, seed=23)
    (pl,) = ops.split_planes([(w, None, False)])

    def run():
        y = torch.empty(Z, M, n, device=d)
        g = ops.make_gemm_args(a[0], w[:n], y[0], b[:n], act_out=_lib.ACT_GELU)
        g.batch, g.sa1, g.sw, g.sy, g.sbias = Z, M * K, n * K, M * n, n
        ops.attach_planes(g, pl)
        tile = _lib.fn("dsc_gemm_split_tile")(g, 0)
        ops.run_gemm(g)
        return tile, y
    (tb, yb), (tw, yw) = both_families(run)
    assert tw == _lib.TILE_WAVE_DENSE and tb not in (tw, -1), (tb, tw)
    assert torch.equal(yb, yw)


def test_wrong_plane_layout_is_an_error():
    """A launch whose planes have the other family's layout fails loudly (never computed on the wrong bytes)."""
    from diffuscene_amd import ops
    a, w = rnd(20480, 512, seed=1), rnd(512, 512, seed=2, scale=0.05)
    (pl,) = ops.split_planes([(w, None, False)])
    y = torch.empty(20480, 512, device=dev())
    g = ops.make_gemm_args(a, w, y)
    want = ops.planes_layout(g)
    assert want == ops.PLANES_FRAGMENT
    ops.attach_planes(g, pl, layout=ops.PLANES_ROWMAJOR)        # claims row-major where the launch wants fragment-major
    with pytest.raises(RuntimeError, match="dsc_gemm_f32"):
        ops.run_gemm(g)


@pytest.mark.parametrize("N,mode", [(80, 2), (80, 0), (70, 2)])
def test_groupnorm_backward_epilogue_matches_the_two_launch_form(N, mode):
    """dsc_gemm_f32 with the GroupNorm-backward epilogue (gnb_*: the input-gradient GEMM of the NEXT layer writes the gradient w.r.t. a fused Block's
    pre-norm activation, its per-scene partial sums and d(scale, shift)) against the two launches it replaces -- the same product without the
    epilogue, then dsc_gn_silu_bwd_f32 -- and against an f64 evaluation of the same formulas."""
    from diffuscene_amd import _lib, ops
    d, scenes, n, K = dev(), 256, 512, 512
    M = scenes * N
    dy2, w = rnd(M, K, seed=41), rnd(n, K, seed=42, scale=0.06)          # dh = dy2 . w^T
    z, gamma, beta = rnd(M, n, seed=43, scale=2.0), rnd(n, seed=44) + 1.5, rnd(n, seed=45)
    ss = rnd(scenes, 2 * n, seed=46, scale=0.3) if mode == 2 else None
    (pl,) = ops.split_planes([(w, None, False)])
    dh = ops.gemm(dy2, w, None, w_planes=pl)
    dz_ref, dg_ref, db_ref, dbias_ref, dss_ref = ops.gn_silu_bwd(z, dh, gamma, beta, ss, mode, scenes, N)
    dz = torch.full((M, n), float("nan"), device=d)
    part = torch.full((scenes, 3 * n), float("nan"), device=d)
    dss = torch.full((scenes, 2 * n), float("nan"), device=d) if mode == 2 else None
    pp = part.data_ptr()
    g = ops.make_gemm_args(dy2, w, dz, gamma=gamma, beta=beta, tokens_per_scene=N, scale_shift=ss, ss_mode=mode,
                           gnb=dict(z=z, dgamma=pp + 4 * n, dbeta=pp + 8 * n, dbias=pp, pstride=3 * n, dss=dss), w_planes=pl)
    assert _lib.fn("dsc_gemm_split_tile")(g, 0) == _lib.TILE_WAVE_DENSE
    ops.run_gemm(g)
    red = part.double().sum(0)

    def rel(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    assert torch.isfinite(dz).all() and rel(dz, dz_ref) < 2e-6, rel(dz, dz_ref)
    assert rel(red[:n], dbias_ref) < 1e-5 and rel(red[n:2 * n], dg_ref) < 1e-5 and rel(red[2 * n:], db_ref) < 1e-5
    if mode == 2:
        assert rel(dss, dss_ref) < 1e-5
    # f64 evaluation of the formulas (train.hip: gn_silu_bwd_reg_kernel)
    z64, dh64 = z.double().view(scenes, N, 8, 64), dh.double().view(scenes, N, 8, 64)
    mu = z64.mean(dim=(1, 3), keepdim=True)
    rs = 1.0 / torch.sqrt(z64.var(dim=(1, 3), unbiased=False, keepdim=True) + 1e-5)
    xh = (z64 - mu) * rs
    ga, be = gamma.double().view(1, 1, 8, 64), beta.double().view(1, 1, 8, 64)
    s1 = 1.0 + (ss[:, :n].double().view(scenes, 1, 8, 64) if mode == 2 else 0.0)
    sh = ss[:, n:].double().view(scenes, 1, 8, 64) if mode == 2 else 0.0
    u = (ga * xh + be) * s1 + sh
    sig = torch.sigmoid(u)
    du = dh64 * (sig * (1 + u * (1 - sig)))
    dxh = du * s1 * ga
    want = rs * (dxh - dxh.mean(dim=(1, 3), keepdim=True) - xh * (dxh * xh).mean(dim=(1, 3), keepdim=True))
    assert rel(dz, want.view(M, n)) < 5e-6


def test_groupnorm_backward_epilogue_refuses_what_it_cannot_do():
    """A launch that sets gnb_z and does not qualify (scenes of <= 64 tokens, a per-slot (scale, shift), too few waves) fails -- it is never run without the epilogue."""
    from diffuscene_amd import ops
    n, K = 512, 512
    for scenes, N, mode in ((256, 21, 0), (256, 80, 3), (64, 80, 0)):
        M = scenes * N
        dy2, w, z = rnd(M, K, seed=1), rnd(n, K, seed=2, scale=0.05), rnd(M, n, seed=3)
        gamma, beta = rnd(n, seed=4) + 1.5, rnd(n, seed=5)
        ss = rnd(N, 2 * n, seed=6) if mode == 3 else None
        dz, part = torch.empty(M, n, device=dev()), torch.empty(scenes, 3 * n, device=dev())
        pp = part.data_ptr()
        g = ops.make_gemm_args(dy2, w, dz, gamma=gamma, beta=beta, tokens_per_scene=N, scale_shift=ss, ss_mode=mode,
                               gnb=dict(z=z, dgamma=pp + 4 * n, dbeta=pp + 8 * n, dbias=pp, pstride=3 * n, dss=None))
        assert ops.planes_layout(g) == -1
        with pytest.raises(RuntimeError, match="dsc_gemm_f32"):
            ops.run_gemm(g)
