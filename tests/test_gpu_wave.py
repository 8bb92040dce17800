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
