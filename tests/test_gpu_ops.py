"""GPU: every C-ABI kernel against a plain fp32/fp64 torch restatement of the same reference op (oracle pieces).
Tolerances are written per test; elementwise diffusion steps must be BIT-EXACT."""
import math
import os

import numpy as np

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import ref_torch as R  # noqa: E402


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + 1000 * len(shape) + sum(shape))
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


@pytest.mark.parametrize("m,n,k", [(1, 128, 32), (37, 5, 64), (160, 384, 512), (333, 25, 512), (256, 1024, 2048),
                                   (1000, 130, 96)])
def test_gemm_plain(m, n, k):
    from diffuscene_amd import ops
    a, w, b = rnd(m, k, seed=1), rnd(n, k, seed=2), rnd(n, seed=3)
    y = ops.gemm(a.to(dev()), w.to(dev()), b.to(dev()))
    ref = a.double() @ w.double().T + b.double()
    assert rel(y, ref) < 2e-6, (m, n, k, rel(y, ref))


def test_gemm_transpose_detecting_identity():
    """A = I-like asymmetric check: catches a swapped C-write or operand transpose."""
    from diffuscene_amd import ops
    m, n, k = 96, 64, 64
    a = torch.zeros(m, k)
    for i in range(m):
        a[i, (i * 7) % k] = 1.0 + i
    w = torch.arange(n * k, dtype=torch.float32).reshape(n, k) / 100.0
    y = ops.gemm(a.to(dev()), w.to(dev()))
    assert rel(y, a.double() @ w.double().T) < 1e-6


@pytest.mark.parametrize("act_in,act_out", [(0, 1), (0, 0), (0, 2)])
def test_gemm_epilogues_two_segments_residual(act_in, act_out):
    from diffuscene_amd import ops
    m, n = 200, 512
    a1, a2, w, b, r = rnd(m, 512, seed=4), rnd(m, 512, seed=5), rnd(n, 1024, seed=6, scale=0.05), rnd(n, seed=7), rnd(m, n, seed=8)
    y = ops.gemm(a1.to(dev()), w.to(dev()), b.to(dev()), a2=a2.to(dev()), residual=r.to(dev()), act_in=act_in, act_out=act_out)
    A = torch.cat([a1, a2], 1).double()
    if act_in == 2:
        A = F.silu(A)
    z = A @ w.double().T + b.double()
    z = F.gelu(z) if act_out == 1 else (F.silu(z) if act_out == 2 else z)
    assert rel(y, z + r.double()) < 3e-6


def test_gemm_strided_output_and_input_views():
    """decoder heads write at a column offset of the (M, C) output (ldy = 65, unaligned); A may be a column view."""
    from diffuscene_amd import ops
    m = 77
    abuf = rnd(m, 1024, seed=9).to(dev())
    a = abuf[:, 512:]
    w, b = rnd(25, 512, seed=10), rnd(25, seed=11)
    out = torch.zeros(m, 65, device=dev())
    ops.gemm(a, w.to(dev()), b.to(dev()), out=out[:, 8:33])
    ref = abuf[:, 512:].double().cpu() @ w.double().T + b.double()
    assert rel(out[:, 8:33], ref) < 2e-6
    assert float(out[:, :8].abs().max()) == 0 and float(out[:, 33:].abs().max()) == 0


@pytest.mark.parametrize("B,N,mode", [(3, 80, 2), (7, 21, 1), (9, 12, 3), (1, 33, 0), (2, 160, 2), (5, 50, 3)])
def test_gemm_groupnorm_silu_block(B, N, mode):
    """Block.forward (denoise_net.py:167-176): WS-conv output -> GroupNorm(8) -> (scale+1, shift) -> SiLU, + residual."""
    from diffuscene_amd import ops
    M, D = B * N, 512
    a, w, b = rnd(M, D, seed=12), rnd(D, D, seed=13, scale=0.1), rnd(D, seed=14)
    gamma, beta = 1 + 0.1 * rnd(D, seed=15), 0.1 * rnd(D, seed=16)
    res = rnd(M, D, seed=17)
    rows = {0: 0, 1: M, 2: B, 3: N}[mode]
    ss = rnd(rows, 2 * D, seed=18) if rows else None
    y = ops.gemm_gn_silu(a.to(dev()), w.to(dev()), b.to(dev()), gamma.to(dev()), beta.to(dev()), N,
                         scale_shift=ss.to(dev()) if ss is not None else None, ss_mode=mode, residual=res.to(dev()))
    z = (a.double() @ w.double().T + b.double()).reshape(B, N, D).permute(0, 2, 1)        # (B, C, N)
    g = F.group_norm(z, 8, gamma.double(), beta.double(), eps=1e-5)
    if mode:
        if mode == 1:
            e = ss.reshape(B, N, 2 * D)
        elif mode == 2:
            e = ss[:, None, :].expand(B, N, 2 * D)
        else:
            e = ss[None].expand(B, N, 2 * D)
        e = e.double().permute(0, 2, 1)
        g = g * (e[:, :D] + 1) + e[:, D:]
    ref = F.silu(g).permute(0, 2, 1).reshape(M, D) + res.double()
    assert rel(y, ref) < 5e-6, rel(y, ref)


def test_gemm_gn_two_segments():
    from diffuscene_amd import ops
    B, N, D = 4, 21, 512
    M = B * N
    a1, a2, w, b = rnd(M, D, seed=19), rnd(M, D, seed=20), rnd(D, 2 * D, seed=21, scale=0.1), rnd(D, seed=22)
    gamma, beta = 1 + 0.1 * rnd(D, seed=23), 0.1 * rnd(D, seed=24)
    y = ops.gemm_gn_silu(a1.to(dev()), w.to(dev()), b.to(dev()), gamma.to(dev()), beta.to(dev()), N, a2=a2.to(dev()))
    z = (torch.cat([a1, a2], 1).double() @ w.double().T + b.double()).reshape(B, N, D).permute(0, 2, 1)
    ref = F.silu(F.group_norm(z, 8, gamma.double(), beta.double(), eps=1e-5)).permute(0, 2, 1).reshape(M, D)
    assert rel(y, ref) < 5e-6


def test_weight_standardize():
    from diffuscene_amd import ops
    ws = [rnd(512, 512, seed=30), rnd(512, 1024, seed=31) + 0.3, rnd(64, 40, seed=32)]
    outs = ops.weight_standardize([w.to(dev()) for w in ws])
    for w, o in zip(ws, outs):
        w3 = w.double()[:, :, None]
        mean = w3.mean(dim=(1, 2), keepdim=True)
        var = w3.var(dim=(1, 2), unbiased=False, keepdim=True)
        ref = ((w3 - mean) * (var + 1e-5).rsqrt())[:, :, 0]
        assert rel(o, ref) < 2e-6


def test_layernorm_with_residual():
    from diffuscene_amd import ops
    x, g, r = rnd(301, 512, seed=33) * 3 + 0.5, 1 + 0.1 * rnd(512, seed=34), rnd(301, 512, seed=35)
    y = ops.layernorm(x.to(dev()), g.to(dev()), residual=r.to(dev()))
    xd = x.double()
    ref = (xd - xd.mean(1, keepdim=True)) * (xd.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt() * g.double() + r.double()
    assert rel(y, ref) < 2e-6
    y2 = ops.layernorm(x.to(dev()), g.to(dev()))
    assert rel(y2, ref - r.double()) < 2e-6


@pytest.mark.parametrize("B,N", [(3, 80), (5, 12), (2, 21), (1, 160)])
def test_linear_attention(B, N):
    from diffuscene_amd import ops
    qkv = rnd(B * N, 384, seed=36) * 2
    out = ops.linear_attention(qkv.to(dev())[:, :128], qkv.to(dev())[:, 128:256], qkv.to(dev())[:, 256:], B, N, N, 32 ** -0.5)
    t = qkv.double().reshape(B, N, 3, 4, 32).permute(2, 0, 3, 4, 1)          # (3, B, h, c, n)
    ref = R._linear_attention_core(t[0], t[1], t[2])                          # (B, 128, N)
    assert rel(out, ref.permute(0, 2, 1).reshape(B * N, 128)) < 3e-6


def test_linear_attention_cross():
    from diffuscene_amd import ops
    B, N, L = 3, 12, 7
    q, kv = rnd(B * N, 128, seed=37) * 2, rnd(B * L, 256, seed=38) * 2
    out = ops.linear_attention(q.to(dev()), kv.to(dev())[:, :128], kv.to(dev())[:, 128:], B, N, L, 32 ** -0.5)
    qh = q.double().reshape(B, N, 4, 32).permute(0, 2, 3, 1)
    kh = kv.double()[:, :128].reshape(B, L, 4, 32).permute(0, 2, 3, 1)
    vh = kv.double()[:, 128:].reshape(B, L, 4, 32).permute(0, 2, 3, 1)
    ref = R._linear_attention_core(qh, kh, vh).permute(0, 2, 1).reshape(B * N, 128)
    assert rel(out, ref) < 3e-6


@pytest.mark.parametrize("B,N", [(3, 80), (4, 12), (1, 160), (2, 70), (2, 96), (1, 113)])   # 320 / 256 / 512 (two passes) / 320 ragged / 384 / 512 threads
def test_softmax_attention(B, N):
    from diffuscene_amd import ops
    qkv = rnd(B * N, 384, seed=39) * 2
    d = qkv.to(dev())
    out = ops.attention(d[:, :128], d[:, 128:256], d[:, 256:], B, N, 32 ** -0.5)
    t = qkv.double().reshape(B, N, 3, 4, 32).permute(2, 0, 3, 4, 1)          # (3, B, h, d, n)
    q, k, v = t[0] * 32 ** -0.5, t[1], t[2]
    attn = torch.einsum("bhdi,bhdj->bhij", q, k).softmax(-1)
    ref = torch.einsum("bhij,bhdj->bhid", attn, v).permute(0, 2, 1, 3).reshape(B * N, 128)
    assert rel(out, ref) < 3e-6


def test_linear_smallk_on_unaligned_slices():
    from diffuscene_amd import ops
    M = 77
    x = rnd(M, 65, seed=40).to(dev())
    for c0, k, act in ((0, 8, 1), (8, 25, 1), (33, 32, 1), (0, 5, 0)):
        w, b = rnd(512, k, seed=41 + k), rnd(512, seed=42 + k)
        y = ops.linear_smallk(x[:, c0:c0 + k], w.to(dev()), b.to(dev()), act_out=act)
        z = x[:, c0:c0 + k].double().cpu() @ w.double().T + b.double()
        assert rel(y, F.gelu(z) if act else z) < 2e-6


def test_time_embedding_table_is_bit_exact_and_fallback_close():
    from diffuscene_amd import ops
    half = 256
    freq = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
    arg = torch.arange(1000)[:, None] * freq[None, :]
    table = torch.cat((arg.sin(), arg.cos()), -1)
    t = torch.tensor([0, 1, 17, 999, 1500], dtype=torch.int64)
    out = ops.time_embedding(t.to(dev()), 512, table.to(dev()), freq.to(dev())).cpu()
    ref = R.sinusoidal_embedding(t, 512)
    assert torch.equal(out[:4], ref[:4])
    assert float((out[4] - ref[4]).abs().max()) < 5e-4       # beyond the table: device sinf/cosf of the same fp32 argument


def test_diffusion_elementwise_bit_exact():
    from diffuscene_amd import ops
    tb = R.schedule_tables(1e-4, 0.02, 1000, "v")
    B, N, C = 6, 21, 65
    x0, noise, v = rnd(B, N, C, seed=50), torch.randn(B, N, C, generator=torch.Generator().manual_seed(1)), rnd(B, N, C, seed=51) * 2
    t = torch.tensor([0, 1, 500, 998, 999, 0], dtype=torch.int64)
    d = {k: tb[k].to(dev()) for k in tb}
    sigma = torch.exp(0.5 * tb["posterior_log_variance_clipped"]).to(dev())
    xt, vt = ops.q_sample(x0.to(dev()), noise.to(dev()), t.to(dev()), d["sqrt_alphas_cumprod"],
                          d["sqrt_one_minus_alphas_cumprod"], want_v=True)
    assert torch.equal(xt.cpu(), R.q_sample(tb, x0, t, noise))
    assert torch.equal(vt.cpu(), R.predict_v(tb, x0, t, noise))
    for clip in (True, False):
        for mt, name, ca, cb in ((2, "v", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod"),
                                 (0, "eps", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod"), (1, "x0", None, None)):
            out = ops.p_sample(x0.to(dev()), v.to(dev()), noise.to(dev()), t.to(dev()), d[ca] if ca else None,
                               d[cb] if cb else None, d["posterior_mean_coef1"], d["posterior_mean_coef2"], sigma, mt, clip)
            ref = R.p_sample_step(tb, x0, t, v, noise, clip, name)
            assert torch.equal(out.cpu(), ref), (name, clip, float((out.cpu() - ref).abs().max()))
    x = rnd(B, N, C, seed=52).to(dev())
    part, pn = rnd(B, 4, C, seed=53), rnd(B, 4, C, seed=54)
    keep = x.clone()
    ops.complete_overwrite(x, part.to(dev()), pn.to(dev()), t.to(dev()), d["sqrt_alphas_cumprod"], d["sqrt_one_minus_alphas_cumprod"])
    assert torch.equal(x[:, :4].cpu(), R.q_sample(tb, part, t, pn)) and torch.equal(x[:, 4:], keep[:, 4:])
    tt = t.to(dev()).clone()
    ops.add_scalar_i64(tt, -1)
    assert torch.equal(tt.cpu(), t - 1)


def test_bad_arguments_are_rejected_not_clamped():
    from diffuscene_amd import ops
    a, w = torch.zeros(8, 48, device=dev()), torch.zeros(8, 48, device=dev())
    with pytest.raises(RuntimeError, match="DSC_EINVAL"):
        ops.gemm(a, w)                                    # K not a multiple of 32
    x = torch.zeros(2 * 200, 512, device=dev())
    wz = torch.zeros(512, 512, device=dev())
    v = torch.zeros(512, device=dev())
    with pytest.raises(RuntimeError, match="DSC_ERANGE"):
        ops.gemm_gn_silu(x, wz, v, v, v, 200)            # more than 160 objects per scene
    # weight planes: the split needs 8-element output rows; wrong plane shapes are caught before the launch
    with pytest.raises(RuntimeError, match="DSC_EINVAL"):
        ops.split_planes([(torch.zeros(64, 36, device=dev()), None, False)])
    with pytest.raises(RuntimeError, match="w_planes"):
        ops.make_gemm_args(x, wz, torch.zeros(400, 512, device=dev()), w_planes=torch.zeros(3, 512, 256, device=dev(), dtype=torch.int16))


@pytest.mark.gpu
def test_postfilter_matches_reference_method(golden_dir):
    """Device post-filter vs the REAL reference delete_empty_from_network_samples (tests/golden/postfilter.npz, produced per
    scene because the reference method only runs for batch_size 1): per-scene mode == the reference on every scene alone;
    the drop-in method == the reference for B = 1 and applies row 0's decision to a larger batch; exact equality."""
    import types
    from oracle.make_golden_postfilter import CASES, case_samples
    from diffuscene_amd import ops
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import DiffusionSceneLayout_DDPM as M
    g = np.load(os.path.join(golden_dir, "postfilter.npz"))
    for name in CASES:
        x, nc, nf = case_samples(name)
        me = types.SimpleNamespace(translation_dim=3, size_dim=3, angle_dim=2, bbox_dim=8, class_dim=nc, objfeat_dim=nf,
                                   n_classes=nc + 1)
        me._split_boxes = types.MethodType(M._split_boxes, me)
        me._keep_rows = types.MethodType(M._keep_rows, me)
        xd = x.to(dev())
        for keep in (False, True):
            per = M.delete_empty_per_scene(me, xd, keep_empty=keep)
            assert len(per) == x.shape[0]
            for b, d in enumerate(per):
                for k, v in d.items():
                    ref = torch.from_numpy(g["%s.keep%d.scene%d.%s" % (name, int(keep), b, k)])
                    assert v.shape == ref.shape and torch.equal(v, ref), (name, keep, b, k)
            whole = M.delete_empty_from_network_samples(me, xd, keep_empty=keep)
            host = M.delete_empty_from_network_samples(me, x, keep_empty=keep)           # CPU tensors: the host path
            for k, v in whole.items():
                ref0 = torch.from_numpy(g["%s.keep%d.scene0.%s" % (name, int(keep), k)])
                assert torch.equal(v[0:1], ref0) and torch.equal(v, host[k]), (name, keep, k)
    # packed layout: kept rows first, zero tail, counts
    x, nc, nf = case_samples("bedroom_b3")
    packed, counts = ops.postfilter_compact(x.to(dev()), 8 + nc - 1, per_scene=True)
    for b in range(x.shape[0]):
        keep = x[b, :, 8 + nc - 1] < 0
        assert int(counts[b]) == int(keep.sum())
        assert torch.equal(packed[b, :int(counts[b])].cpu(), x[b][keep]) and float(packed[b, int(counts[b]):].abs().sum()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("gn", [False, True])
def test_gemm_two_segments_with_different_row_strides(gn):
    """[a | a2] @ W^T where a2 is a column slice of a wider tensor (lda2 != lda1): the LDS-DMA main loop needs one row stride,
    so this shape takes the register-staged interleaved loop (IL = 1); results must equal the dense case bit for bit (same
    arithmetic order) and the fp64 product within 5e-6."""
    from diffuscene_amd import ops
    torch.manual_seed(0)
    B, N, D = 3, 80, 512
    M = B * N
    a = torch.randn(M, 512, device=dev())
    wide = torch.randn(M, 1024, device=dev())
    a2v = wide[:, 256:768]                                   # stride 1024, 16-byte aligned start
    a2d = a2v.contiguous()
    w = torch.randn(D, 1024, device=dev()) * 0.05
    b = torch.randn(D, device=dev())
    if gn:
        gamma, beta = torch.rand(D, device=dev()) + 0.5, torch.randn(D, device=dev()) * 0.1
        y1 = ops.gemm_gn_silu(a, w, b, gamma, beta, N, a2=a2v)
        y2 = ops.gemm_gn_silu(a, w, b, gamma, beta, N, a2=a2d)
        z = (torch.cat([a, a2d], 1).double() @ w.double().T + b.double()).reshape(B, N, D).permute(0, 2, 1)
        ref = F.silu(F.group_norm(z, 8, gamma.double(), beta.double(), eps=1e-5)).permute(0, 2, 1).reshape(M, D)
    else:
        y1 = ops.gemm(a, w, b, a2=a2v)
        y2 = ops.gemm(a, w, b, a2=a2d)
        ref = torch.cat([a, a2d], 1).double() @ w.double().T + b.double()
    assert torch.equal(y1, y2)
    assert float((y1.double() - ref).abs().max() / ref.abs().max()) < 5e-6


@pytest.mark.gpu
def test_splitk_gemm_and_grouped_colsum():
    """dsc_gemm_splitk_f32 (short, deep products: K cut over the batch dimension + fixed-order slab sum with bias / residual)
    and dsc_colsum_grouped_f32 vs fp64."""
    import ctypes as C
    import numpy as np
    from diffuscene_amd import _lib, ops
    torch.manual_seed(1)
    for m, n, K, splits, use_res in ((256, 2048, 2048, 8, True), (80, 128, 2048, 8, False), (256, 2048, 1024, 4, True)):
        a = torch.randn(m, K, device=dev())
        w = torch.randn(n, K, device=dev()) / K ** 0.5
        b = torch.randn(n, device=dev())
        y = torch.randn(m, n, device=dev())
        y0 = y.clone()
        g = ops.make_gemm_args(a, w, y, b, None, y if use_res else None)
        ws = torch.empty(splits * m * n, device=dev())
        _lib.check(_lib.fn("dsc_gemm_splitk_f32")(C.byref(g), splits, ws.data_ptr(), ws.numel(), ops.stream_ptr()), "splitk")
        ref = a.double() @ w.double().T + b.double() + (y0.double() if use_res else 0)
        assert float((y.double() - ref).abs().max() / ref.abs().max()) < 5e-6
    mats = [torch.randn(256, 1536, device=dev()), torch.randn(512, 512, device=dev()), torch.randn(37, 100, device=dev())]
    outs = [torch.empty(x.shape[1], device=dev()) for x in mats]
    arr = (_lib.ColsumItem * len(mats))()
    for i, (x, o) in enumerate(zip(mats, outs)):
        arr[i].x, arr[i].ldx, arr[i].m, arr[i].n, arr[i].out = x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], o.data_ptr()
    table = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev())
    _lib.check(_lib.fn("dsc_colsum_grouped_f32")(table.data_ptr(), len(mats), 1536, ops.stream_ptr()), "colsum_grouped")
    for x, o in zip(mats, outs):
        ref = x.double().sum(0)
        assert float((o.double() - ref).abs().max()) < 1e-4 * float(ref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("m,k,res", [(20480, 128, True), (20000, 128, False), (100, 512, True)])
def test_gemm_with_layernorm_epilogue(m, k, res):
    """dsc_gemm_layernorm_f32: y = LayerNorm_channels(a @ W^T + b) * g (+ residual) (denoise_net.py:93-102 after :216-219) vs the
    fp64 torch evaluation; 5e-6 relative to the output range."""
    import ctypes as C
    from diffuscene_amd import _lib, ops
    torch.manual_seed(2)
    a = torch.randn(m, k, device=dev())
    w = torch.randn(512, k, device=dev()) / k ** 0.5
    b = torch.randn(512, device=dev()) * 0.1
    g = 1 + 0.1 * torch.randn(512, device=dev())
    r = torch.randn(m, 512, device=dev()) if res else None
    y = torch.empty(m, 512, device=dev())
    args = ops.make_gemm_args(a, w, y, b, None, r, gamma=g, beta=g, eps=1e-5)
    _lib.check(_lib.fn("dsc_gemm_layernorm_f32")(C.byref(args), ops.stream_ptr()), "dsc_gemm_layernorm_f32")
    o = a.double() @ w.double().T + b.double()
    mu = o.mean(1, keepdim=True)
    var = o.var(1, unbiased=False, keepdim=True)
    ref = (o - mu) * torch.rsqrt(var + 1e-5) * g.double() + (r.double() if res else 0)
    assert float((y.double() - ref).abs().max() / ref.abs().max()) < 5e-6


def test_grouped_smallk_and_column_gather_match_single_launches():
    """Round 4 (sampling tail): the first layers of the three per-attribute encoders in ONE launch, and the gather that moves the padded
    outputs of the stacked decoder heads into the (M, C) scene tensor -- bit-identical to the single launches / to a slice copy."""
    from diffuscene_amd import ops
    d = dev()
    M, D = 20480 + 7, 512
    x = (torch.rand(M, 65, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(d)
    heads = [(8, 25), (0, 8), (33, 32)]                                     # (first column, width) of class / bbox / objfeat
    ws = [(torch.rand(D, k, 1, generator=torch.Generator().manual_seed(2 + k)) - 0.5).to(d) for _, k in heads]
    bs = [(torch.rand(D, generator=torch.Generator().manual_seed(9 + k)) - 0.5).to(d) for _, k in heads]
    wide = torch.full((M, 3 * D), float("nan"), device=d)
    ops.linear_smallk_grouped([x[:, c0:c0 + k] for c0, k in heads], ws, bs, [wide[:, i * D:(i + 1) * D] for i in range(3)], act_out=1)
    for i, (c0, k) in enumerate(heads):
        single = ops.linear_smallk(x[:, c0:c0 + k], ws[i], bs[i], act_out=1)
        assert torch.equal(wide[:, i * D:(i + 1) * D], single), i
    src = torch.rand(M, 96, device=d)
    dst = torch.full((M, 65), float("nan"), device=d)
    ops.gather_columns(dst, src, [(0, 0, 8), (32, 8, 25), (64, 33, 32)])
    want = torch.cat([src[:, 0:8], src[:, 32:57], src[:, 64:96]], dim=1)
    assert torch.equal(dst, want)
