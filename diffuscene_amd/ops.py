"""Tensor-level wrappers over the C ABI (include/diffuscene_hip.h).

PyTorch is used for device memory and streams only: every function takes CUDA(HIP) fp32 tensors,
passes raw device pointers + the current torch stream to libdiffuscene_hip.so and returns the output
tensor.  There is no CPU path: a CPU tensor raises.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_NONE, ACT_SILU, MEAN_EPS, MEAN_V, MEAN_X0, SS_NONE, SS_PER_SCENE,  # noqa: F401
                   SS_PER_SLOT, SS_PER_TOKEN, GemmArgs, SplitItem, WsItem)


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def _dev(t, name="tensor", dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("diffuscene_amd: %s must live on a HIP device (no CPU fallback exists)" % name)
    if t.dtype != dtype:
        raise RuntimeError("diffuscene_amd: %s must be %s, got %s" % (name, dtype, t.dtype))
    return t


def _mat(t, name):
    """2-D row-major view with contiguous columns; returns (ptr, ld)."""
    _dev(t, name)
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise RuntimeError("diffuscene_amd: %s must be a 2-D tensor with contiguous rows, got shape %s strides %s"
                           % (name, tuple(t.shape), t.stride()))
    return t.data_ptr(), (t.stride(0) if t.shape[0] > 1 else t.shape[1])


def as2d(w):
    """(out, in, 1) conv weight or (out, in) linear weight -> (out, in) view."""
    return w.view(w.shape[0], w.shape[1]) if w.dim() == 3 else w


def make_gemm_args(a, w, y, bias=None, a2=None, residual=None, act_in=ACT_NONE, act_out=ACT_NONE,
                   gamma=None, beta=None, eps=1e-5, tokens_per_scene=0, scale_shift=None, ss_mode=SS_NONE, preact=None,
                   ss_index=None, w_planes=None, actgrad_x=None, gnb=None, row_invariant=False):
    """Build a dsc_gemm_args for  y = epi(act_in([a|a2]) @ w.T + bias).  The returned struct holds raw
    pointers only; the caller keeps the tensors alive.  ``w_planes``: the weight pre-split into bf16 planes (split_planes):
    the product runs on the bf16 matrix cores with f32 accuracy where the kernel supports the shape."""
    g = GemmArgs()
    g.a1, g.lda1 = _mat(a, "a")
    g.k1 = a.shape[1]
    if a2 is not None:
        g.a2, g.lda2 = _mat(a2, "a2")
        g.k2 = a2.shape[1]
        if a2.shape[0] != a.shape[0]:
            raise RuntimeError("a and a2 row mismatch")
    w2 = as2d(w)
    g.w, g.ldw = _mat(w2, "w")
    if w2.shape[1] != g.k1 + g.k2:
        raise RuntimeError("weight K (%d) != activation K (%d)" % (w2.shape[1], g.k1 + g.k2))
    g.m, g.n = a.shape[0], w2.shape[0]
    g.y, g.ldy = _mat(y, "y")
    if tuple(y.shape) != (g.m, g.n):
        raise RuntimeError("output shape %s != (%d, %d)" % (tuple(y.shape), g.m, g.n))
    if bias is not None:
        g.bias = _dev(bias, "bias").data_ptr()
    if residual is not None:
        g.residual, g.ldr = _mat(residual, "residual")
    g.act_in, g.act_out, g.batch = act_in, act_out, 1
    if row_invariant:                      # include/diffuscene_hip.h DSC_GEMM_ROW_INVARIANT: a row's result must not depend on the number of rows
        g.flags |= _lib.GEMM_ROW_INVARIANT
    if gamma is not None:
        g.gamma, g.beta, g.eps = _dev(gamma).data_ptr(), _dev(beta).data_ptr(), eps
        g.tokens_per_scene = tokens_per_scene
    if scale_shift is not None:
        g.scale_shift, g.ld_ss = _mat(scale_shift, "scale_shift")
        g.ss_mode = ss_mode
    if preact is not None:
        g.preact, g.ld_preact = _mat(preact, "preact")
    if actgrad_x is not None:              # y = product * act_out'(actgrad_x) (split kernel only; include/diffuscene_hip.h)
        g.actgrad_x, g.ld_actgrad = _mat(actgrad_x, "actgrad_x")
        if tuple(actgrad_x.shape) != (g.m, g.n):
            raise RuntimeError("actgrad_x shape %s != (%d, %d)" % (tuple(actgrad_x.shape), g.m, g.n))
    if gnb is not None:
        # GroupNorm-backward epilogue (include/diffuscene_hip.h, gnb_*): dict(z=, dgamma=, dbeta=, dbias=, pstride=, dss=None); gamma / beta / eps /
        # tokens_per_scene / scale_shift / ss_mode above describe the Block whose output gradient this product is
        g.gnb_z, g.ld_gnb_z = _mat(gnb["z"], "gnb z")
        g.gnb_dgamma, g.gnb_dbeta, g.gnb_dbias = gnb["dgamma"], gnb["dbeta"], gnb["dbias"]
        g.gnb_pstride = gnb["pstride"]
        if gnb.get("dss") is not None:
            g.gnb_dss, g.ld_gnb_dss = _mat(gnb["dss"], "gnb dss")
    if ss_index is not None:
        if scale_shift is None:
            raise RuntimeError("ss_index (DSC_SS_BY_INDEX) needs the scale_shift table it indexes")
        g.ss_index = _dev(ss_index, "ss_index", torch.int64).data_ptr()
        g.ss_rows = scale_shift.shape[0]              # the device clamps every gathered row into the table
    if w_planes is not None:
        # (3, n, K); a grouped launch (batch set by the caller afterwards) passes the planes of the stacked weights (3, batch n, K)
        if (w_planes.dtype != torch.int16 or w_planes.dim() != 3 or w_planes.shape[0] != 3 or w_planes.shape[1] % g.n
                or w_planes.shape[2] != g.k1 + g.k2 or not w_planes.is_contiguous()):
            raise RuntimeError("w_planes must be a contiguous int16 tensor of shape (3, %d, %d)" % (g.n, g.k1 + g.k2))
        attach_planes(g, w_planes, gn=gamma is not None and gnb is None)       # (gnb: a dense product that borrows the Block's gamma / beta)
    return g


PLANES_ROWMAJOR, PLANES_FRAGMENT = 0, 1


def planes_layout(g, gn=False):
    """Which layout of the weight's bf16 planes this launch wants (dsc_gemm_planes_layout): PLANES_ROWMAJOR (the block-staged split
    kernel), PLANES_FRAGMENT (the wave-autonomous kernel reads MFMA fragments straight from memory), or -1: the launch stays on the
    exact-f32 kernel whatever it is given.  Asked BEFORE planes are made."""
    r = _lib.fn("dsc_gemm_planes_layout")(C.byref(g), 1 if gn else 0)
    if r < -1:
        _lib.check(r, "dsc_gemm_planes_layout")
    return r


def fragment_major(planes):
    """Row-major planes (3, n, K) -> the same values fragment-major, [3][n/16][K/32][lane = (k-octet & 3) * 16 + row % 16][8] (kept in a
    tensor of the same shape)."""
    _, n, k = planes.shape
    return planes.view(3, n // 16, 16, k // 32, 4, 8).permute(0, 1, 3, 4, 2, 5).contiguous().view(3, n, k)


def attach_planes(g, planes, gn=False, layout=None):
    """Hand ``planes`` to the launch.  ``layout`` says how they are laid out (what split_planes(..., fragment=) made); None = row-major
    planes from an ad-hoc caller (tests, tools): where the library wants the other layout for this launch a converted copy is made
    once and kept on the tensor -- plans ask planes_layout() first and split straight into the layout they need."""
    want = planes_layout(g, gn)
    if layout is None:
        layout = getattr(planes, "_dsc_layout", PLANES_ROWMAJOR)
        if want == PLANES_FRAGMENT and layout == PLANES_ROWMAJOR:
            cached = getattr(planes, "_dsc_fragment", None)
            if cached is None or cached[0] != planes._version:
                cached = planes._dsc_fragment = (planes._version, fragment_major(planes))
            planes, layout = cached[1], PLANES_FRAGMENT
            g._planes_keep = planes                 # the struct holds a raw pointer: keep the converted copy with it
    g.w_planes = planes.data_ptr()
    g.w_planes_layout = layout
    return g


def planes_wanted(n, k):
    """Shapes the split-bf16 GEMM path covers (gemm_split.hip): callers skip the plane copy for everything else."""
    return n % 128 == 0 and k % 32 == 0 and 6 * n * k < 0x7fffffff


def split_planes(items, stream=None):
    """Exact 3-way bf16 split of f32 matrices (dsc_split_bf16x3_f32).  items: [(w2d, planes, transpose)] with ``planes`` an int16
    tensor (3, rows, cols) -- or (3, cols, rows) when ``transpose`` -- or None (allocated).  Returns the planes tensors."""
    out, batch = [], []
    for w, planes, tr in items:
        tr = int(tr)                       # bit 0: planes of w^T; bit 1 (2): fragment-major output (PLANES_FRAGMENT)
        frag, tr = tr & 2, tr & 1
        w2 = as2d(_dev(w, "w"))
        ptr, ldw = _mat(w2, "w")
        r, c = w2.shape
        shape = (3, c, r) if tr else (3, r, c)
        if planes is None:
            planes = torch.empty(shape, device=w2.device, dtype=torch.int16)
        elif tuple(planes.shape) != shape or planes.dtype != torch.int16 or not planes.is_contiguous():
            raise RuntimeError("split_planes: planes must be contiguous int16 %s" % (shape,))
        planes._dsc_layout = PLANES_FRAGMENT if frag else PLANES_ROWMAJOR
        out.append(planes)
        batch.append((ptr, ldw, r, c, planes.data_ptr(), (1 if tr else 0) | frag, w2, planes))
    for i in range(0, len(batch), _lib.WS_MAX):
        part = batch[i:i + _lib.WS_MAX]
        arr = (SplitItem * len(part))()
        for j, (ptr, ldw, r, c, pp, tr, _, _) in enumerate(part):
            arr[j].w, arr[j].ldw, arr[j].rows, arr[j].cols, arr[j].planes, arr[j].transpose = ptr, ldw, r, c, pp, tr
        _lib.check(_lib.fn("dsc_split_bf16x3_f32")(arr, len(part), stream if stream is not None else stream_ptr()),
                   "dsc_split_bf16x3_f32")
    return out


def gemm_uses_split(g, gn=False):
    """True when the library would run this launch on the split-bf16 kernel (its own dispatch decision, dsc_gemm_arithmetic)."""
    return _lib.fn("dsc_gemm_arithmetic")(C.byref(g), 1 if gn else 0) == 1


def gemm_would_use_split(g, gn=False):
    """The same question BEFORE any planes exist: would this launch take the split path if the weight's planes were supplied?
    (Callers skip the plane copy -- 6 bytes per weight and a split launch per weight update -- for launches that stay on the
    exact-f32 kernel anyway: too few blocks, unsupported shape, DSC_GEMM=f32.)"""
    return planes_layout(g, gn) >= 0


def run_gemm(g, gn=False, stream=None):
    name = "dsc_gemm_gn_silu_f32" if gn else "dsc_gemm_f32"
    _lib.check(_lib.fn(name)(C.byref(g), stream if stream is not None else stream_ptr()), name)


def gemm(a, w, bias=None, a2=None, residual=None, act_in=ACT_NONE, act_out=ACT_NONE, out=None, w_planes=None, row_invariant=False):
    w2 = as2d(w)
    if out is None:
        out = torch.empty((a.shape[0], w2.shape[0]), device=a.device, dtype=torch.float32)
    run_gemm(make_gemm_args(a, w, out, bias, a2, residual, act_in, act_out, w_planes=w_planes, row_invariant=row_invariant))
    return out


def gemm_gn_silu(a, w_std, bias, gamma, beta, tokens_per_scene, a2=None, scale_shift=None, ss_mode=SS_NONE,
                 residual=None, eps=1e-5, out=None, preact=None, w_planes=None, ss_index=None):
    w2 = as2d(w_std)
    if out is None:
        out = torch.empty((a.shape[0], w2.shape[0]), device=a.device, dtype=torch.float32)
    run_gemm(make_gemm_args(a, w_std, out, bias, a2, residual, gamma=gamma, beta=beta, eps=eps,
                            tokens_per_scene=tokens_per_scene, scale_shift=scale_shift, ss_mode=ss_mode,
                            preact=preact, w_planes=w_planes, ss_index=ss_index), gn=True)
    return out


def linear_smallk(x, w, bias, act_out=ACT_NONE, out=None):
    """x may be a column slice of a wider row-major tensor (arbitrary alignment)."""
    xp, ldx = _mat(x, "x")
    w2 = as2d(w)
    wp, ldw = _mat(w2, "w")
    if out is None:
        out = torch.empty((x.shape[0], w2.shape[0]), device=x.device, dtype=torch.float32)
    yp, ldy = _mat(out, "y")
    _lib.check(_lib.fn("dsc_linear_smallk_f32")(xp, ldx, x.shape[1], wp, ldw,
                                                bias.data_ptr() if bias is not None else None,
                                                yp, ldy, x.shape[0], w2.shape[0], act_out, stream_ptr()),
               "dsc_linear_smallk_f32")
    return out


def make_ws_items(pairs):
    """pairs: list of (w2d, out2d) contiguous tensors -> ctypes array."""
    arr = (WsItem * len(pairs))()
    for i, (w, o) in enumerate(pairs):
        w2, o2 = as2d(_dev(w)), as2d(_dev(o))
        if not (w2.is_contiguous() and o2.is_contiguous()):
            raise RuntimeError("weight_standardize needs contiguous matrices")
        arr[i].w, arr[i].out, arr[i].rows, arr[i].cols = w2.data_ptr(), o2.data_ptr(), w2.shape[0], w2.shape[1]
    return arr


def weight_standardize(weights, outs=None, eps=1e-5):
    """Batched (w - mean_row) * rsqrt(var_row + eps); returns the standardised copies."""
    if outs is None:
        outs = [torch.empty_like(as2d(w)) for w in weights]
    for i in range(0, len(weights), _lib.WS_MAX):
        chunk = list(zip(weights[i:i + _lib.WS_MAX], outs[i:i + _lib.WS_MAX]))
        arr = make_ws_items(chunk)
        _lib.check(_lib.fn("dsc_weight_standardize_f32")(arr, len(chunk), eps, stream_ptr()),
                   "dsc_weight_standardize_f32")
    return outs


def layernorm(x, g, residual=None, eps=1e-5, out=None):
    xp, ldx = _mat(x, "x")
    if out is None:
        out = torch.empty((x.shape[0], x.shape[1]), device=x.device, dtype=torch.float32)
    yp, ldy = _mat(out, "y")
    rp, ldr = _mat(residual, "residual") if residual is not None else (None, 0)
    _lib.check(_lib.fn("dsc_layernorm_f32")(xp, ldx, _dev(g).data_ptr(), rp, ldr, yp, ldy, x.shape[0], x.shape[1],
                                            eps, stream_ptr()), "dsc_layernorm_f32")
    return out


def linear_attention(q, k, v, scenes, nq, nk, scale, out=None):
    qp, ldq = _mat(q, "q")
    kp, ldk = _mat(k, "k")
    vp, ldv = _mat(v, "v")
    if out is None:
        out = torch.empty((q.shape[0], 128), device=q.device, dtype=torch.float32)
    op, ldo = _mat(out, "out")
    _lib.check(_lib.fn("dsc_linear_attention_f32")(qp, ldq, kp, ldk, vp, ldv, op, ldo, scenes, nq, nk, scale,
                                                   stream_ptr()), "dsc_linear_attention_f32")
    return out


def attention(q, k, v, scenes, n, scale, out=None):
    qp, ldq = _mat(q, "q")
    kp, ldk = _mat(k, "k")
    vp, ldv = _mat(v, "v")
    if out is None:
        out = torch.empty((q.shape[0], 128), device=q.device, dtype=torch.float32)
    op, ldo = _mat(out, "out")
    _lib.check(_lib.fn("dsc_attention_f32")(qp, ldq, kp, ldk, vp, ldv, op, ldo, scenes, n, scale, stream_ptr()),
               "dsc_attention_f32")
    return out


def time_embedding(t, dim, table, freq, out=None):
    _dev(t, "t", torch.int64)
    if out is None:
        out = torch.empty((t.shape[0], dim), device=t.device, dtype=torch.float32)
    _lib.check(_lib.fn("dsc_time_embedding_f32")(t.data_ptr(), t.shape[0], dim,
                                                 table.data_ptr() if table is not None else None,
                                                 table.shape[0] if table is not None else 0,
                                                 _dev(freq).data_ptr(), out.data_ptr(), stream_ptr()),
               "dsc_time_embedding_f32")
    return out


def linear_smallk_grouped(xs, ws, biases, outs, act_out=ACT_NONE):
    """``len(xs)`` <= 4 small-K linears of one output width in ONE launch (dsc_linear_smallk_grouped_f32): xs[i] [m, k_i] may be column
    slices of a wider tensor, outs[i] [m, n] column blocks of a wider buffer."""
    n_items = len(xs)
    items = (_lib.SmallKItem * n_items)()
    m, n = outs[0].shape
    for i in range(n_items):
        w2 = as2d(ws[i])
        items[i].x, items[i].ldx = _mat(xs[i], "x")
        items[i].k_in = xs[i].shape[1]
        items[i].w, items[i].ldw = _mat(w2, "w")
        items[i].bias = biases[i].data_ptr() if biases[i] is not None else None
        items[i].y, items[i].ldy = _mat(outs[i], "y")
        if tuple(outs[i].shape) != (m, n) or w2.shape[0] != n or xs[i].shape[0] != m:
            raise RuntimeError("linear_smallk_grouped: every head must produce the same [m, n] block")
    _lib.check(_lib.fn("dsc_linear_smallk_grouped_f32")(items, n_items, m, n, act_out, stream_ptr()), "dsc_linear_smallk_grouped_f32")
    return outs


def gather_columns(dst, src, spans):
    """dst[:, d:d+w] = src[:, s:s+w] for (s, d, w) in spans (<= 4), one launch."""
    arr = (_lib.ColSpan * len(spans))()
    for i, (sc, dc, w) in enumerate(spans):
        arr[i].src_col, arr[i].dst_col, arr[i].width = sc, dc, w
    dp, ldd = _mat(dst, "dst")
    sp, lds = _mat(src, "src")
    _lib.check(_lib.fn("dsc_gather_columns_f32")(dp, ldd, sp, lds, dst.shape[0], arr, len(spans), stream_ptr()), "dsc_gather_columns_f32")
    return dst


def activation(x, act, out=None):
    _dev(x)
    if not x.is_contiguous():
        raise RuntimeError("activation needs a contiguous tensor")
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_lib.fn("dsc_activation_f32")(x.data_ptr(), out.data_ptr(), x.numel(), act, stream_ptr()),
               "dsc_activation_f32")
    return out


# ---------------------------------------------------------------------------------- diffusion steps

def _c(t, name):
    _dev(t, name)
    if not t.is_contiguous():
        raise RuntimeError("diffuscene_amd: %s must be contiguous" % name)
    return t


def _table_rows(*tables):
    """Rows of the schedule tables a kernel indexes with the device timestep (it clamps t into them): all must agree."""
    rows = {int(tb.numel()) for tb in tables if tb is not None}
    if len(rows) != 1:
        raise RuntimeError("diffuscene_amd: schedule tables of different lengths %s" % sorted(rows))
    for tb in tables:
        if tb is not None:
            _c(tb, "schedule table")
    return rows.pop()


def q_sample(x0, noise, t, sqrt_ac, sqrt_1mac, want_v=False):
    _c(x0, "x0"); _c(noise, "noise"); _dev(t, "t", torch.int64)
    xt = torch.empty_like(x0)
    v = torch.empty_like(x0) if want_v else None
    b = x0.shape[0]
    _lib.check(_lib.fn("dsc_q_sample_f32")(x0.data_ptr(), noise.data_ptr(), t.data_ptr(), sqrt_ac.data_ptr(),
                                           sqrt_1mac.data_ptr(), xt.data_ptr(), v.data_ptr() if want_v else None,
                                           b, x0.numel() // b, _table_rows(sqrt_ac, sqrt_1mac), stream_ptr()), "dsc_q_sample_f32")
    return (xt, v) if want_v else xt


def p_sample(x_t, model_out, noise, t, ca, cb, coef1, coef2, sigma, mean_type, clip, out=None, x0_out=None):
    _c(x_t, "x_t"); _c(model_out, "model_out"); _c(noise, "noise"); _dev(t, "t", torch.int64)
    if out is None:
        out = torch.empty_like(x_t)
    b = x_t.shape[0]
    _lib.check(_lib.fn("dsc_p_sample_f32")(x_t.data_ptr(), model_out.data_ptr(), noise.data_ptr(), t.data_ptr(),
                                           ca.data_ptr() if ca is not None else None,
                                           cb.data_ptr() if cb is not None else None,
                                           coef1.data_ptr(), coef2.data_ptr(), sigma.data_ptr(), out.data_ptr(),
                                           x0_out.data_ptr() if x0_out is not None else None,
                                           mean_type, 1 if clip else 0, b, x_t.numel() // b,
                                           _table_rows(ca, cb, coef1, coef2, sigma), stream_ptr()),
               "dsc_p_sample_f32")
    return out


def add_scalar_i64(t, delta):
    _dev(t, "t", torch.int64)
    _lib.check(_lib.fn("dsc_add_scalar_i64")(t.data_ptr(), t.numel(), delta, stream_ptr()), "dsc_add_scalar_i64")
    return t


def postfilter_compact(samples, empty_col, per_scene=False, keep_empty=False):
    """(B,N,C) generated scenes -> (packed (B,N,C) with the kept slots first, counts (B,) int32); see the C header."""
    _c(samples, "samples")
    b, n, c = samples.shape
    packed = torch.empty_like(samples)
    counts = torch.empty((b,), device=samples.device, dtype=torch.int32)
    _lib.check(_lib.fn("dsc_postfilter_compact_f32")(samples.data_ptr(), b, n, c, int(empty_col), 1 if per_scene else 0,
                                                     1 if keep_empty else 0, packed.data_ptr(), counts.data_ptr(),
                                                     stream_ptr()), "dsc_postfilter_compact_f32")
    return packed, counts


def complete_overwrite(x, partial, noise, t, sqrt_ac, sqrt_1mac):
    _c(x, "x"); _c(partial, "partial"); _c(noise, "noise"); _dev(t, "t", torch.int64)
    b, n, c = x.shape
    p = partial.shape[1]
    _lib.check(_lib.fn("dsc_complete_overwrite_f32")(x.data_ptr(), partial.data_ptr(), noise.data_ptr(), t.data_ptr(),
                                                     sqrt_ac.data_ptr(), sqrt_1mac.data_ptr(), b, n, p, c,
                                                     _table_rows(sqrt_ac, sqrt_1mac), stream_ptr()), "dsc_complete_overwrite_f32")
    return x


# ---------------------------------------------------------------------------------- training (backward) kernels

_scratch = {}


def scratch(device, floats):
    """Scratch for split reductions, one buffer per (device, stream): kernels run in order on a stream, so it is reused."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    t = _scratch.get(key)
    if t is None or t.numel() < floats:
        t = torch.empty(max(int(floats), 1 << 20), device=device, dtype=torch.float32)
        _scratch[key] = t
    return t


def transpose(w, out=None, ldo=None):
    wp, ldi = _mat(w, "w")
    rows, cols = w.shape
    if out is None:
        out = torch.empty((cols, rows), device=w.device, dtype=torch.float32)
    op, ld = _mat(out, "out")
    _lib.check(_lib.fn("dsc_transpose_f32")(wp, ldi, op, ld, rows, cols, stream_ptr()), "dsc_transpose_f32")
    return out


def transpose_many(mats):
    """[w_i (rows_i, cols_i)] -> [w_i^T], one launch per DSC_WS_MAX matrices."""
    outs = [torch.empty((m.shape[1], m.shape[0]), device=m.device, dtype=torch.float32) for m in mats]
    for i in range(0, len(mats), _lib.WS_MAX):
        ms, os_ = mats[i:i + _lib.WS_MAX], outs[i:i + _lib.WS_MAX]
        arr = (_lib.WsItem * len(ms))()
        for j, (m, o) in enumerate(zip(ms, os_)):
            if not (_dev(m).is_contiguous() and m.dim() == 2):
                raise RuntimeError("transpose_many needs contiguous 2-D matrices")
            arr[j].w, arr[j].out = m.data_ptr(), o.data_ptr()
            arr[j].rows, arr[j].cols = m.shape
        _lib.check(_lib.fn("dsc_transpose_batched_f32")(arr, len(ms), stream_ptr()), "dsc_transpose_batched_f32")
    return outs


def gemm_tn(a, dy, a2=None, kvalid=None, out=None, want_bias=False):
    """out[n][k] = sum_m dy[m][n] * [a|a2][m][k]  (+ column sums of dy when want_bias) -> out or (out, dbias)"""
    ap, lda = _mat(a, "a")
    dp, ldd = _mat(dy, "dy")
    k1 = a.shape[1]
    a2p, lda2, k2 = (None, 0, 0)
    if a2 is not None:
        a2p, lda2 = _mat(a2, "a2")
        k2 = a2.shape[1]
    K = k1 + k2
    kv = K if kvalid is None else kvalid
    m, n = dy.shape
    if out is None:
        out = torch.empty((n, kv), device=a.device, dtype=torch.float32)
    op, ldo = _mat(out, "out")
    wsf = _lib.fn("dsc_gemm_tn_workspace_floats")(m, n, kv)
    ws = scratch(a.device, wsf) if wsf else None
    db = torch.empty((n,), device=a.device, dtype=torch.float32) if want_bias else None
    _lib.check(_lib.fn("dsc_gemm_tn_f32")(ap, lda, k1, a2p, lda2, k2, dp, ldd, op, ldo,
                                          db.data_ptr() if want_bias else None, m, n, kv,
                                          ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
                                          stream_ptr()), "dsc_gemm_tn_f32")
    return (out, db) if want_bias else out


def colsum(x, out=None):
    xp, ldx = _mat(x, "x")
    m, n = x.shape
    if out is None:
        out = torch.empty((n,), device=x.device, dtype=torch.float32)
    ws = scratch(x.device, 64 * n)
    _lib.check(_lib.fn("dsc_colsum_f32")(xp, ldx, m, n, out.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr()),
               "dsc_colsum_f32")
    return out


def gn_silu_bwd(z, dy, gamma, beta, ss, ss_mode, scenes, n_tok, eps=1e-5):
    """-> dz [M,512], (dgamma, dbeta, dbias) [512] each, dss (per-scene [B,1024] | per-token [M,1024] | None)"""
    zp, ldz = _mat(z, "z")
    dp, ldy = _mat(dy, "dy")
    M, Cc = z.shape
    dz = torch.empty((M, Cc), device=z.device, dtype=torch.float32)
    part = torch.empty((scenes, 3 * Cc), device=z.device, dtype=torch.float32)     # [dgamma | dbeta | dbias] per scene
    dss = None
    sp, ld_ss, ld_dss = None, 0, 0
    if ss is not None and ss_mode != SS_NONE:
        sp, ld_ss = _mat(ss, "ss")
        dss = torch.empty((scenes if ss_mode == SS_PER_SCENE else M, 2 * Cc), device=z.device, dtype=torch.float32)
        ld_dss = 2 * Cc
    _lib.check(_lib.fn("dsc_gn_silu_bwd_f32")(zp, ldz, dp, ldy, _dev(gamma).data_ptr(), _dev(beta).data_ptr(), sp, ld_ss,
                                              ss_mode if sp else SS_NONE, dz.data_ptr(), Cc, part.data_ptr(),
                                              part.data_ptr() + 4 * Cc, part.data_ptr() + 8 * Cc, 3 * Cc,
                                              dss.data_ptr() if dss is not None else None, ld_dss, scenes, n_tok, Cc,
                                              eps, stream_ptr()), "dsc_gn_silu_bwd_f32")
    red = colsum(part)                  # one column sum over the scenes for the three per-channel gradients
    return dz, red[:Cc], red[Cc:2 * Cc], red[2 * Cc:], dss


def weight_standardize_bwd(weights, dw_stds, eps=1e-5):
    outs = [torch.empty_like(as2d(w)) for w in weights]
    for i in range(0, len(weights), _lib.WS_MAX):
        ws, gs, os_ = weights[i:i + _lib.WS_MAX], dw_stds[i:i + _lib.WS_MAX], outs[i:i + _lib.WS_MAX]
        arr = (_lib.WsBwdItem * len(ws))()
        for j, (w, g, o) in enumerate(zip(ws, gs, os_)):
            w2, g2 = as2d(_dev(w)), _dev(g)
            if not (w2.is_contiguous() and g2.is_contiguous()):
                raise RuntimeError("weight_standardize_bwd needs contiguous matrices")
            arr[j].w, arr[j].dw_std, arr[j].dw = w2.data_ptr(), g2.data_ptr(), o.data_ptr()
            arr[j].rows, arr[j].cols = w2.shape
        _lib.check(_lib.fn("dsc_weight_standardize_bwd_f32")(arr, len(ws), eps, stream_ptr()),
                   "dsc_weight_standardize_bwd_f32")
    return outs


def layernorm_bwd(x, g, dy, eps=1e-5):
    xp, ldx = _mat(x, "x")
    dp, ldy = _mat(dy, "dy")
    M, Dd = x.shape
    dx = torch.empty((M, Dd), device=x.device, dtype=torch.float32)
    nblk = min((M + 3) // 4, 512)
    part = torch.empty((nblk, Dd), device=x.device, dtype=torch.float32)
    _lib.check(_lib.fn("dsc_layernorm_bwd_f32")(xp, ldx, _dev(g).data_ptr(), dp, ldy, dx.data_ptr(), Dd, None, 0, part.data_ptr(),
                                                nblk, M, Dd, eps, stream_ptr()), "dsc_layernorm_bwd_f32")
    return dx, colsum(part)


def linear_attention_bwd(q, k, v, dout, dq, dk, dv, scenes, nq, nk, scale):
    args = []
    for t, nme in ((q, "q"), (k, "k"), (v, "v"), (dout, "dout"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        args += list(_mat(t, nme))
    _lib.check(_lib.fn("dsc_linear_attention_bwd_f32")(*args, scenes, nq, nk, scale, stream_ptr()),
               "dsc_linear_attention_bwd_f32")


def attention_bwd(q, k, v, dout, dq, dk, dv, scenes, n, scale):
    args = []
    for t, nme in ((q, "q"), (k, "k"), (v, "v"), (dout, "dout"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        args += list(_mat(t, nme))
    _lib.check(_lib.fn("dsc_attention_bwd_f32")(*args, scenes, n, scale, stream_ptr()), "dsc_attention_bwd_f32")


def activation_bwd(x, dy, act):
    _c(x, "x"); _c(dy, "dy")
    dx = torch.empty_like(x)
    _lib.check(_lib.fn("dsc_activation_bwd_f32")(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), act,
                                                 stream_ptr()), "dsc_activation_bwd_f32")
    return dx


def ddpm_loss(target, out, x_t, t, loss_weight, ca, cb, alphas_cumprod, bounds, dims, separate, iou, mean_type,
              grad_scale=1.0):
    """-> losses_weight [B], parts [B, 9], dout [B, N, C] (grad_scale * d losses_weight[b] / d out[b])"""
    _c(target, "target"); _c(out, "out"); _c(x_t, "x_t"); _dev(t, "t", torch.int64)
    B, N, Cc = out.shape
    losses = torch.empty((B,), device=out.device, dtype=torch.float32)
    parts = torch.empty((B, 9), device=out.device, dtype=torch.float32)
    dout = torch.empty_like(out)
    barr = (C.c_float * 12)(*[float(v) for v in bounds]) if bounds is not None else None
    _lib.check(_lib.fn("dsc_ddpm_loss_f32")(
        target.data_ptr(), out.data_ptr(), x_t.data_ptr(), t.data_ptr(), loss_weight.data_ptr(),
        ca.data_ptr() if ca is not None else None, cb.data_ptr() if cb is not None else None,
        alphas_cumprod.data_ptr() if alphas_cumprod is not None else None, barr,
        losses.data_ptr(), parts.data_ptr(), dout.data_ptr(), B, N, Cc, dims["translation_dim"], dims["size_dim"],
        dims["bbox_dim"], dims["class_dim"], dims["objectness_dim"], dims["objfeat_dim"], 1 if separate else 0,
        1 if iou else 0, mean_type, float(grad_scale), _table_rows(loss_weight, ca, cb, alphas_cumprod), stream_ptr()),
        "dsc_ddpm_loss_f32")
    return losses, parts, dout
