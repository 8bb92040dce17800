"""torch.autograd.Function wrappers: forward AND backward of every op are hand-written HIP kernels
(csrc/gemm_mfma.hip, blocks.hip, train.hip); autograd only orders the calls and accumulates parameter
gradients.  ``unet1d_train_forward`` is the differentiable twin of ``engine.Plan`` (reference
Unet1D.forward, denoise_net.py:507-593).
"""
import torch
from torch.autograd import Function

from . import ops
from ._lib import ACT_GELU, ACT_SILU, SS_NONE, SS_PER_SCENE, SS_PER_SLOT, SS_PER_TOKEN
from .ops import as2d

D = 512
HID = 128


def _dense(t):
    """Gradient tensors arrive as arbitrary views; kernels want contiguous rows with 16-byte aligned starts."""
    if t.stride(-1) != 1 or (t.dim() == 2 and t.shape[0] > 1 and t.stride(0) % 4) or t.data_ptr() % 16:
        return t.contiguous()
    return t


def _pad_cols(t, mult=32):
    """[M, n] -> zero-padded contiguous [M, ceil(n, mult)] (reduction dims of the MFMA kernels are multiples of 32)."""
    n = t.shape[1]
    npad = (n + mult - 1) // mult * mult
    if npad == n:
        return t
    out = torch.zeros((t.shape[0], npad), device=t.device, dtype=t.dtype)
    out[:, :n].copy_(t)
    return out


class LinearFn(Function):
    """y = [a | a2] @ w.T + bias (+ residual)   -- nn.Conv1d(k=1) / nn.Linear"""

    @staticmethod
    def forward(ctx, a, w, bias, a2, residual):
        y = ops.gemm(a, w, bias, a2=a2, residual=residual)
        ctx.save_for_backward(a, w, a2)
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        a, w, a2 = ctx.saved_tensors
        w2 = as2d(w)
        n, K = w2.shape
        k1 = a.shape[1]
        dyp = _pad_cols(_dense(dy))                 # [M, n_pad]
        npad = dyp.shape[1]
        da = da2 = dw = db = None
        if ctx.needs_input_grad[0] or (a2 is not None and ctx.needs_input_grad[3]):
            wt = torch.zeros((K, npad), device=w.device) if npad != n else torch.empty((K, n), device=w.device)
            ops.transpose(w2, out=wt[:, :n] if npad != n else wt)
            if npad > 4096:
                # long reductions (the packed 19x1024 time-MLP outputs): accumulate in chunks of 2048 so the fp32
                # error stays that of a blocked sum instead of one 19456-term sequential chain
                d_in = None
                for c0 in range(0, npad, 2048):
                    c1 = min(c0 + 2048, npad)
                    d_in = ops.gemm(dyp[:, c0:c1], wt[:, c0:c1], residual=d_in)
            else:
                d_in = ops.gemm(dyp, wt)            # [M, K] = dy @ w
            da = d_in[:, :k1] if a2 is not None else d_in
            da2 = d_in[:, k1:] if a2 is not None else None
        if ctx.needs_input_grad[1]:
            if ctx.has_bias and ctx.needs_input_grad[2]:
                dw, db = ops.gemm_tn(a, dyp, a2=a2, want_bias=True)      # bias gradient rides on the same pass over dy
                db = db[:n]
            else:
                dw = ops.gemm_tn(a, dyp, a2=a2)     # [n_pad, K]
            dw = dw[:n].reshape(w.shape)
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.colsum(dyp)[:n]
        return da, dw, db, da2, (dy if ctx.has_res else None)


class SmallKLinearFn(Function):
    """First encoder layer on an un-aligned column slice of the (B,N,C) data tensor (no input gradient)."""

    @staticmethod
    def forward(ctx, x_slice, w, bias):
        y = ops.linear_smallk(x_slice, w, bias)
        xpad = torch.zeros((x_slice.shape[0], 32 if x_slice.shape[1] <= 32 else 64), device=x_slice.device)
        xpad[:, :x_slice.shape[1]].copy_(x_slice)
        ctx.save_for_backward(xpad)
        ctx.k, ctx.wshape = x_slice.shape[1], w.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        xpad, = ctx.saved_tensors
        dyd = _dense(dy)
        dw, db = ops.gemm_tn(xpad, dyd, kvalid=ctx.k, want_bias=True)
        return None, dw.reshape(ctx.wshape), db


class ActFn(Function):
    @staticmethod
    def forward(ctx, x, act):
        ctx.save_for_backward(x)
        ctx.act = act
        return ops.activation(x, act)

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        return ops.activation_bwd(x, dy.contiguous(), ctx.act), None


class WeightStandardizeAllFn(Function):
    """All weight-standardised conv weights of the network in one batched launch (forward and backward)."""

    @staticmethod
    def forward(ctx, *weights):
        outs = ops.weight_standardize(list(weights))
        ctx.save_for_backward(*weights)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        weights = ctx.saved_tensors
        gs = [g.contiguous() if g is not None else torch.zeros_like(as2d(w)) for g, w in zip(grads, weights)]
        dws = ops.weight_standardize_bwd(list(weights), gs)
        return tuple(dw.reshape(w.shape) for dw, w in zip(dws, weights))


class ConvGnSiluFn(Function):
    """Block.forward: y = SiLU(GroupNorm8([a|a2] @ w_std.T + bias) * (scale+1) + shift) (+ residual)"""

    @staticmethod
    def forward(ctx, a, w_std, bias, gamma, beta, a2, ss, residual, n_tok, ss_mode, w_std_t=None):
        ctx.w_std_t = w_std_t          # optional pre-transposed weight (K, 512): all layers transposed in one launch
        z = torch.empty((a.shape[0], w_std.shape[0]), device=a.device, dtype=torch.float32)
        y = ops.gemm_gn_silu(a, w_std, bias, gamma, beta, n_tok, a2=a2, scale_shift=ss,
                             ss_mode=ss_mode if ss is not None else SS_NONE, residual=residual, preact=z)
        ctx.save_for_backward(a, w_std, gamma, beta, a2, ss, z)
        ctx.n_tok, ctx.ss_mode, ctx.has_res = n_tok, (ss_mode if ss is not None else SS_NONE), residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        a, w_std, gamma, beta, a2, ss, z = ctx.saved_tensors
        dyd = _dense(dy)
        M = z.shape[0]
        scenes = M // ctx.n_tok
        dz, dgamma, dbeta, dbias, dss = ops.gn_silu_bwd(z, dyd, gamma, beta, ss, ctx.ss_mode, scenes, ctx.n_tok)
        if dss is not None and ctx.ss_mode == SS_PER_SLOT:
            dss = dss.view(scenes, ctx.n_tok, -1).sum(0)
        k1 = a.shape[1]
        da = da2 = None
        if ctx.needs_input_grad[0] or (a2 is not None and ctx.needs_input_grad[5]):
            d_in = ops.gemm(dz, ctx.w_std_t if ctx.w_std_t is not None else ops.transpose(w_std))
            da = d_in[:, :k1] if a2 is not None else d_in
            da2 = d_in[:, k1:] if a2 is not None else None
        dw = ops.gemm_tn(a, dz, a2=a2)
        return da, dw, dbias, dgamma, dbeta, da2, dss, (dy if ctx.has_res else None), None, None, None


class LayerNormFn(Function):
    """channel LayerNorm with gain (+ residual)"""

    @staticmethod
    def forward(ctx, x, g, residual):
        y = ops.layernorm(x, g, residual=residual)
        ctx.save_for_backward(x, g)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        dx, dg = ops.layernorm_bwd(x, g, _dense(dy))
        return dx, dg.reshape(g.shape), (dy if ctx.has_res else None)


class LinearAttentionSelfFn(Function):
    @staticmethod
    def forward(ctx, qkv, scenes, n, scale):
        out = ops.linear_attention(qkv[:, :HID], qkv[:, HID:2 * HID], qkv[:, 2 * HID:], scenes, n, n, scale)
        ctx.save_for_backward(qkv)
        ctx.cfg = (scenes, n, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, = ctx.saved_tensors
        scenes, n, scale = ctx.cfg
        d = torch.empty_like(qkv)
        ops.linear_attention_bwd(qkv[:, :HID], qkv[:, HID:2 * HID], qkv[:, 2 * HID:], _dense(dout),
                                 d[:, :HID], d[:, HID:2 * HID], d[:, 2 * HID:], scenes, n, n, scale)
        return d, None, None, None


class LinearAttentionCrossFn(Function):
    @staticmethod
    def forward(ctx, q, kv, scenes, nq, nk, scale):
        out = ops.linear_attention(q, kv[:, :HID], kv[:, HID:], scenes, nq, nk, scale)
        ctx.save_for_backward(q, kv)
        ctx.cfg = (scenes, nq, nk, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv = ctx.saved_tensors
        scenes, nq, nk, scale = ctx.cfg
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        ops.linear_attention_bwd(q, kv[:, :HID], kv[:, HID:], _dense(dout), dq, dkv[:, :HID], dkv[:, HID:],
                                 scenes, nq, nk, scale)
        return dq, dkv, None, None, None, None


class AttentionFn(Function):
    @staticmethod
    def forward(ctx, qkv, scenes, n, scale):
        out = ops.attention(qkv[:, :HID], qkv[:, HID:2 * HID], qkv[:, 2 * HID:], scenes, n, scale)
        ctx.save_for_backward(qkv)
        ctx.cfg = (scenes, n, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, = ctx.saved_tensors
        scenes, n, scale = ctx.cfg
        d = torch.empty_like(qkv)
        ops.attention_bwd(qkv[:, :HID], qkv[:, HID:2 * HID], qkv[:, 2 * HID:], _dense(dout),
                          d[:, :HID], d[:, HID:2 * HID], d[:, 2 * HID:], scenes, n, scale)
        return d, None, None, None


# --------------------------------------------------------------------------------------------------------------
def linear_any(x, weight, bias=None):
    """nn.Linear semantics for any leading shape and any in_features on the HIP GEMM (forward and backward): the reduction
    dimension is zero-padded to a multiple of 32 when needed (differentiable copies; these are the small conditioning layers)."""
    lead, k = x.shape[:-1], x.shape[-1]
    x2 = x.reshape(-1, k)
    w2 = as2d(weight)
    if k % 32:
        kp = (k + 31) // 32 * 32
        xp = torch.zeros((x2.shape[0], kp), device=x.device, dtype=x.dtype)
        xp[:, :k] = x2
        wp = torch.zeros((w2.shape[0], kp), device=x.device, dtype=x.dtype)
        wp[:, :k] = w2
        x2, w2 = xp, wp
    elif not x2.is_contiguous() or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    y = LinearFn.apply(x2, w2, bias, None, None)
    return y.reshape(*lead, y.shape[-1])


def linear(a, mod, a2=None, residual=None):
    return LinearFn.apply(a, mod.weight, mod.bias, a2, residual)


def unet1d_train_forward(net, x, t, context, context_cross):
    """Differentiable Unet1D forward on HIP kernels.  x (B,N,C), t (B,) int64 -> (B,N,C)."""
    B, N, C = x.shape
    M = B * N
    dev = x.device
    if N > 160:
        raise RuntimeError("diffuscene_amd: at most 160 objects per scene")
    eng = net.engine(dev)                    # only for the time table / block ordering
    xf = x.reshape(M, C)

    # weight standardisation of all 56 convs: one launch forward, one backward
    ws_mods = eng.ws_mods
    ws_list = WeightStandardizeAllFn.apply(*[m.weight for m in ws_mods])
    ws = {id(m): w for m, w in zip(ws_mods, ws_list)}
    # W^T of every standardised weight for the dA GEMMs of the backward: one launch instead of one per layer
    ws_t = {id(m): t for m, t in zip(ws_mods, ops.transpose_many([as2d(w.detach()) for w in ws_list]))} \
        if torch.is_grad_enabled() else {}

    # conditioning
    temb = ops.time_embedding(t, D, eng.time_table, eng.time_freq)
    t1 = ActFn.apply(linear(temb, net.time_mlp[1]), ACT_GELU)
    t2 = ActFn.apply(linear(t1, net.time_mlp[3]), ACT_SILU)
    wt = torch.cat([rb.mlp[1].weight for rb in eng.t_blocks], 0)
    bt = torch.cat([rb.mlp[1].bias for rb in eng.t_blocks], 0)
    ss_t = LinearFn.apply(t2, wt, bt, None, None)                       # (B, 19*1024)
    ss_c, ctx_mode = None, SS_NONE
    if context is not None and eng.c_pack_w is not None:
        shared = context.stride(0) == 0 or B == 1
        ctx_mode = SS_PER_SLOT if shared else SS_PER_TOKEN
        crow = context[0] if shared else context.reshape(M, context.shape[-1])
        cact = ActFn.apply(crow.contiguous(), ACT_SILU)
        wc = torch.cat([rb.mlp[1].weight for rb in eng.c_blocks], 0)
        bc = torch.cat([rb.mlp[1].bias for rb in eng.c_blocks], 0)
        ss_c = LinearFn.apply(cact, wc, bc, None, None)

    def t_ss(rb):
        i = eng.t_index[id(rb)]
        return ss_t[:, i * 2 * D:(i + 1) * 2 * D], SS_PER_SCENE

    def c_ss(rb):
        if ss_c is None:
            return None, SS_NONE
        i = eng.c_index[id(rb)]
        return ss_c[:, i * 2 * D:(i + 1) * 2 * D], ctx_mode

    def resblock(rb, a, a2, ss_pair):
        ss, mode = ss_pair
        if ss is not None:
            ss = ss.contiguous() if ss.data_ptr() % 16 or ss.stride(0) % 4 else ss
        h = ConvGnSiluFn.apply(a, ws[id(rb.block1.proj)], rb.block1.proj.bias, rb.block1.norm.weight,
                               rb.block1.norm.bias, a2, ss, None, N, mode, ws_t.get(id(rb.block1.proj)))
        r = LinearFn.apply(a, rb.res_conv.weight, rb.res_conv.bias, a2, None) if rb.has_res_conv else a
        return ConvGnSiluFn.apply(h, ws[id(rb.block2.proj)], rb.block2.proj.bias, rb.block2.norm.weight,
                                  rb.block2.norm.bias, None, None, r, N, SS_NONE, ws_t.get(id(rb.block2.proj)))

    def linattn(blk, a):
        att = blk.fn.fn
        y = LayerNormFn.apply(a, blk.fn.norm.g.view(-1), None)
        qkv = LinearFn.apply(y, att.to_qkv.weight, None, None, None)
        o = LinearAttentionSelfFn.apply(qkv, B, N, float(att.scale))
        o = linear(o, att.to_out[0])
        return LayerNormFn.apply(o, att.to_out[1].g.view(-1), a)

    def crossattn(blk, a):
        att = blk.fn.fn
        L = context_cross.shape[1]
        y = LayerNormFn.apply(a, blk.fn.norm.g.view(-1), None)
        q = LinearFn.apply(y, att.to_q.weight, None, None, None)
        kv = LinearFn.apply(context_cross.reshape(B * L, -1), att.to_kv.weight, None, None, None)
        o = LinearAttentionCrossFn.apply(q, kv, B, N, L, float(att.scale))
        o = linear(o, att.to_out[0])
        return LayerNormFn.apply(o, att.to_out[1].g.view(-1), a)

    def fullattn(blk, a):
        att = blk.fn.fn
        y = LayerNormFn.apply(a, blk.fn.norm.g.view(-1), None)
        qkv = LinearFn.apply(y, att.to_qkv.weight, None, None, None)
        o = AttentionFn.apply(qkv, B, N, float(att.scale))
        return LinearFn.apply(o, att.to_out.weight, att.to_out.bias, None, a)

    def enc(seq, c0, k, acc):
        h = ActFn.apply(SmallKLinearFn.apply(xf[:, c0:c0 + k], seq[0].weight, seq[0].bias), ACT_GELU)
        h = ActFn.apply(linear(h, seq[2]), ACT_GELU)
        return linear(h, seq[4], residual=acc)

    text = net.text_condition and context_cross is not None
    if net.seperate_all:
        bb, nc, no, nf = net.bbox_dim, net.class_dim, net.objectness_dim, net.objfeat_dim
        e = enc(net.class_embedf, bb, nc, None)
        e = enc(net.bbox_embedf, 0, bb, e)
        if no > 0:
            e = enc(net.objectness_embedf, bb + nc, no, e)
        if nf > 0:
            e = enc(net.objfeat_embedf, bb + nc + no, nf, e)
        h = linear(e, net.init_conv)
    else:
        h = SmallKLinearFn.apply(xf, net.init_conv.weight, net.init_conv.bias)
    r = h
    skips = []
    for lvl in net.downs:
        b0, b1, ac, b2, la, down = lvl
        h = resblock(b0, h, None, c_ss(b0))
        h = resblock(b1, h, None, t_ss(b1))
        skips.append(h)
        if text:
            h = crossattn(ac, h)
        h = resblock(b2, h, None, t_ss(b2))
        h = linattn(la, h)
        skips.append(h)
        if isinstance(down, torch.nn.Conv1d):
            h = linear(h, down)
    h = resblock(net.mid_block0, h, None, c_ss(net.mid_block0))
    h = resblock(net.mid_block1, h, None, t_ss(net.mid_block1))
    if text:
        h = crossattn(net.mid_attn_cross, h)
    h = fullattn(net.mid_attn, h)
    h = resblock(net.mid_block2, h, None, t_ss(net.mid_block2))
    for lvl in net.ups:
        b0, b1, ac, b2, la, up = lvl
        h = resblock(b0, h, None, c_ss(b0))
        h = resblock(b1, h, skips.pop(), t_ss(b1))
        if text:
            h = crossattn(ac, h)
        h = resblock(b2, h, skips.pop(), t_ss(b2))
        h = linattn(la, h)
        if isinstance(up, torch.nn.Conv1d):
            h = linear(h, up)
    h = resblock(net.final_res_block, h, r, t_ss(net.final_res_block))
    if net.seperate_all:
        heads = [net.bbox_hidden2output, net.class_hidden2output]
        if net.objectness_dim > 0:
            heads.append(net.objectness_hidden2output)
        if net.objfeat_dim > 0:
            heads.append(net.objfeat_hidden2output)
        outs = []
        for seq in heads:
            d1 = ActFn.apply(linear(h, seq[0]), ACT_GELU)
            d2 = ActFn.apply(linear(d1, seq[2]), ACT_GELU)
            outs.append(linear(d2, seq[4]))
        out = torch.cat(outs, dim=1)
    else:
        out = linear(h, net.final_conv)
    return out.view(B, N, -1)
