"""Launch plans for the Unet1D denoiser on MI355X.

A plan is the whole forward pass (reference: denoise_net.py:507-593) flattened ONCE into a list of C-ABI
calls with pre-built argument structs and statically allocated activation buffers; running it is a loop of
~140 ctypes calls on the current stream with no allocation, no Python tensor ops and fixed pointers --
exactly what a hipGraph capture needs.  Activations are token-major [B*N, channels]; torch.cat of skip
connections never happens (second K segment of the GEMM), the conv layout (B, C, N) is never materialised.

Derived weights (refreshed when a parameter changes): weight-standardised copies of the 56 WS-convs
(constant during sampling -> standardised once), and the 19 time-MLP / 9 context-MLP
Linear layers packed into one [19*1024, 2048] / [9*1024, ctx] matrix so each is ONE GEMM per forward.
"""
import ctypes as C

import torch

from . import _lib, ops
from ._lib import ACT_GELU, ACT_NONE, ACT_SILU, SS_BY_INDEX, SS_NONE, SS_PER_SCENE, SS_PER_SLOT, SS_PER_TOKEN

D = 512
HID = 128
DEC_PAD = 32        # rows per head of the stacked, zero-padded decoder output projections (bbox 8, class <= 32, objfeat 32); a wider head
                    # (objfeat_dim = 64) raises the engine's dec_pad to the next multiple


def _own(t):
    """What a plan keeps for a raw pointer it baked into a launch: tensors as they are, an nn.Parameter as a detached ALIAS of its
    current storage.  `p.data = ...` (flat.FlatStorage re-homes every parameter on the first training step; module.to()) swaps a
    Parameter's storage in place and frees the old one; the alias keeps the storage the pointer refers to alive for as long as the
    plan (and any hipGraph captured from it) lives, so a plan that outlives such a move reads stale weights -- and says so, see
    Plan.run -- instead of freed memory."""
    if isinstance(t, torch.nn.Parameter):
        return t.detach()
    if isinstance(t, (tuple, list)):
        return tuple(_own(u) for u in t)
    return t


class StalePlanError(RuntimeError):
    pass


class _Pool:
    def __init__(self, device):
        self.device = device
        self.free_list = {}
        self.all = []
        self.bytes = 0

    def get(self, rows, cols):
        key = (rows, cols)
        lst = self.free_list.get(key)
        if lst:
            return lst.pop()
        t = torch.empty((rows, cols), device=self.device, dtype=torch.float32)
        self.all.append(t)
        self.bytes += t.numel() * 4
        return t

    def put(self, t):
        self.free_list.setdefault((t.shape[0], t.shape[1]), []).append(t)


class Plan:
    """Static launch list for one (B, N, conditioning) signature."""

    def __init__(self, eng, B, N, ctx_mode, ctx_dim, L, text_dim, time_table=False):
        self.eng, self.B, self.N, self.M = eng, B, N, B * N
        self.generation = eng.generation          # the parameter storages this plan's pointers refer to (DenoiserEngine.refresh)
        dev = eng.device
        self.pool = _Pool(dev)
        self.keep = []          # tensors / structs referenced by raw pointer
        self.steps = []         # (cfunc, args_tuple)
        # launches whose inputs do not change during a reverse loop (sampling plans only): the context MLPs of the instance
        # embedding and the K/V projections of the text tokens (LinearAttentionCross: denoise_net.py:283-286) depend on the
        # conditioning alone -- they run once per loop in DenoiserEngine.prepare(), not 1000 times inside the captured step
        self.pre_steps = []
        net = eng.net
        C_in = net.channels
        self.x_in = torch.empty((self.M, C_in), device=dev, dtype=torch.float32)
        self.t_in = torch.empty((B,), device=dev, dtype=torch.int64)
        ctx_rows = {SS_NONE: 0, SS_PER_SLOT: N, SS_PER_TOKEN: self.M}[ctx_mode]
        self.ctx_in = torch.empty((ctx_rows, ctx_dim), device=dev, dtype=torch.float32) if ctx_rows else None
        self.cross_in = torch.empty((B * L, text_dim), device=dev, dtype=torch.float32) if L else None
        self.ctx_mode = ctx_mode
        self.L = L
        self.time_table = time_table
        self.out = torch.empty((self.M, net.out_dim), device=dev, dtype=torch.float32)
        self._build()
        self.tiled_steps = self.steps            # one launch per layer (bench.py times the dominant kernel from these)

    def gemm_args(self):
        """('gn' | 'plain', dsc_gemm_args) of every GEMM launch of the step, in launch order -- what tests and bench.py hand to
        dsc_gemm_arithmetic to learn which kernel (split-bf16 / exact-f32 MFMA) each launch runs."""
        kinds = {id(_lib.fn("dsc_gemm_gn_silu_f32")): "gn", id(_lib.fn("dsc_gemm_f32")): "plain"}
        out = []
        for f, a in self.steps:
            k = kinds.get(id(f))
            if k is not None:
                out.append((k, a[0]._obj))            # ctypes.byref(struct) keeps the struct in ._obj
        return out

    # ---- step emitters -------------------------------------------------------------------------
    def gemm(self, a, w, out, bias=None, a2=None, residual=None, act_in=ACT_NONE, act_out=ACT_NONE, row_invariant=False):
        g = ops.make_gemm_args(a, w, out, bias, a2, residual, act_in, act_out, row_invariant=row_invariant)
        lay = ops.planes_layout(g)                       # planes only for launches that will use them, in the layout their kernel reads
        pl = self.eng.planes_of(w, lay) if lay >= 0 else None
        if pl is not None:
            ops.attach_planes(g, pl, layout=lay)
        self.keep.append(_own((g, a, w, out, bias, a2, residual, pl)))
        self.steps.append((_lib.fn("dsc_gemm_f32"), (C.byref(g),)))
        return out

    def gemm_batched(self, a, w, out, bias, batch, sa, sw, sy, sbias, act_out=ACT_NONE):
        """`batch` independent products in one launch: operand / output pointers advance by (sa, sw, sy, sbias) floats per
        problem (a, w, out, bias describe problem 0: views into the wider buffers)."""
        # `w` is the first row block of the stacked weights: the split path takes the planes of the whole stack
        stacked = w._base if w._base is not None else w
        g = ops.make_gemm_args(a, w, out, bias, None, None, ACT_NONE, act_out)
        g.batch, g.sa1, g.sw, g.sy, g.sbias = batch, sa, sw, sy, sbias
        # the kernel indexes the planes as [(z n + col) K] from the START of the stack: `w` must be its first row block, the stack a
        # dense [batch n][K] matrix and the problem stride one block; anything else multiplies on the f32 kernel
        whole = (stacked.dim() == 2 and stacked.is_contiguous() and w.data_ptr() == stacked.data_ptr()
                 and stacked.shape[0] == batch * w.shape[0] and stacked.shape[1] == w.shape[1] and sw == w.shape[0] * w.shape[1])
        lay = ops.planes_layout(g) if whole else -1
        pl = self.eng.planes_of(stacked, lay) if lay >= 0 else None
        if pl is not None:
            ops.attach_planes(g, pl, layout=lay)
        self.keep.append(_own((g, a, w, out, bias, pl)))
        self.steps.append((_lib.fn("dsc_gemm_f32"), (C.byref(g),)))
        return out

    def gemm_gn(self, a, w, out, bias, gamma, beta, a2=None, ss=None, ss_mode=SS_NONE, residual=None):
        g = ops.make_gemm_args(a, w, out, bias, a2, residual, gamma=gamma, beta=beta, eps=1e-5,
                               tokens_per_scene=self.N, scale_shift=ss, ss_mode=ss_mode if ss is not None else SS_NONE,
                               ss_index=self.t_in if (ss is not None and ss_mode == SS_BY_INDEX) else None)
        lay = ops.planes_layout(g, gn=True)
        pl = self.eng.planes_of(w, lay) if lay >= 0 else None
        if pl is not None:
            ops.attach_planes(g, pl, gn=True, layout=lay)
        self.keep.append(_own((g, a, w, out, bias, a2, residual, gamma, beta, ss, pl)))
        self.steps.append((_lib.fn("dsc_gemm_gn_silu_f32"), (C.byref(g),)))
        return out

    def call(self, name, *args, keep=()):
        self.keep.append(_own(keep))
        self.steps.append((_lib.fn(name), args))

    def hoist(self, first):
        """Move the launches emitted since index ``first`` out of the per-step list (sampling plans: time_table)."""
        if self.time_table:
            self.pre_steps += self.steps[first:]
            del self.steps[first:]
            return True
        return False

    def check_current(self):
        """Raise if the network's parameters moved to other storage since this plan was built (the engine noticed in refresh()): the
        pointers baked into the launches then refer to the OLD storages -- kept alive by the plan, so nothing faults, but the values
        are no longer the model's."""
        if self.generation != self.eng.generation:
            raise StalePlanError("diffuscene_amd: this launch plan was built for parameter storages the model no longer uses (flat storage "
                                 "of the first training step, module.to(), load into new tensors): rebuild it -- DenoiserEngine.prepare() / "
                                 "graph_sample_loop() do -- instead of running or replaying it")

    def run_pre(self):
        self.check_current()
        s = ops.stream_ptr()
        for f, a in self.pre_steps:
            rc = f(*a, s)
            if rc:
                _lib.check(rc, f.__name__)

    def layernorm(self, x, g, out, residual=None):
        self.call("dsc_layernorm_f32", x.data_ptr(), x.stride(0), g.data_ptr(),
                  residual.data_ptr() if residual is not None else None,
                  residual.stride(0) if residual is not None else 0,
                  out.data_ptr(), out.stride(0), x.shape[0], x.shape[1], 1e-5, keep=(x, g, out, residual))
        return out

    def out_proj_ln(self, a, conv, ln_gain, residual):
        """to_out = Conv1d + LayerNorm of LinearAttention (:216-219) and the PreNorm block's residual.  Large batches: ONE launch
        (dsc_gemm_layernorm_f32: blocks of 96 token rows span all 512 channels, the LayerNorm runs in the GEMM epilogue); with
        fewer than 160 row blocks that kernel would leave CUs idle, so small batches keep GEMM + layernorm launches."""
        M = self.M
        out = self.pool.get(M, D)
        if (M + 95) // 96 >= 160:
            g = ops.make_gemm_args(a, conv.weight, out, conv.bias, None, residual, gamma=ln_gain.view(-1), beta=ln_gain.view(-1),
                                   eps=1e-5)
            self.keep.append(_own((g, a, conv.weight, out, conv.bias, residual, ln_gain)))
            self.steps.append((_lib.fn("dsc_gemm_layernorm_f32"), (C.byref(g),)))
            return out
        o = self.gemm(a, conv.weight, self.pool.get(M, D), conv.bias)
        self.layernorm(o, ln_gain, out, residual=residual)
        self.pool.put(o)
        return out

    # ---- network pieces ------------------------------------------------------------------------
    def resblock(self, rb, x, x2, ss, ss_mode):
        e, M = self.eng, self.M
        h = self.pool.get(M, D)
        self.gemm_gn(x, e.ws[id(rb.block1.proj)], h, rb.block1.proj.bias, rb.block1.norm.weight, rb.block1.norm.bias,
                     a2=x2, ss=ss, ss_mode=ss_mode)
        if rb.has_res_conv:
            r = self.pool.get(M, D)
            self.gemm(x, rb.res_conv.weight, r, rb.res_conv.bias, a2=x2)
        else:
            r = x
        out = self.pool.get(M, D)
        self.gemm_gn(h, e.ws[id(rb.block2.proj)], out, rb.block2.proj.bias, rb.block2.norm.weight, rb.block2.norm.bias,
                     residual=r)
        self.pool.put(h)
        if rb.has_res_conv:
            self.pool.put(r)
        return out

    def t_ss(self, rb):
        i = self.eng.t_index[id(rb)]
        return self.ss_t[:, i * 2 * D:(i + 1) * 2 * D], (SS_BY_INDEX if self.time_table else SS_PER_SCENE)

    def c_ss(self, rb):
        if self.ss_c is None:
            return None, SS_NONE
        i = self.eng.c_index[id(rb)]
        return self.ss_c[:, i * 2 * D:(i + 1) * 2 * D], self.ctx_mode

    def linattn(self, blk, x):
        """Residual(PreNorm(LinearAttention)) -> new buffer (x stays valid)."""
        M, B, N = self.M, self.B, self.N
        att = blk.fn.fn
        y = self.layernorm(x, blk.fn.norm.g, self.pool.get(M, D))
        qkv = self.gemm(y, att.to_qkv.weight, self.pool.get(M, 3 * HID))
        self.pool.put(y)
        a = self.pool.get(M, HID)
        self.call("dsc_linear_attention_f32", qkv.data_ptr(), 3 * HID, qkv.data_ptr() + 4 * HID, 3 * HID,
                  qkv.data_ptr() + 8 * HID, 3 * HID, a.data_ptr(), HID, B, N, N, float(att.scale), keep=(qkv, a))
        self.pool.put(qkv)
        out = self.out_proj_ln(a, att.to_out[0], att.to_out[1].g, x)
        self.pool.put(a)
        return out

    def crossattn(self, blk, x):
        M, B, N, L = self.M, self.B, self.N, self.L
        att = blk.fn.fn
        y = self.layernorm(x, blk.fn.norm.g, self.pool.get(M, D))
        q = self.gemm(y, att.to_q.weight, self.pool.get(M, HID))
        self.pool.put(y)
        first = len(self.steps)
        kv = self.gemm(self.cross_in, att.to_kv.weight, self.pool.get(B * L, 2 * HID))
        kv_is_invariant = self.hoist(first)             # sampling: computed once per loop, its buffer is never recycled
        a = self.pool.get(M, HID)
        self.call("dsc_linear_attention_f32", q.data_ptr(), HID, kv.data_ptr(), 2 * HID, kv.data_ptr() + 4 * HID,
                  2 * HID, a.data_ptr(), HID, B, N, L, float(att.scale), keep=(q, kv, a))
        self.pool.put(q)
        if not kv_is_invariant:
            self.pool.put(kv)
        out = self.out_proj_ln(a, att.to_out[0], att.to_out[1].g, x)
        self.pool.put(a)
        return out

    def fullattn(self, blk, x):
        M, B, N = self.M, self.B, self.N
        att = blk.fn.fn
        y = self.layernorm(x, blk.fn.norm.g, self.pool.get(M, D))
        qkv = self.gemm(y, att.to_qkv.weight, self.pool.get(M, 3 * HID))
        self.pool.put(y)
        a = self.pool.get(M, HID)
        self.call("dsc_attention_f32", qkv.data_ptr(), 3 * HID, qkv.data_ptr() + 4 * HID, 3 * HID,
                  qkv.data_ptr() + 8 * HID, 3 * HID, a.data_ptr(), HID, B, N, float(att.scale), keep=(qkv, a))
        self.pool.put(qkv)
        out = self.gemm(a, att.to_out.weight, self.pool.get(M, D), att.to_out.bias, residual=x)
        self.pool.put(a)
        return out

    def _build(self):
        e, net, M, B = self.eng, self.eng.net, self.M, self.B
        pool = self.pool
        # ---- conditioning: time MLP, then all 19 per-block Linear(2048->1024) as ONE GEMM ----------
        if self.time_table:
            # sampling: every (scale, shift) row depends on the integer timestep only -> one table row per timestep,
            # computed once per weight version by the same GEMMs (DenoiserEngine.ss_table); the per-step time MLP
            # (3 GEMMs, M = B) disappears and the epilogue gathers row t[scene]
            self.ss_t = e.ss_table()
        else:
            temb = pool.get(B, D)
            self.call("dsc_time_embedding_f32", self.t_in.data_ptr(), B, D, e.time_table.data_ptr(),
                      e.time_table.shape[0], e.time_freq.data_ptr(), temb.data_ptr(), keep=(temb,))
            # row_invariant: ss_table() below builds the same rows with m = T for the captured loops -- both must take the same kernel
            t1 = self.gemm(temb, net.time_mlp[1].weight, pool.get(B, 4 * D), net.time_mlp[1].bias, act_out=ACT_GELU, row_invariant=True)
            # every consumer applies SiLU first (ResnetBlock.mlp) -> fold it into this epilogue
            t2 = self.gemm(t1, net.time_mlp[3].weight, pool.get(B, 4 * D), net.time_mlp[3].bias, act_out=ACT_SILU, row_invariant=True)
            self.ss_t = self.gemm(t2, e.t_pack_w, pool.get(B, e.t_pack_w.shape[0]), e.t_pack_b, row_invariant=True)
        if self.ctx_in is not None:
            first = len(self.steps)
            cact = pool.get(self.ctx_in.shape[0], self.ctx_in.shape[1])       # SiLU of ResnetBlock.mlp (:181-184), once
            self.call("dsc_activation_f32", self.ctx_in.data_ptr(), cact.data_ptr(), self.ctx_in.numel(), ACT_SILU,
                      keep=(cact,))
            self.ss_c = self.gemm(cact, e.c_pack_w, pool.get(self.ctx_in.shape[0], e.c_pack_w.shape[0]), e.c_pack_b)
            self.hoist(first)                         # sampling: the context never changes inside a reverse loop
        else:
            self.ss_c = None
        # ---- input embedding ------------------------------------------------------------------------
        if net.seperate_all:
            bb, nc, no, nf = net.bbox_dim, net.class_dim, net.objectness_dim, net.objfeat_dim
            emb = pool.get(M, D)
            H = len(e.enc_heads)
            h1 = pool.get(M, H * D)
            # layer 1: tiny K on un-aligned column slices -- all heads in ONE launch (round 4: three launches of 22-40 us, each too
            # short to fill the chip; bit-identical results)
            items = (_lib.SmallKItem * H)()
            for i, (seq, c0, k) in enumerate(e.enc_heads):
                xs = self.x_in[:, c0:c0 + k]
                items[i].x, items[i].ldx, items[i].k_in = xs.data_ptr(), self.x_in.stride(0), k
                items[i].w, items[i].ldw, items[i].bias = seq[0].weight.data_ptr(), k, seq[0].bias.data_ptr()
                items[i].y, items[i].ldy = h1.data_ptr() + 4 * i * D, H * D
            self.call("dsc_linear_smallk_grouped_f32", items, H, M, D, ACT_GELU,
                      keep=(items, h1, [(seq[0].weight, seq[0].bias) for seq, _, _ in e.enc_heads]))
            h2 = pool.get(M, H * 2 * D)
            self.gemm_batched(h1[:, :D], e.enc_w2[:2 * D], h2[:, :2 * D], e.enc_b2[:2 * D], H, D, 2 * D * D, 2 * D, 2 * D,
                              act_out=ACT_GELU)
            pool.put(h1)
            self.gemm(h2, e.enc_w3, emb, e.enc_b3)                    # sum over the heads = one K-concatenated product
            pool.put(h2)
            x = self.gemm(emb, net.init_conv.weight, pool.get(M, D), net.init_conv.bias)
            pool.put(emb)
        else:
            x = pool.get(M, D)
            k = net.channels
            self.call("dsc_linear_smallk_f32", self.x_in.data_ptr(), self.x_in.stride(0), k,
                      net.init_conv.weight.data_ptr(), k, net.init_conv.bias.data_ptr(), x.data_ptr(), D, M, D,
                      ACT_NONE, keep=(x, net.init_conv.weight, net.init_conv.bias))
        r = x           # `r = x.clone()` of the reference: the buffer is simply never recycled
        skips = []
        text = net.text_condition

        def step(rb, xin, x2, ss_pair, free_in=True):
            out = self.resblock(rb, xin, x2, *ss_pair)
            if free_in and xin is not r and all(xin is not s for s in skips):
                pool.put(xin)
            return out

        for lvl in net.downs:
            b0, b1, ac, b2, la, down = lvl
            x = step(b0, x, None, self.c_ss(b0))
            x = step(b1, x, None, self.t_ss(b1))
            skips.append(x)
            if text:
                x = self.crossattn(ac, x)           # x (a skip) stays alive
            x = step(b2, x, None, self.t_ss(b2))
            xo = self.linattn(la, x)
            pool.put(x)
            x = xo
            skips.append(x)
            if isinstance(down, torch.nn.Conv1d):
                x = self.gemm(x, down.weight, pool.get(M, D), down.bias)
        x = step(net.mid_block0, x, None, self.c_ss(net.mid_block0))
        x = step(net.mid_block1, x, None, self.t_ss(net.mid_block1))
        if text:
            xo = self.crossattn(net.mid_attn_cross, x)
            pool.put(x)
            x = xo
        xo = self.fullattn(net.mid_attn, x)
        pool.put(x)
        x = step(net.mid_block2, xo, None, self.t_ss(net.mid_block2))
        for lvl in net.ups:
            b0, b1, ac, b2, la, up = lvl
            x = step(b0, x, None, self.c_ss(b0))
            s = skips.pop()
            xo = self.resblock(b1, x, s, *self.t_ss(b1))
            pool.put(x)
            pool.put(s)
            x = xo
            if text:
                xo = self.crossattn(ac, x)
                pool.put(x)
                x = xo
            s = skips.pop()
            xo = self.resblock(b2, x, s, *self.t_ss(b2))
            pool.put(x)
            pool.put(s)
            xo2 = self.linattn(la, xo)
            pool.put(xo)
            x = xo2
            if isinstance(up, torch.nn.Conv1d):
                xo = self.gemm(x, up.weight, pool.get(M, D), up.bias)
                pool.put(x)
                x = xo
        xo = self.resblock(net.final_res_block, x, r, *self.t_ss(net.final_res_block))
        pool.put(x)
        x = xo
        # ---- output heads, written straight into the (M, C) output at their column offsets -----------
        if net.seperate_all:
            Hd = len(e.dec_heads)
            d1 = self.gemm(x, e.dec_w1, pool.get(M, Hd * 2 * D), e.dec_b1, act_out=ACT_GELU)
            d2 = pool.get(M, Hd * D)
            self.gemm_batched(d1[:, :2 * D], e.dec_w2[:D], d2[:, :D], e.dec_b2[:D], Hd, 2 * D, D * 2 * D, D, D, act_out=ACT_GELU)
            pool.put(d1)
            # layer 3: the narrow output projections (8 / 25 / 32 columns) as ONE batched launch on weights zero-padded to DEC_PAD rows
            # per head, into a padded [M, Hd * DEC_PAD] buffer, then one gather into the heads' columns of the (M, C) output (round 4:
            # three launches of 25-28 us for 0.05 % of the flops; the products of the valid rows are unchanged: same kernel, same tile)
            P = e.dec_pad
            pad = pool.get(M, Hd * P)
            self.gemm_batched(d2[:, :D], e.dec_w3p[:P], pad[:, :P], e.dec_b3p[:P], Hd, D, P * D, P, P)
            spans = (_lib.ColSpan * Hd)()
            col = 0
            for i, (seq, width) in enumerate(e.dec_heads):
                spans[i].src_col, spans[i].dst_col, spans[i].width = i * P, col, width
                col += width
            self.call("dsc_gather_columns_f32", self.out.data_ptr(), self.out.stride(0), pad.data_ptr(), pad.stride(0), M, spans, Hd,
                      keep=(spans, pad))
            pool.put(pad)
            pool.put(d2)
        else:
            self.gemm(x, net.final_conv.weight, self.out, net.final_conv.bias)

    def run(self):
        self.check_current()
        s = ops.stream_ptr()
        for f, a in self.steps:
            rc = f(*a, s)
            if rc:
                _lib.check(rc, f.__name__)


class DenoiserEngine:
    """Owns derived weights and plans of one Unet1D on one device."""

    def __init__(self, net, device):
        _lib.load()
        if device.type != "cuda":
            raise RuntimeError("diffuscene_amd: the denoiser runs on a HIP device only (got %s); there is no CPU "
                               "fallback -- use the oracle in tests" % device)
        self.net, self.device = net, device
        self.plans = {}
        self.generation = 0           # bumped whenever a parameter's storage moved (refresh): plans of an older generation are stale
        self.sig = None
        self.ws = {}
        # split-bf16 GEMM path (csrc/gemm_split.hip): bf16 planes of every weight a plan multiplies with, re-split by refresh()
        # whenever the parameters change.  With the exact-f32 arithmetic selected (_lib.split_enabled() False: DSC_GEMM=f32 or
        # set_gemm_arithmetic("f32")) no planes are made; Unet1D.engine() rebuilds the engine when the switch has moved since.
        self.split = _lib.split_enabled()
        self.mode = _lib.gemm_mode()             # Unet1D.engine() rebuilds the engine when the kernel switches have moved since
        self._planes = {}
        ws_mods, t_blocks, c_blocks = [], [], []
        for rb, kind in net.resblocks_in_order():
            ws_mods += [rb.block1.proj, rb.block2.proj]
            (t_blocks if kind == "t" else c_blocks).append(rb)
        self.ws_mods, self.t_blocks, self.c_blocks = ws_mods, t_blocks, c_blocks
        self.t_index = {id(rb): i for i, rb in enumerate(t_blocks)}
        self.c_index = {id(rb): i for i, rb in enumerate(c_blocks)}
        self.ws_out = [torch.empty((m.weight.shape[0], m.weight.shape[1]), device=device) for m in ws_mods]
        self.ws = {id(m): o for m, o in zip(ws_mods, self.ws_out)}
        emb = t_blocks[0].mlp[1].weight.shape[1]
        self.t_pack_w = torch.empty((len(t_blocks) * 2 * D, emb), device=device)
        self.t_pack_b = torch.empty((len(t_blocks) * 2 * D,), device=device)
        if c_blocks and c_blocks[0].mlp is not None:
            cdim = c_blocks[0].mlp[1].weight.shape[1]
            self.c_pack_w = torch.empty((len(c_blocks) * 2 * D, cdim), device=device)
            self.c_pack_b = torch.empty((len(c_blocks) * 2 * D,), device=device)
        else:
            self.c_pack_w = self.c_pack_b = None
        self.time_table = net.time_table.to(device)
        self.time_freq = net.time_freq.to(device)
        # The per-attribute encoder / decoder MLPs (_encoder_mlp / _decoder_mlp, denoise_net.py:484-504) are independent
        # three-layer stacks of identical shape.  One launch per layer for ALL heads instead of one per head: more tiles per
        # launch than the CUs hold at once, so the prologue / epilogue of one tile runs under the K loop of another.
        #   encoders: layer 2 as a batched GEMM (weights [H][1024][512]), layer 3 as ONE GEMM over the concatenated hidden
        #             (sum over heads = K concatenation, weights [512][H*1024], biases summed)
        #   decoders: layer 1 as ONE GEMM (same input: weights stacked [H*1024][512]), layer 2 batched ([H][512][1024]); layer 3
        #             stays one narrow GEMM per head (a block-diagonal [C_out][H*512] form triples the serial K loop of a
        #             latency-bound launch)
        self.enc_heads, self.dec_heads = [], []
        if net.seperate_all:
            bb, nc, no, nf = net.bbox_dim, net.class_dim, net.objectness_dim, net.objfeat_dim
            self.enc_heads = [(net.class_embedf, bb, nc), (net.bbox_embedf, 0, bb)]
            self.dec_heads = [(net.bbox_hidden2output, bb), (net.class_hidden2output, nc)]
            if no > 0:
                self.enc_heads.append((net.objectness_embedf, bb + nc, no))
                self.dec_heads.append((net.objectness_hidden2output, no))
            if nf > 0:
                self.enc_heads.append((net.objfeat_embedf, bb + nc + no, nf))
                self.dec_heads.append((net.objfeat_hidden2output, nf))
            H, Hd = len(self.enc_heads), len(self.dec_heads)
            self.enc_w2 = torch.empty((H * 2 * D, D), device=device)
            self.enc_b2 = torch.empty((H * 2 * D,), device=device)
            self.enc_w3 = torch.empty((D, H * 2 * D), device=device)
            self.enc_b3 = torch.empty((D,), device=device)
            self.dec_w1 = torch.empty((Hd * 2 * D, D), device=device)
            self.dec_b1 = torch.empty((Hd * 2 * D,), device=device)
            self.dec_w2 = torch.empty((Hd * D, 2 * D), device=device)
            self.dec_b2 = torch.empty((Hd * D,), device=device)
            self.dec_pad = DEC_PAD * ((max(width for _, width in self.dec_heads) + DEC_PAD - 1) // DEC_PAD)
            self.dec_w3p = torch.zeros((Hd * self.dec_pad, D), device=device)      # output projections, zero-padded to dec_pad rows each
            self.dec_b3p = torch.zeros((Hd * self.dec_pad,), device=device)

    def planes_of(self, w, layout=0):
        """bf16 planes (3, n, K) of a weight the plans multiply with, in ``layout`` (ops.PLANES_ROWMAJOR / PLANES_FRAGMENT: what the
        launch's kernel reads, ops.planes_layout); None where the split path does not apply.  The entry is split on registration --
        the derived weights are current whenever a plan is being built -- and again by every refresh()."""
        if not self.split:
            return None
        w2 = ops.as2d(w.detach() if w.requires_grad else w)
        if w2.dim() != 2 or w2.stride(1) != 1 or not ops.planes_wanted(w2.shape[0], w2.shape[1]):
            return None
        key = (w2.data_ptr(), tuple(w2.shape), w2.stride(0), layout)
        ent = self._planes.get(key)
        if ent is None:
            planes = torch.empty((3,) + tuple(w2.shape), device=self.device, dtype=torch.int16)
            with torch.no_grad():
                ops.split_planes([(w2, planes, 2 * layout)])
            ent = self._planes[key] = (w2, planes, 2 * layout)
        return ent[1]

    def _signature(self):
        """Order-sensitive fingerprint of (version, pointer) of every parameter + the epoch counter of raw-pointer
        optimizer updates (optim.FusedAdam also bumps p._version; the counter covers updates that could not)."""
        from .optim import weights_epoch
        return hash((weights_epoch(),) + tuple((p._version, p.data_ptr()) for p in self.net.parameters()))

    def params_moved(self):
        """True (and every existing plan is marked stale) when a parameter lives in other storage than at the last refresh()."""
        ptrs = tuple(p.data_ptr() for p in self.net.parameters())
        if getattr(self, "_ptrs", None) is not None and self._ptrs != ptrs:
            self.refresh()
            return True
        return False

    def refresh(self, force=False):
        """Re-derive standardised / packed weights if any parameter changed (in-place update or reallocation)."""
        sig = self._signature()
        if not force and sig == self.sig:
            return
        ptrs = tuple(p.data_ptr() for p in self.net.parameters())
        if getattr(self, "_ptrs", None) != ptrs:
            if getattr(self, "_ptrs", None) is not None:
                self.generation += 1    # plans (and graphs captured from them) still held elsewhere now refuse to run
            self.plans.clear()          # plans hold raw parameter pointers
            self._planes.clear()
            self._ptrs = ptrs
        with torch.no_grad():
            ops.weight_standardize([m.weight for m in self.ws_mods], self.ws_out, 1e-5)
            for i, rb in enumerate(self.t_blocks):
                self.t_pack_w[i * 2 * D:(i + 1) * 2 * D].copy_(rb.mlp[1].weight)
                self.t_pack_b[i * 2 * D:(i + 1) * 2 * D].copy_(rb.mlp[1].bias)
            if self.c_pack_w is not None:
                for i, rb in enumerate(self.c_blocks):
                    self.c_pack_w[i * 2 * D:(i + 1) * 2 * D].copy_(rb.mlp[1].weight)
                    self.c_pack_b[i * 2 * D:(i + 1) * 2 * D].copy_(rb.mlp[1].bias)
            if self.enc_heads:
                H2 = 2 * D
                for i, (seq, _, _) in enumerate(self.enc_heads):
                    self.enc_w2[i * H2:(i + 1) * H2].copy_(seq[2].weight.view(H2, D))
                    self.enc_b2[i * H2:(i + 1) * H2].copy_(seq[2].bias)
                    self.enc_w3[:, i * H2:(i + 1) * H2].copy_(seq[4].weight.view(D, H2))
                self.enc_b3.copy_(torch.stack([seq[4].bias for seq, _, _ in self.enc_heads]).sum(0))
                for i, (seq, width) in enumerate(self.dec_heads):
                    self.dec_w1[i * H2:(i + 1) * H2].copy_(seq[0].weight.view(H2, D))
                    self.dec_b1[i * H2:(i + 1) * H2].copy_(seq[0].bias)
                    self.dec_w2[i * D:(i + 1) * D].copy_(seq[2].weight.view(D, H2))
                    self.dec_b2[i * D:(i + 1) * D].copy_(seq[2].bias)
                    self.dec_w3p[i * self.dec_pad:i * self.dec_pad + width].copy_(seq[4].weight.view(width, D))
                    self.dec_b3p[i * self.dec_pad:i * self.dec_pad + width].copy_(seq[4].bias)
            if self._planes:
                ops.split_planes(list(self._planes.values()))
        self.sig = sig
        self._ss_table_sig = None          # the per-timestep table is stale now (recomputed in place on demand)

    def ss_table(self):
        """(scale, shift) of all 19 time-conditioned blocks for every tabulated timestep: [T, 19*1024].  Row t is what the
        per-step time MLP produces for t (same kernels, rows are independent), so sampling results do not change."""
        T = self.time_table.shape[0]
        if getattr(self, "_ss_table", None) is None:
            self._ss_table = torch.empty((T, self.t_pack_w.shape[0]), device=self.device)
            self._ss_table_sig = None
        if self._ss_table_sig != self.sig:
            net = self.net
            with torch.no_grad():
                t1 = ops.gemm(self.time_table, net.time_mlp[1].weight, net.time_mlp[1].bias, act_out=ACT_GELU, row_invariant=True)
                t2 = ops.gemm(t1, net.time_mlp[3].weight, net.time_mlp[3].bias, act_out=ACT_SILU, row_invariant=True)
                ops.gemm(t2, self.t_pack_w, self.t_pack_b, out=self._ss_table, row_invariant=True)
            self._ss_table_sig = self.sig
        return self._ss_table

    def plan_for(self, B, N, ctx_mode, ctx_dim, L, text_dim, time_table=False, slot=0):
        key = (B, N, ctx_mode, ctx_dim, L, text_dim, time_table) + ((slot,) if slot else ())
        p = self.plans.get(key)
        if p is None:
            if N > _lib.MAX_TOKENS_PER_SCENE:
                raise RuntimeError("diffuscene_amd: at most %d objects per scene are supported by the fused "
                                   "GroupNorm / attention kernels (got %d)" % (_lib.MAX_TOKENS_PER_SCENE, N))
            p = Plan(self, B, N, ctx_mode, ctx_dim, L, text_dim, time_table)
            self.plans[key] = p
        return p

    @torch.no_grad()
    def prepare(self, B, N, context, context_cross, refresh=True, time_table=False, slot=0):
        """Select / build the plan for this signature and upload the step-invariant conditioning.
        time_table=True (reverse loops: integer timesteps below the table size) replaces the time MLP by a table."""
        if refresh:
            self.refresh()
        if time_table:
            self.ss_table()
        ctx_mode, ctx_dim = SS_NONE, 0
        if context is not None and self.c_pack_w is not None:
            ctx_dim = context.shape[-1]
            # instance embeddings broadcast over the batch (diffusion_scene_layout_ddpm.py:174-175) arrive as a
            # stride-0 expand: only N rows go through the 9 context MLPs instead of B*N
            ctx_mode = SS_PER_SLOT if (context.stride(0) == 0 or B == 1) else SS_PER_TOKEN
        L = text_dim = 0
        if context_cross is not None and self.net.text_condition:
            L, text_dim = context_cross.shape[1], context_cross.shape[2]
        p = self.plan_for(B, N, ctx_mode, ctx_dim, L, text_dim, time_table, slot)
        if ctx_mode != SS_NONE:
            p.ctx_in.copy_(context[0] if ctx_mode == SS_PER_SLOT else context.reshape(B * N, ctx_dim))
        if L:
            p.cross_in.copy_(context_cross.reshape(B * L, text_dim))
        if p.pre_steps:
            p.run_pre()                 # loop-invariant launches of a sampling plan (after refresh(): weights and planes are current)
        return p

    @torch.no_grad()
    def forward(self, x, t, context, context_cross, clone_out=True, refresh=True):
        """x (B,N,C) fp32, t (B,) int64, context (B,N,ctx)|None, context_cross (B,L,text)|None -> (B,N,C)."""
        B, N, Cc = x.shape
        p = self.prepare(B, N, context, context_cross, refresh)
        p.x_in.copy_(x.reshape(B * N, Cc))
        p.t_in.copy_(t)
        p.run()
        out = p.out.view(B, N, -1)
        return out.clone() if clone_out else out
