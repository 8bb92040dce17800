"""Static training plan: the whole DDPM training step of the denoiser -- q_sample, Unet1D forward (reference
denoise_net.py:507-593), the p_losses objective (diffusion_ddpm.py:520-652) and the hand-written backward of every layer --
flattened ONCE into two launch lists over statically allocated buffers.

What this replaces: round 1 ran the backward through ``torch.autograd`` (autograd_ops.py: one Function per fused op).  That
path allocates every activation and gradient per step, accumulates multi-consumer gradients with ATen adds, concatenates
the packed conditioning weights with ``torch.cat`` and leaves ~135 fills / 65 adds / 24 cats per step between the HIP
kernels.  Here

* every buffer is allocated when the plan is built; a step is a loop of C-ABI calls with fixed pointers, so the whole step
  (forward + loss + backward + gradient norm) is captured in ONE hipGraph on a single GPU;
* weight gradients are written by the TN GEMMs directly into the flat gradient buffer ``G`` (flat.py); the packed time /
  context MLP weights and their gradients are views of ``P`` / ``G``;
* gradients of multi-consumer activations (skip connections, residuals) are accumulated by the GEMM epilogue
  (``residual == y``) or aliased, not by separate add kernels, wherever the dataflow allows;
* under data parallelism the plan knows which launch finishes which slice of ``G`` and hands finished buckets to the reducer
  while the rest of the backward is still being enqueued.

The plan is written against a tiny op vocabulary (``gemm``, ``gemm_gn``, ``gemm_tn``, ``gn_bwd`` ...) that a backend lowers.
``HipBackend`` lowers to libdiffuscene_hip.so calls (the product).  tests/plan_sim.py provides a CPU backend of the same
vocabulary on torch ops so that the plan's dataflow (buffer aliasing, accumulation order, slices) is verified against
torch.autograd without a GPU; the product never imports it.
"""
import ctypes as C
import os

import torch

from ._lib import ACT_GELU, ACT_NONE, ACT_SILU, SS_NONE, SS_PER_SCENE, SS_PER_SLOT, SS_PER_TOKEN

D = 512
HID = 128


def _as2d(w):
    return w.view(w.shape[0], w.shape[1]) if w.dim() == 3 else w


class H:
    """Activation handle: forward tensor + (build-time) gradient tensor."""
    __slots__ = ("t", "g", "ng", "act_of", "g_pre", "uses", "gn_of", "gn_part")

    def __init__(self, t, needs_grad=True):
        self.t, self.g, self.ng = t, None, needs_grad
        self.uses = 0             # consumers counted while the forward list is built (TrainPlan._use)
        self.act_of = None        # (pre-activation tensor, kind) when this handle is act(pre-activation) with ONE consumer
        self.g_pre = None         # gradient w.r.t. the pre-activation, written directly by the consumer's input-gradient GEMM
        self.gn_of = None         # dict(z, gamma, beta, ss, ss_mode, dss, n_tok) when this handle is the output of a fused Block (conv_gn) with ONE consumer
        self.gn_part = None       # [scenes][3 D] partial sums written by the consumer's input-gradient GEMM together with g_pre


# ======================================================================================================================
def tn_token_slices(groups, tile_n, blocks_target, tile_k=128):
    """Token slices of one grouped weight-gradient launch.  groups: (m tokens, n, k) per layer; tile_n x 128 output tiles;
    blocks_target: how many blocks the long tiles should make (rounds x resident blocks).  -> (splits, m_ref).

    m_ref is the token length that CARRIES the launch -- the one with the most tiles x tokens of work -- and the groups at least half
    that long are the "long" ones whose tiles set the duration.  (Not the LONGEST group: with text conditioning the few k / v
    projections over B x L text tokens are longer than the B x N object tokens of every other layer; slicing by them cut a text step's
    850 bulk tiles into 32 slices of 48 tokens each -- 3.7 GB of slabs per step, 5.0 of its 12.2 ms.)  A slice is never shorter than
    256 tokens of the bulk: below that the slab traffic outweighs the parallelism."""
    def tiles(n, k):
        return ((n + tile_n - 1) // tile_n) * ((k + tile_k - 1) // tile_k)
    work = {}
    for m, n, k in groups:
        work[m] = work.get(m, 0) + tiles(n, k) * m
    m_ref = max(work, key=lambda m: (work[m], m))
    long_tiles = sum(tiles(n, k) for m, n, k in groups if 2 * m >= m_ref)
    return max(1, min(32, m_ref // 256, -(-blocks_target // max(long_tiles, 1)))), m_ref


def long_k_splits(K):
    """Slabs of ONE split-K launch over a long reduction (dsc_gemm_splitk_f32: K % (32 * splits) == 0, splits <= 64): the most slabs that
    keep a slab >= 256 terms -- short fp32 chains, one launch.  0 = the shape does not fit (the caller accumulates in chunks)."""
    return next((s for s in range(64, 1, -1) if K % (32 * s) == 0 and K // s >= 256), 0)


def tn_block_map(groups, n_xcd=8, tile_k=128):
    """Block placement of one grouped weight-gradient launch on the split-bf16 kernel (256 x tile_k tiles: 128, or 256 for the round-6 body).
    groups: (m tokens, n, k) per group, in table order.  -> list of (group, tile of the group) per PHYSICAL block id, (-1, -1) = idle.

    Consecutive workgroup ids are dealt round-robin over the XCDs (id % 8), each with its own L2, and the tiles of one layer share its
    operand strips -- so every "long" group (one that carries the launch: at least half the reference token length of tn_token_slices)
    goes WHOLE to one XCD, longest first, always to the XCD with the least work so far; the tiles of the remaining groups (the packed
    time-MLP gradient: 1216 tiles over 256 tokens; tiny conditioning layers) are dealt one by one to the SHORTEST list (their token work is negligible, padding
    blocks are not), behind the long ones.  Every XCD then walks its list in order: physical block b = 8 * position + xcd."""
    def tiles(n, k):
        return ((n + 255) // 256) * ((k + tile_k - 1) // tile_k)
    _, m_ref = tn_token_slices(groups, 256, 768)
    lists = [[] for _ in range(n_xcd)]
    work = [0] * n_xcd                                   # token steps queued per XCD
    order = sorted(range(len(groups)), key=lambda i: (-groups[i][0], -tiles(groups[i][1], groups[i][2]), i))
    short = []
    for gi in order:
        m, n, k = groups[gi]
        nt = tiles(n, k)
        if 2 * m >= m_ref and nt <= 64:
            x = min(range(n_xcd), key=lambda j: (work[j], j))
            lists[x].extend((gi, t) for t in range(nt))
            work[x] += nt * m
        else:
            short.append(gi)
    for gi in short:                                     # (by list LENGTH: their token work is negligible, idle padding is not)
        m, n, k = groups[gi]
        for t in range(tiles(n, k)):
            x = min(range(n_xcd), key=lambda j: (len(lists[j]), j))
            lists[x].append((gi, t))
    depth = max(len(l) for l in lists)
    out = []
    for pos in range(depth):
        for x in range(n_xcd):
            out.append(lists[x][pos] if pos < len(lists[x]) else (-1, -1))
    return out


class HipBackend:
    """Lowers plan ops to (cfunc, args) launches of libdiffuscene_hip.so with pointers fixed at build time."""

    name = "hip"

    def __init__(self, device):
        from . import _lib
        _lib.load()
        self.lib = _lib
        self.device = device
        self.keep = []
        self.scratch_floats = 0
        self.scratch = None
        self._scratch_users = []
        # split-bf16 GEMM path (csrc/gemm_split.hip): bf16 planes of every weight operand of a large product; the plan inserts the
        # launches that refresh them (once per step and phase) -- the exact-f32 arithmetic (_lib.split_enabled() False) makes none
        self.split = _lib.split_enabled()
        self._planes = {}
        self._planes_new = []
        self.plane_bytes = 0
        # Transposed weight copies (operands of the input-gradient GEMMs): the plan registers buffer -> source, and the planes of such
        # a buffer are split STRAIGHT from the source with the transpose folded in (dsc_split_bf16x3_f32, transpose = 1).  The f32
        # transpose itself is only launched for buffers some launch reads as f32 (`f32_reads`: products on the exact-f32 kernel)
        self.transposed_of = {}                  # data_ptr of a dense [K][n] buffer -> source weight [n][K]
        self.f32_reads = set()                   # storage pointers of weight operands read WITHOUT planes

    def planes_of(self, w, rows, layout=0):
        """bf16 planes of weight operand ``w`` ([n][K], rows contiguous) for a product with ``rows`` activation rows, in ``layout``
        (ops.planes_layout of the launch: row-major for the block-staged split kernel, fragment-major for the wave-autonomous one), or None."""
        from . import ops
        if not self.split or rows < 256 or w.dim() != 2 or w.stride(1) != 1 or not ops.planes_wanted(w.shape[0], w.shape[1]):
            return None
        key = (w.data_ptr(), tuple(w.shape), w.stride(0), layout)
        ent = self._planes.get(key)
        if ent is None:
            planes = torch.empty((3,) + tuple(w.shape), device=self.device, dtype=torch.int16)
            self.plane_bytes += planes.numel() * 2
            src = self.transposed_of.get(w.data_ptr())
            if src is not None and (not w.is_contiguous() or tuple(src.shape) != (w.shape[1], w.shape[0])):
                src = None                       # a column slice / padded view of the transposed buffer: split the buffer itself
            if src is None:
                self.f32_reads.add(w.untyped_storage().data_ptr())       # the split launch reads this buffer as f32
            ent = self._planes[key] = (w, planes, src, 2 * layout)
            self._planes_new.append(ent)
        return ent[1]

    def split_steps(self):
        """Launches that (re-)split every weight registered since the last call; the plan places them where those weights are
        final for the step (after the weight standardisation in the forward list, after the transposes in the backward list)."""
        from . import ops
        ents, self._planes_new = self._planes_new, []
        steps = []
        for i in range(0, len(ents), self.lib.WS_MAX):
            part = ents[i:i + self.lib.WS_MAX]
            arr = (self.lib.SplitItem * len(part))()
            for j, (w, planes, src, frag) in enumerate(part):      # frag: 2 = fragment-major output (DSC_SPLIT_FRAGMENT)
                if src is not None:              # planes of w = src^T, read from the source: the f32 copy w is never needed for this
                    ptr, ldw = ops._mat(src, "w")
                    arr[j].w, arr[j].ldw, arr[j].rows, arr[j].cols, arr[j].planes, arr[j].transpose = (
                        ptr, ldw, src.shape[0], src.shape[1], planes.data_ptr(), 1 | frag)
                    continue
                ptr, ldw = ops._mat(w, "w")
                arr[j].w, arr[j].ldw, arr[j].rows, arr[j].cols, arr[j].planes, arr[j].transpose = (
                    ptr, ldw, w.shape[0], w.shape[1], planes.data_ptr(), frag)
            steps.append(self._call("dsc_split_bf16x3_f32", arr, len(part), keep=(arr, part)))
        return steps

    # -- helpers
    def _mat(self, t):
        from .ops import _mat
        return _mat(t, "plan tensor")

    def _call(self, name, *args, keep=()):
        from .engine import _own
        self.keep.append(_own(keep))          # parameters as aliases of their storage: the launch's pointers stay owned by the plan
        return (self.lib.fn(name), args, name)

    def finalize(self):
        """Allocate the shared split-reduction workspace (kernels of one stream run in order, so it is shared)."""
        self.scratch = torch.empty(max(self.scratch_floats, 1), device=self.device, dtype=torch.float32)
        for fix in self._scratch_users:
            fix(self.scratch)

    def _with_scratch(self, floats, build):
        """``build(ptr, floats)`` -> step; the pointer is patched in once the workspace exists."""
        self.scratch_floats = max(self.scratch_floats, int(floats))
        holder = {}

        def fix(scr):
            holder["step"] = build(scr.data_ptr(), scr.numel())
        self._scratch_users.append(fix)
        return holder

    # -- forward ops
    def fuse_act_ok(self, a, w, out, a2=None, bias=None, preact=None, actgrad_x=None):
        """Can this product carry an activation epilogue of the training step (pre-activation also stored / result multiplied by the
        derivative)?  Only the split-bf16 kernel implements them: ask the library whether it takes the launch -- with the argument
        struct of the launch itself (bias / preact / actgrad_x pointers and strides take part in its decision: alignment)."""
        from . import ops
        if not self.split or a.shape[0] < 256 or os.environ.get("DSC_FUSE_ACT", "1") == "0":      # (A/B switch of the measurement tools)
            return False
        g = ops.make_gemm_args(a, w, out, bias, a2, act_out=ACT_GELU if (preact is not None or actgrad_x is not None) else ACT_NONE,
                               preact=preact, actgrad_x=actgrad_x)
        lay = ops.planes_layout(g)
        return lay >= 0 and self.planes_of(w, a.shape[0], lay) is not None

    def gemm(self, a, w, out, bias=None, a2=None, residual=None, act_out=ACT_NONE, preact=None, actgrad_x=None):
        from . import ops
        m, K = a.shape
        if preact is not None or actgrad_x is not None:
            g = ops.make_gemm_args(a, w, out, bias, a2, residual, ACT_NONE, act_out, preact=preact, actgrad_x=actgrad_x)
            lay = ops.planes_layout(g)
            pl = self.planes_of(w, m, lay) if lay >= 0 else None
            if pl is None:
                raise RuntimeError("HipBackend.gemm: an activation epilogue was planned for a launch the split kernel does not take")
            ops.attach_planes(g, pl, layout=lay)
            return self._call("dsc_gemm_f32", C.byref(g), keep=(g, a, w, out, bias, a2, residual, pl, preact, actgrad_x))
        # short, deep products (time / context MLPs: m = B or N rows, K >= 1024): split K over the batch dimension
        tiles = ((m + 63) // 64) * ((out.shape[1] + 63) // 64)        # output tiles: fewer than CUs -> parallelise K instead
        if a2 is None and act_out == ACT_NONE and tiles < 256 and K >= 1024 and (m * out.shape[1]) % 4 == 0:
            splits = next((s for s in (8, 4, 2) if K % (32 * s) == 0 and K // s >= 128), 0)
            if splits:
                fn = self.lib.fn("dsc_gemm_splitk_f32")
                floats = splits * m * out.shape[1]
                g = ops.make_gemm_args(a, w, out, bias, a2, residual, ACT_NONE, act_out)
                from .engine import _own
                self.keep.append(_own((g, a, w, out, bias, residual)))       # the plan owns the storages its raw pointers refer to
                self.f32_reads.add(w.untyped_storage().data_ptr())
                return self._with_scratch(floats, lambda wp, wn: (fn, (C.byref(g), splits, wp, wn), "dsc_gemm_splitk_f32"))
        g = ops.make_gemm_args(a, w, out, bias, a2, residual, ACT_NONE, act_out)
        lay = ops.planes_layout(g)                      # planes only for launches that will use them, in the layout their kernel reads
        pl = self.planes_of(w, m, lay) if lay >= 0 else None
        if pl is not None:
            ops.attach_planes(g, pl, layout=lay)
        else:
            self.f32_reads.add(w.untyped_storage().data_ptr())
        return self._call("dsc_gemm_f32", C.byref(g), keep=(g, a, w, out, bias, a2, residual, pl))

    def _gnbwd_args(self, dy, wt, dz, gn, part):
        from . import ops
        Cc = dz.shape[1]
        # row layout [dbias | dgamma | dbeta] (the order of the parameters in G), as gn_bwd; part None: the question only (any aligned addresses)
        pp, ps = (part.data_ptr(), part.stride(0)) if part is not None else (0x100000, 3 * Cc)
        g = ops.make_gemm_args(dy, wt, dz, gamma=gn["gamma"], beta=gn["beta"], eps=1e-5, tokens_per_scene=gn["n_tok"],
                               scale_shift=gn["ss"], ss_mode=gn["ss_mode"] if gn["ss"] is not None else SS_NONE,
                               gnb=dict(z=gn["z"], dgamma=pp + 4 * Cc, dbeta=pp + 8 * Cc, dbias=pp, pstride=ps, dss=gn["dss"]))
        return g

    def fuse_gnbwd_ok(self, dy, wt, like, gn):
        """Can the input-gradient GEMM dh = dy . W carry the GroupNorm backward of the Block that produced h in its epilogue (round 6: the
        wave-autonomous kernel's row-layout epilogue; include/diffuscene_hip.h, gnb_*)?  The library decides, on a struct of the launch's
        shape (``like``: a tensor of the output's shape and alignment -- nothing is allocated for the question)."""
        from . import ops
        if not self.split or os.environ.get("DSC_FUSE_GNBWD", "1") == "0" or gn["ss_mode"] not in (SS_NONE, SS_PER_SCENE):
            return False
        g = self._gnbwd_args(dy, wt, like, gn, None)
        return ops.planes_layout(g) == ops.PLANES_FRAGMENT and self.planes_of(wt, dy.shape[0], ops.PLANES_FRAGMENT) is not None

    def gemm_gnbwd(self, dy, wt, dz, gn, part):
        from . import ops
        g = self._gnbwd_args(dy, wt, dz, gn, part)
        lay = ops.planes_layout(g)
        pl = self.planes_of(wt, dy.shape[0], lay) if lay >= 0 else None
        if pl is None:
            raise RuntimeError("HipBackend.gemm_gnbwd: planned for a launch the wave-autonomous kernel does not take")
        ops.attach_planes(g, pl, layout=lay)
        return self._call("dsc_gemm_f32", C.byref(g), keep=(g, dy, wt, dz, gn["z"], gn["gamma"], gn["beta"], gn["ss"], gn["dss"], part, pl))

    def gemm_long_k(self, a, w, out, residual=None):
        """out = a @ w^T (+ residual) for a LONG reduction (K > 4096) with few output tiles: one dsc_gemm_splitk_f32 launch with the most
        slabs (<= 64) that keep a slab >= 256 terms -- or None when the shape does not fit that entry point (the caller chunks)."""
        from . import ops
        m, K = a.shape
        n = out.shape[1]
        if (m * n) % 4 or K % 32:
            return None
        splits = long_k_splits(K)
        if not splits:
            return None
        fn = self.lib.fn("dsc_gemm_splitk_f32")
        g = ops.make_gemm_args(a, w, out, None, None, residual, ACT_NONE, ACT_NONE)
        from .engine import _own
        self.keep.append(_own((g, a, w, out, residual)))
        self.f32_reads.add(w.untyped_storage().data_ptr())
        floats = splits * m * n
        return self._with_scratch(floats, lambda wp, wn: (fn, (C.byref(g), splits, wp, wn), "dsc_gemm_splitk_f32"))

    def gemm_gn(self, a, w, out, bias, gamma, beta, n_tok, a2=None, ss=None, ss_mode=SS_NONE, residual=None, preact=None):
        from . import ops
        g = ops.make_gemm_args(a, w, out, bias, a2, residual, gamma=gamma, beta=beta, eps=1e-5, tokens_per_scene=n_tok,
                               scale_shift=ss, ss_mode=ss_mode if ss is not None else SS_NONE, preact=preact)
        lay = ops.planes_layout(g, gn=True)
        pl = self.planes_of(w, a.shape[0], lay) if lay >= 0 else None
        if pl is not None:
            ops.attach_planes(g, pl, gn=True, layout=lay)
        else:
            self.f32_reads.add(w.untyped_storage().data_ptr())
        return self._call("dsc_gemm_gn_silu_f32", C.byref(g), keep=(g, a, w, out, bias, gamma, beta, a2, ss, residual, preact, pl))

    def smallk(self, x, w, bias, out, act_out=ACT_NONE):
        xp, ldx = self._mat(x)
        wp, ldw = self._mat(w)
        yp, ldy = self._mat(out)
        return self._call("dsc_linear_smallk_f32", xp, ldx, x.shape[1], wp, ldw, bias.data_ptr() if bias is not None else None,
                          yp, ldy, x.shape[0], w.shape[0], act_out, keep=(x, w, bias, out))

    def ws(self, weights, outs):
        from . import ops
        steps = []
        for i in range(0, len(weights), self.lib.WS_MAX):
            arr = ops.make_ws_items(list(zip(weights[i:i + self.lib.WS_MAX], outs[i:i + self.lib.WS_MAX])))
            steps.append(self._call("dsc_weight_standardize_f32", arr, len(arr), 1e-5, keep=(arr, weights, outs)))
        return steps

    def time_embedding(self, t, table, freq, out):
        return self._call("dsc_time_embedding_f32", t.data_ptr(), t.shape[0], out.shape[1], table.data_ptr(), table.shape[0],
                          freq.data_ptr(), out.data_ptr(), keep=(t, table, freq, out))

    def act(self, x, out, kind):
        assert x.is_contiguous() and out.is_contiguous()
        return self._call("dsc_activation_f32", x.data_ptr(), out.data_ptr(), x.numel(), kind, keep=(x, out))

    def layernorm(self, x, g, out, residual=None):
        xp, ldx = self._mat(x)
        yp, ldy = self._mat(out)
        rp, ldr = self._mat(residual) if residual is not None else (None, 0)
        return self._call("dsc_layernorm_f32", xp, ldx, g.data_ptr(), rp, ldr, yp, ldy, x.shape[0], x.shape[1], 1e-5,
                          keep=(x, g, out, residual))

    def linattn(self, q, k, v, out, scenes, nq, nk, scale):
        a = []
        for t in (q, k, v, out):
            a += list(self._mat(t))
        return self._call("dsc_linear_attention_f32", *a, scenes, nq, nk, scale, keep=(q, k, v, out))

    def attn(self, q, k, v, out, scenes, n, scale):
        a = []
        for t in (q, k, v, out):
            a += list(self._mat(t))
        return self._call("dsc_attention_f32", *a, scenes, n, scale, keep=(q, k, v, out))

    def q_sample(self, x0, noise, t, sqrt_ac, sqrt_1mac, xt, v):
        b = x0.shape[0]
        return self._call("dsc_q_sample_f32", x0.data_ptr(), noise.data_ptr(), t.data_ptr(), sqrt_ac.data_ptr(),
                          sqrt_1mac.data_ptr(), xt.data_ptr(), v.data_ptr() if v is not None else None, b, x0.numel() // b,
                          int(sqrt_ac.numel()), keep=(x0, noise, t, sqrt_ac, sqrt_1mac, xt, v))

    def loss(self, target, out, x_t, t, tb, ca, cb, bounds, dims, separate, iou, mean_type, losses, parts, dout, scale):
        B, N, Cc = out.shape
        barr = (C.c_float * 12)(*[float(v) for v in bounds]) if bounds is not None else None
        return self._call("dsc_ddpm_loss_f32", target.data_ptr(), out.data_ptr(), x_t.data_ptr(), t.data_ptr(),
                          tb["loss_weight"].data_ptr(), ca.data_ptr() if ca is not None else None,
                          cb.data_ptr() if cb is not None else None, tb["alphas_cumprod"].data_ptr(), barr,
                          losses.data_ptr(), parts.data_ptr(), dout.data_ptr(), B, N, Cc, dims["translation_dim"],
                          dims["size_dim"], dims["bbox_dim"], dims["class_dim"], dims["objectness_dim"], dims["objfeat_dim"],
                          1 if separate else 0, 1 if iou else 0, mean_type, float(scale), int(tb["loss_weight"].numel()),
                          keep=(target, out, x_t, t, tb, ca, cb, barr, losses, parts, dout))

    # -- backward ops
    def gemm_tn(self, a, dy, out, a2=None, kvalid=None, dbias=None):
        ap, lda = self._mat(a)
        dp, ldd = self._mat(dy)
        a2p, lda2, k2 = (None, 0, 0)
        if a2 is not None:
            a2p, lda2 = self._mat(a2)
            k2 = a2.shape[1]
        k1 = a.shape[1]
        kv = (k1 + k2) if kvalid is None else kvalid
        m, n = dy.shape
        op, ldo = self._mat(out)
        wsf = self.lib.fn("dsc_gemm_tn_workspace_floats")(m, n, kv)
        fn = self.lib.fn("dsc_gemm_tn_f32")
        self.keep.append((a, dy, out, a2, dbias))
        dbp = dbias.data_ptr() if dbias is not None else None
        return self._with_scratch(wsf, lambda wp, wn: (fn, (ap, lda, k1, a2p, lda2, k2, dp, ldd, op, ldo, dbp, m, n, kv,
                                                            wp if wsf else None, wn if wsf else 0), "dsc_gemm_tn_f32"))

    def gemm_tn_grouped(self, items):
        """items: dicts(a, dy, out, a2, kvalid, dbias) -> ONE launch (plus the slab reduction when the tokens are split)."""
        import numpy as np
        # longest groups first (the block scheduler hands out tiles in id order): the short ones fill the tail
        items = sorted(items, key=lambda it: -it["dy"].shape[0])
        arr = (self.lib.TnGroup * len(items))()
        tile0, ws_off = 0, 0
        tile0s = 0                                    # 256 x 128 tile numbering of the split-bf16 form
        use_split = self.split
        per_group = []
        shapes = [(it["dy"].shape[0], it["dy"].shape[1], it["a"].shape[1] + (it["a2"].shape[1] if it.get("a2") is not None else 0))
                  for it in items]
        for i, it in enumerate(items):
            a, dy, out, a2, dbias = it["a"], it["dy"], it["out"], it.get("a2"), it.get("dbias")
            ap, lda = self._mat(a)
            dp, ldd = self._mat(dy)
            op, ldo = self._mat(out)
            k1 = a.shape[1]
            a2p, lda2, k2 = (None, 0, 0)
            if a2 is not None:
                a2p, lda2 = self._mat(a2)
                k2 = a2.shape[1]
            m, n = dy.shape
            kv = (k1 + k2) if it.get("kvalid") is None else it["kvalid"]
            # the checks dsc_gemm_tn_f32 makes on the host (the grouped entry point only sees a device table)
            if (k1 & 3) or (k2 & 3) or (n & 3) or (k2 > 0 and k1 % 128) or not (1 <= kv <= k1 + k2):
                raise RuntimeError("gemm_tn_grouped: bad shape m=%d n=%d k1=%d k2=%d kvalid=%d" % (m, n, k1, k2, kv))
            if (ap % 16) or (lda & 3) or (dp % 16) or (ldd & 3) or (k2 and ((a2p % 16) or (lda2 & 3))):
                raise RuntimeError("gemm_tn_grouped: operands must be 16-byte aligned with row strides multiple of 4")
            if a.shape[0] != m or (a2 is not None and a2.shape[0] != m) or tuple(out.shape) != (n, kv):
                raise RuntimeError("gemm_tn_grouped: shape mismatch")
            g = arr[i]
            g.a1, g.lda1, g.k1, g.a2, g.lda2, g.k2 = ap, lda, k1, a2p, lda2, k2
            g.dy, g.ldd, g.out, g.ldo = dp, ldd, op, ldo
            g.dbias = dbias.data_ptr() if dbias is not None else None
            g.m, g.n, g.kvalid, g.tile0 = m, n, kv, tile0
            nt = ((n + 127) // 128) * ((k1 + k2 + 127) // 128)
            tile0 += nt
            g.tile0s = tile0s
            nts = ((n + 255) // 256) * ((k1 + k2 + 127) // 128)
            tile0s += nts
            if m * max(lda, ldd, lda2) * 4 >= 0x7fffffff:
                use_split = False                      # 32-bit byte offsets inside an operand
            per_group.append((n, kv, ldo))
        total = tile0
        if use_split:
            # one 8-wave block per CU: cut the tokens until the long tiles make ~3 rounds of 256 blocks (measured, living80: the three
            # grouped launches of a step 8.35 ms at 8 rounds / 7.9 ms at 3 / 8.6 ms at 2 -- fewer slabs to write and re-read against a
            # longer tail)
            # round 6: 256 x 256 tiles (four 512-register waves, half the staged bytes per MFMA) unless the form says otherwise or a
            # group's K segments do not fall on 256-wide tiles; with half as many tiles the long ones should make ~2 rounds
            tile_k = 256 if (self.lib.fn("dsc_get_tn_split_form")() == 2
                             and all(not (it.get("a2") is not None and it["a"].shape[1] % 256) for it in items)) else 128
            # (living80: 506 long tiles = 1.98 rounds un-sliced -- a target of 2 whole rounds would cut them into two slices with slabs and a
            # reduction launch for 6 missing tiles; arrange: 470 -- hence the margin of 64)
            rounds = int(os.environ.get("DSC_TN_ROUNDS", "3" if tile_k == 128 else "2"))
            splits, _ = tn_token_slices(shapes, 256, rounds * 256 - (64 if tile_k == 256 else 0), tile_k)
            ws_off = 0
            if splits > 1:
                for i, (n, kv, ldo) in enumerate(per_group):
                    if ldo != kv:
                        raise RuntimeError("gemm_tn_grouped: split outputs must be dense [n][kvalid]")
                    arr[i].ws_offset = ws_off
                    ws_off += (n * kv + n) * splits
            table = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(self.device)
            bmap = np.asarray(tn_block_map(shapes, tile_k=tile_k), dtype=np.int32).reshape(-1, 2)
            want_tiles = sum(((n + 255) // 256) * ((k + tile_k - 1) // tile_k) for _, n, k in shapes)
            assert int((bmap[:, 0] >= 0).sum()) == want_tiles, "block map must cover every 256 x %d tile exactly once" % tile_k
            bmap_dev = torch.from_numpy(bmap.copy()).to(self.device)
            self.keep.append((table, items, bmap_dev))
            fn = self.lib.fn("dsc_gemm_tn_grouped_split_f32")
            tp, cnt, bp, nblk = table.data_ptr(), len(items), bmap_dev.data_ptr(), bmap.shape[0]
            return self._with_scratch(ws_off, lambda wp, wn: (fn, (tp, cnt, total, bp, nblk, splits, wp if splits > 1 else None,
                                                                    wn if splits > 1 else 0, ws_off, tile_k), "dsc_gemm_tn_grouped_split_f32"))
        # the long tiles set the duration: cut the tokens until they make ~8 rounds of 2 blocks per CU (tail < 1/8); with all
        # layers of a step in one group that is 2 slabs per gradient instead of the 32 of a per-layer launch
        splits, _ = tn_token_slices(shapes, 128, 8 * 512)
        if splits > 1:
            for i, (n, kv, ldo) in enumerate(per_group):
                if ldo != kv:
                    raise RuntimeError("gemm_tn_grouped: split outputs must be dense [n][kvalid]")
                arr[i].ws_offset = ws_off
                ws_off += (n * kv + n) * splits
        table = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(self.device)
        self.keep.append((table, items))
        fn = self.lib.fn("dsc_gemm_tn_grouped_f32")
        tp, cnt = table.data_ptr(), len(items)
        return self._with_scratch(ws_off, lambda wp, wn: (fn, (tp, cnt, total, splits, wp if splits > 1 else None,
                                                                wn if splits > 1 else 0, ws_off), "dsc_gemm_tn_grouped_f32"))

    def colsum(self, x, out):
        xp, ldx = self._mat(x)
        m, n = x.shape
        fn = self.lib.fn("dsc_colsum_f32")
        self.keep.append((x, out))
        op = out.data_ptr()
        return self._with_scratch(64 * n, lambda wp, wn: (fn, (xp, ldx, m, n, op, wp, wn), "dsc_colsum_f32"))

    def colsum_grouped(self, pairs):
        """pairs: [(x [m, n] with contiguous rows, out [n] contiguous)] -> ONE launch."""
        import numpy as np
        arr = (self.lib.ColsumItem * len(pairs))()
        max_n = 0
        for i, (x, out) in enumerate(pairs):
            xp, ldx = self._mat(x)
            assert out.is_contiguous() and out.numel() == x.shape[1]
            arr[i].x, arr[i].ldx, arr[i].m, arr[i].n, arr[i].out = xp, ldx, x.shape[0], x.shape[1], out.data_ptr()
            max_n = max(max_n, x.shape[1])
        table = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(self.device)
        return self._call("dsc_colsum_grouped_f32", table.data_ptr(), len(pairs), max_n, keep=(table, pairs))

    def gn_bwd(self, z, dy, gamma, beta, ss, ss_mode, dz, part, dss, scenes, n_tok):
        zp, ldz = self._mat(z)
        dp, ldy = self._mat(dy)
        dzp, lddz = self._mat(dz)
        Cc = z.shape[1]
        sp, ld_ss = self._mat(ss) if ss is not None else (None, 0)
        dsp, ld_dss = self._mat(dss) if dss is not None else (None, 0)
        pp = part.data_ptr()                       # row layout [dbias | dgamma | dbeta] (the order of the parameters in G)
        return self._call("dsc_gn_silu_bwd_f32", zp, ldz, dp, ldy, gamma.data_ptr(), beta.data_ptr(), sp, ld_ss,
                          ss_mode if sp else SS_NONE, dzp, lddz, pp + 4 * Cc, pp + 8 * Cc, pp, part.stride(0), dsp, ld_dss,
                          scenes, n_tok, Cc, 1e-5, keep=(z, dy, gamma, beta, ss, dz, part, dss))

    def ws_bwd(self, weights, dws, outs):
        steps = []
        for i in range(0, len(weights), self.lib.WS_MAX):
            ws_, gs, os_ = weights[i:i + self.lib.WS_MAX], dws[i:i + self.lib.WS_MAX], outs[i:i + self.lib.WS_MAX]
            arr = (self.lib.WsBwdItem * len(ws_))()
            for j, (w, g, o) in enumerate(zip(ws_, gs, os_)):
                w2, o2 = _as2d(w), _as2d(o)
                assert w2.is_contiguous() and g.is_contiguous() and o2.is_contiguous()
                arr[j].w, arr[j].dw_std, arr[j].dw = w2.data_ptr(), g.data_ptr(), o2.data_ptr()
                arr[j].rows, arr[j].cols = w2.shape
            steps.append(self._call("dsc_weight_standardize_bwd_f32", arr, len(ws_), 1e-5, keep=(arr, ws_, gs, os_)))
        return steps

    def layernorm_bwd(self, x, g, dy, dx, dg_part, addend=None):
        xp, ldx = self._mat(x)
        dp, ldy = self._mat(dy)
        dxp, lddx = self._mat(dx)
        ap, lda = self._mat(addend) if addend is not None else (None, 0)
        return self._call("dsc_layernorm_bwd_f32", xp, ldx, g.data_ptr(), dp, ldy, dxp, lddx, ap, lda, dg_part.data_ptr(),
                          dg_part.shape[0], x.shape[0], x.shape[1], 1e-5, keep=(x, g, dy, dx, dg_part, addend))

    def linattn_bwd(self, q, k, v, dout, dq, dk, dv, scenes, nq, nk, scale):
        a = []
        for t in (q, k, v, dout, dq, dk, dv):
            a += list(self._mat(t))
        return self._call("dsc_linear_attention_bwd_f32", *a, scenes, nq, nk, scale, keep=(q, k, v, dout, dq, dk, dv))

    def attn_bwd(self, q, k, v, dout, dq, dk, dv, scenes, n, scale):
        a = []
        for t in (q, k, v, dout, dq, dk, dv):
            a += list(self._mat(t))
        return self._call("dsc_attention_bwd_f32", *a, scenes, n, scale, keep=(q, k, v, dout, dq, dk, dv))

    def act_bwd(self, x, dy, dx, kind):
        assert x.is_contiguous() and dy.is_contiguous() and dx.is_contiguous()
        return self._call("dsc_activation_bwd_f32", x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), kind,
                          keep=(x, dy, dx))

    def transpose_many(self, pairs):
        """pairs of (w [r,c] contiguous, out [c,r] contiguous) -> batched launches; strided outputs -> single launches."""
        steps, batch = [], []
        for w, o in pairs:
            if w.is_contiguous() and o.is_contiguous():
                batch.append((w, o))
            else:
                wp, ldi = self._mat(w)
                op, ldo = self._mat(o)
                steps.append(self._call("dsc_transpose_f32", wp, ldi, op, ldo, w.shape[0], w.shape[1], keep=(w, o)))
        for i in range(0, len(batch), self.lib.WS_MAX):
            part = batch[i:i + self.lib.WS_MAX]
            arr = (self.lib.WsItem * len(part))()
            for j, (w, o) in enumerate(part):
                arr[j].w, arr[j].out = w.data_ptr(), o.data_ptr()
                arr[j].rows, arr[j].cols = w.shape
            steps.append(self._call("dsc_transpose_batched_f32", arr, len(part), keep=(arr, part)))
        return steps

    def copy(self, dst, src):
        dp, ldd = self._mat(dst)
        sp, lds = self._mat(src)
        return self._call("dsc_copy2d_f32", dp, ldd, sp, lds, dst.shape[0], dst.shape[1], keep=(dst, src))

    def add(self, dst, src):
        dp, ldd = self._mat(dst)
        sp, lds = self._mat(src)
        return self._call("dsc_add2d_f32", dp, ldd, sp, lds, dst.shape[0], dst.shape[1], keep=(dst, src))

    # -- execution
    def run(self, steps, stream):
        check = self.lib.check
        for st in steps:
            if isinstance(st, dict):
                st = st["step"]
            f, a, name = st
            rc = f(*a, stream)
            if rc:
                check(rc, name)


# ======================================================================================================================
class TrainPlan:
    """Forward + loss + backward launch lists of one (B, N, conditioning) signature of one Unet1D."""

    def __init__(self, net, flat, diff, B, N, ctx_mode, ctx_dim, L, text_dim, backend, per_block_grads=False,
                 ctx_param=None, tables=None, grad_scale=None, tn_flush_floats=None):
        self.net, self.flat, self.diff = net, flat, diff
        self.B, self.N, self.M = B, N, B * N
        self.be = backend
        self.device = backend.device
        self.ctx_mode, self.ctx_dim, self.L, self.text_dim = ctx_mode, ctx_dim, L, text_dim
        self.per_block_grads = per_block_grads
        # Single GPU (None): every weight-gradient GEMM of the step waits for ONE grouped launch at the end of the backward (nothing
        # forces an earlier one since round 4: out-of-place LayerNorm accumulation, column-exact hazard check).  Data parallel: the
        # pending group is launched whenever it holds this many gradient floats, so that buckets of G finish -- and their
        # all-reduces start -- while the rest of the backward is still running (train_step.PlanRunner: a third of G -> 3 launches)
        self.tn_flush_floats = tn_flush_floats
        self.fwd, self.bwd = [], []
        self._tape = []
        self._transposes = []
        self._cur = self.fwd
        self.bwd_writes = []              # (index of the LAST step of a backward op, (G offset, length)) per finished gradient
        self._tn_pending = []             # deferred weight-gradient GEMMs (leaves of the backward): one grouped launch
        self._cs_pending = []             # deferred column sums of per-scene gradient partials: one grouped launch
        self.n_adds = 0
        self.bytes = 0                    # statically allocated activation / gradient bytes (the runner's cache budget)
        C_in = net.channels
        dev = self.device
        # ---- static inputs
        self.x0 = self.new3(B, N, C_in)
        self.noise = self.new3(B, N, C_in)
        self.t = torch.zeros((B,), device=dev, dtype=torch.int64)
        ctx_rows = {SS_NONE: 0, SS_PER_SLOT: N, SS_PER_TOKEN: self.M}[ctx_mode]
        # ctx_param: the context IS a parameter (learnable instance embedding, shared over the batch): read it in place and
        # write its gradient straight into G -- no autograd boundary at all for the shipped unconditional configs
        self.ctx_grad_param = ctx_param
        if ctx_param is not None:
            assert ctx_mode == SS_PER_SLOT and tuple(ctx_param.shape) == (N, ctx_dim)
            self.ctx_in = H(ctx_param.detach())
        else:
            self.ctx_in = H(self.new(ctx_rows, ctx_dim)) if ctx_rows else None
        self.cross_in = H(self.new(B * L, text_dim)) if L else None
        # ---- outputs
        self.x_t = self.new3(B, N, C_in)
        self.out = self.new(self.M, net.out_dim)
        self.dout = self.new(self.M, net.out_dim)
        self.losses = torch.zeros((B,), device=dev)
        self.parts = torch.zeros((B, 9), device=dev)
        self._tables = tables
        # loss = losses.mean() over the (global) batch: d loss / d losses[b] = 1/B, or 1/(B * world) when the ranks' gradients
        # are SUMMED by the all-reduce (the mean over ranks is folded in here instead of a second pass over G)
        self.grad_scale = 1.0 / B if grad_scale is None else float(grad_scale)
        self._build()
        if hasattr(self.be, "finalize"):
            self.be.finalize()
        self.bytes += getattr(self.be, "plane_bytes", 0)          # the plan's own bf16 weight planes count against the cache budget

    # ------------------------------------------------------------------------------------------------ buffers
    def new(self, rows, cols):
        self.bytes += 4 * rows * cols
        return torch.empty((rows, cols), device=self.device, dtype=torch.float32)

    def new3(self, a, b, c):
        self.bytes += 4 * a * b * c
        return torch.empty((a, b, c), device=self.device, dtype=torch.float32)

    def zeros(self, rows, cols):
        self.bytes += 4 * rows * cols
        return torch.zeros((rows, cols), device=self.device, dtype=torch.float32)

    def emit(self, step):
        if isinstance(step, list):
            self._cur.extend(step)
        else:
            self._cur.append(step)

    def gview(self, p):
        return _as2d(self.flat.grad_view(p)) if p.dim() != 1 else self.flat.grad_view(p)

    def wrote(self, *params):
        """Mark parameters whose gradient is final once the steps emitted so far have run."""
        idx = len(self.bwd) - 1
        for p in params:
            if p is not None:
                self.bwd_writes.append((idx, self.flat.grad_range(p)))

    def wrote_range(self, off, n):
        self.bwd_writes.append((len(self.bwd) - 1, (off, n)))

    # ------------------------------------------------------------------------------------------------ gradient plumbing
    @staticmethod
    def _dense_ok(t):
        return t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0

    # Invariant that makes aliasing safe: a gradient tensor is only ever handed to handles whose backward emitters run
    # strictly LATER (they belong to earlier forward ops), and every emitter finishes reading its own dy before it hands dy
    # on.  So accumulating in place into an aliased buffer can never disturb a reader.
    # Deferred weight-gradient GEMMs (self.tn) read their dy LATER, so a gradient buffer that a pending one reads must not be
    # accumulated into before that launch: _guard flushes the pending group first.
    def _guard(self, t):
        """Call before modifying gradient tensor ``t`` in place."""
        if not self._tn_pending:
            return
        key = t.untyped_storage().data_ptr()
        for it in self._tn_pending:
            if it["dy"].untyped_storage().data_ptr() == key and self._may_overlap(t, it["dy"]):
                self.flush_tn()
                return

    @staticmethod
    def _may_overlap(u, v):
        """Two views of ONE storage: False only when they are provably disjoint -- row-major 2-D views with the same row stride whose
        column ranges do not meet (the two halves of a [M, 1024] skip-connection pair buffer: accumulating into one half must not force
        out the weight-gradient GEMMs that read the other)."""
        if u.dim() != 2 or v.dim() != 2 or u.stride(1) != 1 or v.stride(1) != 1 or u.stride(0) != v.stride(0):
            return True
        S = u.stride(0)
        cu, cv = u.storage_offset() % S, v.storage_offset() % S
        if cu + u.shape[1] > S or cv + v.shape[1] > S:
            return True
        return not (cu + u.shape[1] <= cv or cv + v.shape[1] <= cu)

    def g_alias(self, h, src):
        """grad(h) += src where src is a finished gradient tensor: no copy when it is the first contribution."""
        if not h.ng:
            return
        if h.g is None:
            h.g = src
        else:
            self._guard(h.g)
            self.emit(self.be.add(h.g, src))
            self.n_adds += 1

    def g_target(self, h):
        """-> (tensor to write a [rows, cols] contribution into, accumulate?)."""
        if h.g is None:
            h.g = self.new(*h.t.shape)
            return h.g, False
        self._guard(h.g)
        return h.g, True

    def g_write(self, h, produce):
        """Contribution from a kernel that cannot accumulate: ``produce(dst)`` emits it."""
        if not h.ng:
            return
        if h.g is None:
            h.g = self.new(*h.t.shape)
            produce(h.g)
        else:
            tmp = self.new(*h.t.shape)
            produce(tmp)
            self._guard(h.g)
            self.emit(self.be.add(h.g, tmp))
            self.n_adds += 1

    def g_write_add(self, h, produce):
        """Contribution from a kernel that can ADD the gradient ``h`` already holds while it writes its own (``produce(dst, addend)``:
        dst = addend + contribution), OUT OF PLACE: the old buffer is only read -- weight-gradient GEMMs still pending on it need no
        flush, other handles aliasing it keep their (complete) value -- and ``h.g`` becomes the new buffer.  No add launch."""
        if not h.ng:
            return
        old = h.g
        dst = self.new(*h.t.shape)
        produce(dst, old)
        h.g = dst

    @staticmethod
    def _use(*handles):
        """Count a forward consumer of each handle (the activation-derivative fusion of g_gemm needs "exactly one")."""
        for h in handles:
            if h is not None:
                h.uses += 1

    def g_gemm(self, a, a2, dy, wt):
        """grad([a | a2]) += dy @ wt^T-layout weight (wt is [K, n]: the GEMM computes dy . wt^T)."""
        k1 = a.t.shape[1]
        K = wt.shape[0]
        if a2 is None:
            if not a.ng:
                return
            if (a.act_of is not None and a.uses == 1 and a.g is None and a.g_pre is None and dy.shape[1] <= 4096
                    and hasattr(self.be, "fuse_act_ok") and self.be.fuse_act_ok(dy, wt, a.t, actgrad_x=a.act_of[0])):
                # a = act(u) and this GEMM is its ONLY consumer (counted while the forward was built, not inferred from the order of the
                # backward): d u = (dy . W) * act'(u) in the GEMM's epilogue -- the activation's backward launch (read u, read d a,
                # write d u) disappears.  An activation with a second consumer keeps the separate launch.
                u, kind = a.act_of
                a.g_pre = self.new(*a.t.shape)
                self.emit(self.be.gemm(dy, wt, a.g_pre, act_out=kind, actgrad_x=u))
                return
            if (a.gn_of is not None and a.uses == 1 and a.g is None and a.g_pre is None and hasattr(self.be, "fuse_gnbwd_ok")):
                # a = Block(...) output and this GEMM is its ONLY consumer: the gradient w.r.t. the Block's pre-norm activation comes straight
                # out of this GEMM's epilogue (round 6) -- the gn_silu_bwd launch of that Block (read z, read d a, write d z) disappears
                gn = a.gn_of
                if self.be.fuse_gnbwd_ok(dy, wt, a.t, gn):
                    dz = self.new(*a.t.shape)
                    part = self.new(a.t.shape[0] // gn["n_tok"], 3 * a.t.shape[1])
                    a.g_pre, a.gn_part = dz, part
                    self.emit(self.be.gemm_gnbwd(dy, wt, dz, gn, part))
                    return
            dst, acc = self.g_target(a)
            self._gemm_acc(dy, wt, dst, acc)
            return
        want1, want2 = a.ng, a2.ng
        if not (want1 or want2):
            return
        pair_fresh = a.g is None and a2.g is None and want1 and want2
        same_parent = (a.g is not None and a2.g is not None and a.g.stride(0) == K and a2.g.stride(0) == K
                       and a2.g.data_ptr() == a.g.data_ptr() + 4 * k1)
        if pair_fresh:
            buf = self.new(dy.shape[0], K)
            a.g, a2.g = buf[:, :k1], buf[:, k1:]
            self._gemm_acc(dy, wt, buf, False)
        elif same_parent:
            buf = torch.as_strided(a.g, (a.g.shape[0], K), (K, 1))
            self._guard(buf)
            self._gemm_acc(dy, wt, buf, True)
        else:
            tmp = self.new(dy.shape[0], K)
            self._gemm_acc(dy, wt, tmp, False)
            if want1:
                self.g_alias(a, tmp[:, :k1])
            if want2:
                self.g_alias(a2, tmp[:, k1:])

    def _gemm_acc(self, dy, wt, dst, acc):
        n = dy.shape[1]
        if n > 4096 and hasattr(self.be, "gemm_long_k"):
            # long reductions (the packed 19 x 1024 time-MLP outputs, K = 19456; the 9 x 1024 context-MLP outputs): ONE split-K launch
            # whose slabs are <= 512 terms long (blocked sum: the fp32 error of short chains, as the chunked form below) instead of ten
            # chunk launches of 25 us each (round 4)
            step = self.be.gemm_long_k(dy, wt, dst, dst if acc else None)
            if step is not None:
                self.emit(step)
                return
        if n > 4096:
            # long reductions: accumulate in chunks of 2048 so that the fp32 error stays that of a blocked sum instead of one
            # 19456-term sequential chain
            for c0 in range(0, n, 2048):
                c1 = min(c0 + 2048, n)
                self.emit(self.be.gemm(dy[:, c0:c1], wt[:, c0:c1], dst, residual=dst if (acc or c0 > 0) else None))
        else:
            self.emit(self.be.gemm(dy, wt, dst, residual=dst if acc else None))

    def tn(self, a, dy, out, a2=None, kvalid=None, dbias=None, params=(), after=None):
        """Weight gradient out = dy^T [a | a2]: deferred -- it is a leaf of the backward pass and its operands stay alive, so
        all of them run as one grouped launch (flush_tn) instead of one M-split launch + slab reduction per layer."""
        self._tn_pending.append({"a": a, "dy": dy, "out": out, "a2": a2, "kvalid": kvalid, "dbias": dbias,
                                 "params": tuple(p for p in params if p is not None), "after": after})
        if self.tn_flush_floats and sum(it["out"].numel() for it in self._tn_pending) >= self.tn_flush_floats:
            self.flush_tn()

    def colsum_later(self, x, out, params=(), after=None):
        """out = column sums of x, deferred into the next grouped launch (flush_tn): leaves of the backward, like the weight
        gradients.  Backends without a grouped form run it in place."""
        if not hasattr(self.be, "colsum_grouped"):
            self.emit(self.be.colsum(x, out))
            if after is not None:
                after()
            self.wrote(*params)
            return
        self._cs_pending.append({"x": x, "out": out, "params": tuple(params), "after": after})

    def flush_tn(self):
        cs, self._cs_pending = self._cs_pending, []
        if cs:
            self.emit(self.be.colsum_grouped([(it["x"], it["out"]) for it in cs]))
            for it in cs:
                if it["after"] is not None:
                    it["after"]()
            self.wrote(*[p for it in cs for p in it["params"]])
        items, self._tn_pending = self._tn_pending, []
        if not items:
            return
        self.emit(self.be.gemm_tn_grouped(items))
        for it in items:
            if it["after"] is not None:
                it["after"]()
        self.wrote(*[p for it in items for p in it["params"]])

    def wT(self, w2d, npad=None):
        """Transposed copy [K, n(pad)] of a weight, refreshed once per step at the start of the backward pass."""
        n, K = w2d.shape
        if npad is None or npad == n:
            buf = self.new(K, n)
            self._transposes.append((w2d, buf))
            if hasattr(self.be, "transposed_of"):
                self.be.transposed_of[buf.data_ptr()] = w2d
            return buf
        buf = self.zeros(K, npad)
        self._transposes.append((w2d, buf[:, :n]))
        return buf

    # ------------------------------------------------------------------------------------------------ layers
    def linear(self, a, weight, bias, a2=None, residual=None, out=None, act_out=ACT_NONE, wt=None, act=None):
        """y = [a | a2] @ W^T + b (+ residual).  ``out`` may be a column slice of a wider tensor (decoder heads).
        ``act``: returns act(y) instead -- ONE launch when the product runs on the split kernel (the pre-activation y is stored next
        to act(y) by the GEMM's epilogue; its backward needs it), the GEMM followed by an activation launch otherwise."""
        w2 = _as2d(weight)
        n, K = w2.shape
        rows = a.t.shape[0]
        y = H(out if out is not None else self.new(rows, n))
        self._use(a, a2, residual)
        assert act_out == ACT_NONE, "training keeps the pre-activation (its backward needs it): use act="
        # probed with the launch's own operands (bias, the pre-activation buffer): what the library refuses here it would refuse below
        fused = (act is not None and residual is None and out is None and hasattr(self.be, "fuse_act_ok")
                 and self.be.fuse_act_ok(a.t, w2, y.t, a2.t if a2 is not None else None, bias=bias, preact=y.t))
        if fused:
            ya_t = self.new(rows, n)
            self.emit(self.be.gemm(a.t, w2, ya_t, bias, a2.t if a2 is not None else None, None, act, preact=y.t))
        else:
            self.emit(self.be.gemm(a.t, w2, y.t, bias, a2.t if a2 is not None else None,
                                   residual.t if residual is not None else None, act_out))

        def bw():
            dy = y.g
            if dy is None:
                return
            npad = (n + 31) // 32 * 32
            if npad != n:
                dyp = self.zeros(rows, npad)
                self.emit(self.be.copy(dyp[:, :n], dy))
            elif not self._dense_ok(dy):
                dyp = self.new(rows, n)
                self.emit(self.be.copy(dyp, dy))
            else:
                dyp = dy
            if a.ng or (a2 is not None and a2.ng):
                self.g_gemm(a, a2, dyp, wt if wt is not None else self.wT(w2, npad))
            gw = self.gview(weight)
            gb = self.flat.grad_view(bias) if bias is not None else None
            if npad != n:
                tw = self.new(npad, K)
                tb = self.new(1, npad) if bias is not None else None

                def unpad():
                    self.emit(self.be.copy(gw, tw[:n]))
                    if bias is not None:
                        self.emit(self.be.copy(gb.view(1, n), tb[:, :n]))
                self.tn(a.t, dyp, tw, a2.t if a2 is not None else None, dbias=tb.view(-1) if tb is not None else None,
                        params=(weight, bias), after=unpad)
            else:
                self.tn(a.t, dyp, gw, a2.t if a2 is not None else None, dbias=gb, params=(weight, bias))
            if residual is not None:
                self.g_alias(residual, dy)
        self._tape.append(bw)
        if act is None:
            return y
        # the activation's backward is registered AFTER the linear's: the tape runs in reverse, d y must exist before bw() above reads it
        return self.act(y, act, out=ya_t if fused else None)

    def smallk(self, x_slice, weight, bias):
        """First encoder layer on an un-aligned column slice of x_t (no input gradient)."""
        w2 = _as2d(weight)
        n, k = w2.shape
        rows = x_slice.shape[0]
        y = H(self.new(rows, n))
        self.emit(self.be.smallk(x_slice, w2, bias, y.t))
        xpad = self.zeros(rows, 32 if k <= 32 else 64)
        self.emit(self.be.copy(xpad[:, :k], x_slice))          # staged for the weight-gradient MFMA pass

        def bw():
            if y.g is None:
                return
            self.tn(xpad, y.g, self.gview(weight), kvalid=k, dbias=self.flat.grad_view(bias), params=(weight, bias))
        self._tape.append(bw)
        return y

    def act(self, x, kind, out=None):
        """y = act(x).  ``out``: the producing GEMM already wrote act(x) there, next to x (linear(..., act=kind)): no launch."""
        y = H(out if out is not None else self.new(*x.t.shape))
        if out is None:
            self.emit(self.be.act(x.t, y.t, kind))
        y.act_of = (x.t, kind)

        def bw():
            if y.g_pre is not None:
                # the consumer's input-gradient GEMM already applied act'(x) (g_gemm): its output IS d x
                self.g_alias(x, y.g_pre)
                if y.g is None:
                    return
                # (not reachable while g_gemm fuses single-consumer activations only; kept correct rather than asserted: a gradient that
                # arrived un-fused gets act' applied by the separate launch and is added)
            if y.g is None or not x.ng:
                return
            dy = y.g
            if not dy.is_contiguous():
                d = self.new(*dy.shape)
                self.emit(self.be.copy(d, dy))
                dy = d
            self.g_write(x, lambda dst: self.emit(self.be.act_bwd(x.t, dy, dst, kind)))
        self._tape.append(bw)
        return y

    def conv_gn(self, a, blk, a2=None, ss=None, ss_mode=SS_NONE, dss=None, residual=None):
        """Block.forward (:167-176) incl. weight standardisation: one fused GEMM; z = pre-norm output kept for backward."""
        conv, norm = blk.proj, blk.norm
        w_std = self.ws_std[id(conv)]
        rows = a.t.shape[0]
        y, z = H(self.new(rows, D)), self.new(rows, D)
        self._use(a, a2, residual)
        self.emit(self.be.gemm_gn(a.t, w_std, y.t, conv.bias, norm.weight, norm.bias, self.N, a2.t if a2 is not None else None,
                                  ss, ss_mode, residual.t if residual is not None else None, z))

        if residual is None:
            # candidate for the fused backward: if this output ends up with ONE consumer whose input-gradient GEMM runs the wave-autonomous kernel,
            # that GEMM's epilogue does this Block's GroupNorm backward (g_gemm); per-scene or no (scale, shift) only
            y.gn_of = dict(z=z, gamma=norm.weight, beta=norm.bias, ss=ss, ss_mode=ss_mode if ss is not None else SS_NONE,
                           dss=dss if (ss is not None and ss_mode == SS_PER_SCENE) else None, n_tok=self.N)

        def bw():
            dy = y.g
            if dy is None and y.g_pre is None:
                return
            scenes = rows // self.N
            slot_tmp = None
            if y.g_pre is not None:                        # d z and the partial sums were written by the consumer's input-gradient GEMM
                dz, part = y.g_pre, y.gn_part
            else:
                dz = self.new(rows, D)
                part = self.new(scenes, 3 * D)
                dss_arg = dss
                if ss is not None and ss_mode == SS_PER_SLOT:
                    slot_tmp = self.new(rows, 2 * D)          # per-token, reduced over the batch below
                    dss_arg = slot_tmp
                self.emit(self.be.gn_bwd(z, dy, norm.weight, norm.bias, ss, ss_mode, dz, part, dss_arg, scenes, self.N))
            o_b, o_g, o_be = (self.flat.grad_range(p)[0] for p in (conv.bias, norm.weight, norm.bias))
            if o_g == o_b + D and o_be == o_g + D:
                self.colsum_later(part, self.flat.G[o_b:o_b + 3 * D], params=(conv.bias, norm.weight, norm.bias))
            else:
                tmp = self.new(1, 3 * D)

                def scatter(tmp=tmp):
                    for i, p in enumerate((conv.bias, norm.weight, norm.bias)):
                        self.emit(self.be.copy(self.flat.grad_view(p).view(1, D), tmp[:, i * D:(i + 1) * D]))
                self.colsum_later(part, tmp.view(-1), params=(conv.bias, norm.weight, norm.bias), after=scatter)
            if slot_tmp is not None:
                red = self.new(self.N, 2 * D)
                self.emit(self.be.colsum(slot_tmp.view(scenes, self.N * 2 * D), red.view(-1)))
                self.emit(self.be.copy(dss, red))
            self.g_gemm(a, a2, dz, self.ws_t[id(conv)])
            self.tn(a.t, dz, self.dws[id(conv)], a2.t if a2 is not None else None)
            self._ws_pending.append(conv)
            if residual is not None:
                self.g_alias(residual, dy)
        self._tape.append(bw)
        return y

    def resblock(self, rb, a, a2, ss_pair, post=None):
        """ResnetBlock.forward (:190-206).  ``post`` is emitted right after the block's backward (the tape runs in reverse, so
        it is pushed first): data parallelism finishes the block's slices of G there."""
        if post is not None:
            self._tape.append(post)
        ss, mode, dss = ss_pair
        h = self.conv_gn(a, rb.block1, a2=a2, ss=ss, ss_mode=mode, dss=dss)
        r = self.linear(a, rb.res_conv.weight, rb.res_conv.bias, a2=a2) if rb.has_res_conv else a
        return self.conv_gn(h, rb.block2, residual=r)

    def layernorm(self, x, gain, residual=None):
        g = gain.view(-1)
        y = H(self.new(*x.t.shape))
        self._use(x, residual)
        self.emit(self.be.layernorm(x.t, g, y.t, residual.t if residual is not None else None))

        def bw():
            dy = y.g
            if dy is None:
                return
            M_ = x.t.shape[0]
            nblk = min((M_ + 3) // 4, 512)
            part = self.new(nblk, x.t.shape[1])
            self.g_write_add(x, lambda dst, addend: self.emit(self.be.layernorm_bwd(x.t, g, dy, dst, part, addend)))
            self.colsum_later(part, self.flat.grad_view(gain).view(-1), params=(gain,))
            if residual is not None:
                self.g_alias(residual, dy)
        self._tape.append(bw)
        return y

    def linattn(self, blk, a):
        att = blk.fn.fn
        B, N = self.B, self.N
        y = self.layernorm(a, blk.fn.norm.g)
        qkv = self.linear(y, att.to_qkv.weight, None)
        o = H(self.new(self.M, HID))
        q, k, v = qkv.t[:, :HID], qkv.t[:, HID:2 * HID], qkv.t[:, 2 * HID:]
        self.emit(self.be.linattn(q, k, v, o.t, B, N, N, float(att.scale)))

        def bw():
            if o.g is None:
                return
            d = self.new(self.M, 3 * HID)
            qkv.g = d
            self.emit(self.be.linattn_bwd(q, k, v, o.g, d[:, :HID], d[:, HID:2 * HID], d[:, 2 * HID:], B, N, N,
                                          float(att.scale)))
        self._tape.append(bw)
        p = self.linear(o, att.to_out[0].weight, att.to_out[0].bias)
        return self.layernorm(p, att.to_out[1].g, residual=a)

    def crossattn(self, blk, a):
        att = blk.fn.fn
        B, N, L = self.B, self.N, self.L
        y = self.layernorm(a, blk.fn.norm.g)
        qh = self.linear(y, att.to_q.weight, None)
        kv = self.linear(self.cross_in, att.to_kv.weight, None)
        o = H(self.new(self.M, HID))
        self.emit(self.be.linattn(qh.t, kv.t[:, :HID], kv.t[:, HID:], o.t, B, N, L, float(att.scale)))

        def bw():
            if o.g is None:
                return
            dq, dkv = self.new(self.M, HID), self.new(B * L, 2 * HID)
            qh.g, kv.g = dq, dkv
            self.emit(self.be.linattn_bwd(qh.t, kv.t[:, :HID], kv.t[:, HID:], o.g, dq, dkv[:, :HID], dkv[:, HID:], B, N, L,
                                          float(att.scale)))
        self._tape.append(bw)
        p = self.linear(o, att.to_out[0].weight, att.to_out[0].bias)
        return self.layernorm(p, att.to_out[1].g, residual=a)

    def fullattn(self, blk, a):
        att = blk.fn.fn
        B, N = self.B, self.N
        y = self.layernorm(a, blk.fn.norm.g)
        qkv = self.linear(y, att.to_qkv.weight, None)
        o = H(self.new(self.M, HID))
        q, k, v = qkv.t[:, :HID], qkv.t[:, HID:2 * HID], qkv.t[:, 2 * HID:]
        self.emit(self.be.attn(q, k, v, o.t, B, N, float(att.scale)))

        def bw():
            if o.g is None:
                return
            d = self.new(self.M, 3 * HID)
            qkv.g = d
            self.emit(self.be.attn_bwd(q, k, v, o.g, d[:, :HID], d[:, HID:2 * HID], d[:, 2 * HID:], B, N, float(att.scale)))
        self._tape.append(bw)
        return self.linear(o, att.to_out.weight, att.to_out.bias, residual=a)

    def encoder(self, seq, c0, k, acc):
        xf = self.x_t.view(self.M, -1)
        h = self.act(self.smallk(xf[:, c0:c0 + k], seq[0].weight, seq[0].bias), ACT_GELU)
        h = self.linear(h, seq[2].weight, seq[2].bias, act=ACT_GELU)
        return self.linear(h, seq[4].weight, seq[4].bias, residual=acc)

    # ------------------------------------------------------------------------------------------------ build
    def _build(self):
        net, be, B, N, M = self.net, self.be, self.B, self.N, self.M
        diff = self.diff
        tb = self._tables if self._tables is not None else diff.tables(self.device)
        self.tb = tb
        fl = self.flat
        # ---- q_sample + training target (:531-546)
        want_v = diff.model_mean_type == "v"
        self.v_buf = self.new3(B, N, net.channels) if want_v else None
        self.emit(be.q_sample(self.x0, self.noise, self.t, tb["sqrt_alphas_cumprod"], tb["sqrt_one_minus_alphas_cumprod"],
                              self.x_t, self.v_buf))
        self.target = {"v": self.v_buf, "eps": self.noise, "x0": self.x0}[diff.model_mean_type]
        # ---- weight standardisation of all WS-convs: one launch forward, one (or one per block) backward
        blocks = net.resblocks_in_order()
        ws_mods = []
        for rb, _ in blocks:
            ws_mods += [rb.block1.proj, rb.block2.proj]
        self.ws_mods = ws_mods
        self.ws_std = {id(m): self.new(*_as2d(m.weight).shape) for m in ws_mods}
        self.dws = {id(m): self.new(*_as2d(m.weight).shape) for m in ws_mods}
        self.ws_t = {id(m): self.new(_as2d(m.weight).shape[1], _as2d(m.weight).shape[0]) for m in ws_mods}
        self._ws_pending = []
        self.emit(be.ws([_as2d(m.weight) for m in ws_mods], [self.ws_std[id(m)] for m in ws_mods]))
        self._fwd_split_at = len(self.fwd)            # the forward list's weight-plane splits go here (standardised weights final)
        for m in ws_mods:
            self._transposes.append((self.ws_std[id(m)], self.ws_t[id(m)]))
            if hasattr(be, "transposed_of"):
                be.transposed_of[self.ws_t[id(m)].data_ptr()] = self.ws_std[id(m)]
        # ---- conditioning
        t_blocks = [rb for rb, kind in blocks if kind == "t"]
        c_blocks = [rb for rb, kind in blocks if kind == "c" and rb.mlp is not None]
        t_index = {id(rb): i for i, rb in enumerate(t_blocks)}
        c_index = {id(rb): i for i, rb in enumerate(c_blocks)}
        eng_table = net.time_table.to(self.device)
        eng_freq = net.time_freq.to(self.device)
        temb = H(self.new(B, D), needs_grad=False)
        self.emit(be.time_embedding(self.t, eng_table, eng_freq, temb.t))
        t1 = self.linear(temb, net.time_mlp[1].weight, net.time_mlp[1].bias, act=ACT_GELU)
        t2 = self.linear(t1, net.time_mlp[3].weight, net.time_mlp[3].bias, act=ACT_SILU)
        tw, tgw = fl.packed["t_w"]
        tbias, tgb = fl.packed["t_b"]
        ss_t = H(self.new(B, tw.shape[0]))
        self.emit(be.gemm(t2.t, tw, ss_t.t, tbias))
        dss_t = self.new(B, tw.shape[0])
        ss_t.g = dss_t
        tw_t = self.wT(tw)

        def bw_time_pack():
            # every time-conditioned block has written its [B, 1024] slice of dss_t by now
            self.g_gemm(t2, None, dss_t, tw_t)
            if self.per_block_grads:
                return                       # the per-block TN launches were emitted next to each block (bucket overlap)
            self.tn(t2.t, dss_t, tgw, dbias=tgb, after=lambda: (self.wrote_range(*fl.pack_range["t_w"]),
                                                                 self.wrote_range(*fl.pack_range["t_b"])))
        self._tape.append(bw_time_pack)
        self._t_pack = (t2, dss_t, tgw, tgb)

        ss_c = dss_c = None
        if self.ctx_in is not None and c_blocks:
            cact = self.act(self.ctx_in, ACT_SILU)
            cw, cgw = fl.packed["c_w"]
            cb, cgb = fl.packed["c_b"]
            ss_c = H(self.new(self.ctx_in.t.shape[0], cw.shape[0]))
            self.emit(be.gemm(cact.t, cw, ss_c.t, cb))
            dss_c = self.new(self.ctx_in.t.shape[0], cw.shape[0])
            ss_c.g = dss_c
            cw_t = self.wT(cw)

            def bw_ctx_pack():
                self.g_gemm(cact, None, dss_c, cw_t)
                self.tn(cact.t, dss_c, cgw, dbias=cgb, after=lambda: (self.wrote_range(*fl.pack_range["c_w"]),
                                                                      self.wrote_range(*fl.pack_range["c_b"])))
            self._tape.append(bw_ctx_pack)

        def t_ss(rb):
            i = t_index[id(rb)]
            sl = slice(i * 2 * D, (i + 1) * 2 * D)
            return ss_t.t[:, sl], SS_PER_SCENE, dss_t[:, sl]

        def c_ss(rb):
            if ss_c is None:
                return None, SS_NONE, None
            i = c_index[id(rb)]
            sl = slice(i * 2 * D, (i + 1) * 2 * D)
            return ss_c.t[:, sl], self.ctx_mode, dss_c[:, sl]

        # data parallel: weight gradients are flushed (grouped TN launch + weight-standardisation backward) every few blocks so
        # that finished buckets of G can leave while the backward continues; a single GPU flushes once, at the end
        self._post_count = 0

        def post_flush():
            self._post_count += 1
            if self._post_count % 7 == 0:
                self._flush_ws_pending()

        def cblock(rb, a):
            return self.resblock(rb, a, None, c_ss(rb), post=post_flush if self.per_block_grads else None)

        def tb_(rb, a, a2=None):
            post = None
            if self.per_block_grads:
                i = t_index[id(rb)]
                sl = slice(i * 2 * D, (i + 1) * 2 * D)

                def post(i=i, sl=sl):
                    # data parallel: this block's rows of the packed time-MLP gradient join the pending group right after the
                    # block's backward, so that the bucket holding them can be all-reduced while the rest of the backward runs
                    def done(i=i):
                        o, _ = fl.pack_range["t_w"]
                        self.wrote_range(o + i * 2 * D * tw.shape[1], 2 * D * tw.shape[1])
                        ob, _ = fl.pack_range["t_b"]
                        self.wrote_range(ob + i * 2 * D, 2 * D)
                    self.tn(t2.t, dss_t[:, sl], tgw[sl], dbias=tgb[sl], after=done)
                    post_flush()
            return self.resblock(rb, a, a2, t_ss(rb), post=post)

        # ---- input embedding
        xf = self.x_t.view(M, -1)
        if net.seperate_all:
            bb, nc, no, nf = net.bbox_dim, net.class_dim, net.objectness_dim, net.objfeat_dim
            e = self.encoder(net.class_embedf, bb, nc, None)
            e = self.encoder(net.bbox_embedf, 0, bb, e)
            if no > 0:
                e = self.encoder(net.objectness_embedf, bb + nc, no, e)
            if nf > 0:
                e = self.encoder(net.objfeat_embedf, bb + nc + no, nf, e)
            h = self.linear(e, net.init_conv.weight, net.init_conv.bias)
        else:
            h = self.smallk(xf, net.init_conv.weight, net.init_conv.bias)
        r = h
        skips = []
        text = net.text_condition and self.cross_in is not None

        for lvl in net.downs:
            b0, b1, ac, b2, la, down = lvl
            h = cblock(b0, h)
            h = tb_(b1, h)
            skips.append(h)
            if text:
                h = self.crossattn(ac, h)
            h = tb_(b2, h)
            h = self.linattn(la, h)
            skips.append(h)
            if isinstance(down, torch.nn.Conv1d):
                h = self.linear(h, down.weight, down.bias)
        h = cblock(net.mid_block0, h)
        h = tb_(net.mid_block1, h)
        if text:
            h = self.crossattn(net.mid_attn_cross, h)
        h = self.fullattn(net.mid_attn, h)
        h = tb_(net.mid_block2, h)
        for lvl in net.ups:
            b0, b1, ac, b2, la, up = lvl
            h = cblock(b0, h)
            h = tb_(b1, h, skips.pop())
            if text:
                h = self.crossattn(ac, h)
            h = tb_(b2, h, skips.pop())
            h = self.linattn(la, h)
            if isinstance(up, torch.nn.Conv1d):
                h = self.linear(h, up.weight, up.bias)
        h = tb_(net.final_res_block, h, r)
        # ---- output heads, written straight into the (M, C) output at their column offsets
        self.head_outs = []
        if net.seperate_all:
            heads = [(net.bbox_hidden2output, net.bbox_dim), (net.class_hidden2output, net.class_dim)]
            if net.objectness_dim > 0:
                heads.append((net.objectness_hidden2output, net.objectness_dim))
            if net.objfeat_dim > 0:
                heads.append((net.objfeat_hidden2output, net.objfeat_dim))
            col = 0
            for seq, width in heads:
                d1 = self.linear(h, seq[0].weight, seq[0].bias, act=ACT_GELU)
                d2 = self.linear(d1, seq[2].weight, seq[2].bias, act=ACT_GELU)
                o = self.linear(d2, seq[4].weight, seq[4].bias, out=self.out[:, col:col + width])
                o.g = self.dout[:, col:col + width]
                self.head_outs.append(o)
                col += width
        else:
            o = self.linear(h, net.final_conv.weight, net.final_conv.bias, out=self.out)
            o.g = self.dout
            self.head_outs.append(o)
        # ---- loss + d loss / d out (loss = losses.mean() -> grad_scale 1/B)
        from . import ops
        ca, cb_ = diff._coeffs(tb)
        arrange = bool(getattr(diff, "room_arrange_condition", False))
        if arrange:
            dims = dict(translation_dim=diff.translation_dim, size_dim=0, bbox_dim=net.channels, class_dim=0,
                        objectness_dim=0, objfeat_dim=0)
            iou, bounds = False, None
        else:
            dims = dict(translation_dim=diff.translation_dim, size_dim=diff.size_dim, bbox_dim=diff.bbox_dim,
                        class_dim=diff.class_dim, objectness_dim=diff.objectness_dim, objfeat_dim=diff.objfeat_dim)
            iou = bool(diff.loss_iou)
            bounds = (list(diff._centroids[0]) + list(diff._centroids[1]) + list(diff._sizes[0]) + list(diff._sizes[1])) \
                if iou else None
        mean_type = {"eps": ops.MEAN_EPS, "x0": ops.MEAN_X0, "v": ops.MEAN_V}[diff.model_mean_type]
        if hasattr(be, "split_steps"):                # bf16 planes of the forward's weight operands, refreshed once per step
            sp = be.split_steps()
            self.fwd[self._fwd_split_at:self._fwd_split_at] = sp
        self.n_fwd_only = len(self.fwd)
        self.emit(be.loss(self.target, self.out.view(B, N, -1), self.x_t, self.t, tb, ca, cb_, bounds, dims,
                          bool(diff.loss_separate), iou, mean_type, self.losses, self.parts, self.dout.view(B, N, -1),
                          self.grad_scale))
        self._build_backward()

    def _build_backward(self):
        be = self.be
        self._cur = self.bwd
        assert not self.bwd
        self._ws_flushed = 0
        for f in reversed(self._tape):
            f()
        self._flush_ws_pending()                      # single GPU: all 56 weight-standardisation backwards in one launch
        # context gradient of a parameter-backed context (learnable instance embedding): straight into G
        if self.ctx_in is not None and self.ctx_grad_param is not None and self.ctx_in.g is not None:
            self.emit(be.copy(self.gview(self.ctx_grad_param), self.ctx_in.g))
            self.wrote(self.ctx_grad_param)
        # the transposes of this step's weights run first; they were registered while the emitters ran
        body = self.bwd
        self.bwd = []
        self._cur = self.bwd
        tr = self._transposes
        if hasattr(be, "f32_reads"):
            # a transposed f32 copy is launched only when some product reads it as f32 (a launch on the exact-f32 kernel, or a plane
            # split of a slice of it); where every consumer takes bf16 planes, the planes come straight from the source weight with
            # the transpose folded into the split (HipBackend.split_steps) and the copy is never made.  The buffers skipped here are
            # poisoned once, so that a read nobody planned for shows up as NaN gradients instead of stale values
            keep = [(w, o) for w, o in tr if o.untyped_storage().data_ptr() in be.f32_reads]
            for w, o in tr:
                if o.untyped_storage().data_ptr() not in be.f32_reads:
                    o.fill_(float("nan"))
            self.n_transposes_skipped = len(tr) - len(keep)
            tr = keep
        if tr:
            self.emit(be.transpose_many(tr))
        if hasattr(be, "split_steps"):                # planes of the transposed weights (operands of the input-gradient GEMMs)
            self.emit(be.split_steps())
        shift = len(self.bwd)
        self.bwd.extend(body)
        self.bwd_writes = [(i + shift, r) for i, r in self.bwd_writes]
        self.d_ctx = self.ctx_in.g if self.ctx_in is not None else None
        self.d_cross = self.cross_in.g if self.cross_in is not None else None

    def _flush_ws_pending(self):
        self.flush_tn()                               # the d w_std operands of the weight-standardisation backward
        convs = self._ws_pending[self._ws_flushed:]
        if not convs:
            return
        self._ws_flushed = len(self._ws_pending)
        ws = [_as2d(m.weight) for m in convs]
        self.emit(self.be.ws_bwd(ws, [self.dws[id(m)] for m in convs], [self.gview(m.weight) for m in convs]))
        self.wrote(*[m.weight for m in convs])

    # ------------------------------------------------------------------------------------------------ run
    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else None

    def run_forward(self, with_loss=True):
        self.be.run(self.fwd if with_loss else self.fwd[:self.n_fwd_only], self.stream())

    def run_backward(self, on_progress=None):
        """Enqueue the backward launches; ``on_progress(i)`` is called after launch ``i`` has been enqueued (the data-parallel
        reducer uses it to start the all-reduce of finished buckets)."""
        if on_progress is None:
            self.be.run(self.bwd, self.stream())
            return
        s = self.stream()
        for i in range(len(self.bwd)):
            self.be.run(self.bwd[i:i + 1], s)
            on_progress(i)

    def run_backward_range(self, lo, hi):
        """Launches lo .. hi-1 of the backward list (one captured segment of the data-parallel step)."""
        self.be.run(self.bwd[lo:hi], self.stream())

    def bucket_schedule(self, buckets):
        """For contiguous G ranges [(start, end)]: {launch index: [bucket ids finished by that launch]}; buckets nothing
        in the plan writes (wrapper-level parameters) are returned under the key None."""
        last = [None] * len(buckets)
        for idx, (off, n) in self.bwd_writes:
            for b, (s, e) in enumerate(buckets):
                if off < e and off + n > s:
                    last[b] = idx if last[b] is None else max(last[b], idx)
        sched = {}
        for b, idx in enumerate(last):
            sched.setdefault(idx, []).append(b)
        return sched
