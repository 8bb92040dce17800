"""CPU backend of the training plan's op vocabulary (TEST INFRASTRUCTURE -- never imported by the product).

diffuscene_amd/train_plan.py builds the training step as a list of named ops over statically allocated tensors and a
backend lowers them; the product backend (``HipBackend``) lowers to libdiffuscene_hip.so launches.  This backend executes the
same ops with torch on the CPU (fp64 internally, results stored in the plan's fp32 buffers), following the semantics
documented for each entry point in include/diffuscene_hip.h, so that the plan's DATAFLOW -- which buffer every launch reads
and writes, gradient accumulation of multi-consumer activations, aliasing, slices into the flat gradient buffer -- is checked
against torch.autograd over the oracle without a GPU.  Kernel numerics are checked separately on the GPU (tests/test_gpu_*).
"""
import math

import torch
import torch.nn.functional as F

ACT_NONE, ACT_GELU, ACT_SILU = 0, 1, 2
SS_NONE, SS_PER_TOKEN, SS_PER_SCENE, SS_PER_SLOT = 0, 1, 2, 3
EPS = 1e-5


def _act(y, kind):
    if kind == ACT_GELU:
        return F.gelu(y)
    if kind == ACT_SILU:
        return F.silu(y)
    return y


def _gn_forward(z, gamma, beta, ss_rows, n_tok):
    """z [M, 512] fp64 -> SiLU(GroupNorm8(z) * (scale + 1) + shift); ss_rows [M, 1024] or None."""
    M, Cc = z.shape
    B = M // n_tok
    zz = z.view(B, n_tok, 8, Cc // 8)
    mu = zz.mean(dim=(1, 3), keepdim=True)
    var = zz.var(dim=(1, 3), unbiased=False, keepdim=True)
    xh = ((zz - mu) * (var + EPS).rsqrt()).reshape(M, Cc)
    u = xh * gamma + beta
    if ss_rows is not None:
        u = u * (ss_rows[:, :Cc] + 1.0) + ss_rows[:, Cc:]
    return F.silu(u)


def _ss_expand(ss, mode, M, n_tok):
    """scale/shift rows per token for the conditioning modes of dsc_gemm_gn_silu_f32."""
    if ss is None or mode == SS_NONE:
        return None
    if mode == SS_PER_TOKEN:
        return ss
    B = M // n_tok
    if mode == SS_PER_SCENE:
        return ss[:, None, :].expand(B, n_tok, ss.shape[1]).reshape(M, -1)
    if mode == SS_PER_SLOT:
        return ss[None, :, :].expand(B, n_tok, ss.shape[1]).reshape(M, -1)
    raise ValueError(mode)


def _heads(t, rows_per_scene):
    """[B*n, 128] -> (B, 4, 32, n)"""
    B = t.shape[0] // rows_per_scene
    return t.reshape(B, rows_per_scene, 4, 32).permute(0, 2, 3, 1)


def _linattn(q, k, v, scenes, nq, nk, scale):
    qh, kh, vh = _heads(q, nq), _heads(k, nk), _heads(v, nk)
    qh = qh.softmax(dim=-2) * scale
    kh = kh.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", kh, vh)
    out = torch.einsum("bhde,bhdn->bhen", ctx, qh)                 # (B, 4, 32, nq)
    return out.permute(0, 3, 1, 2).reshape(scenes * nq, 128)


def _attn(q, k, v, scenes, n, scale):
    qh, kh, vh = _heads(q, n) * scale, _heads(k, n), _heads(v, n)
    sim = torch.einsum("bhdi,bhdj->bhij", qh, kh)
    att = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhdj->bhid", att, vh)                  # (B, 4, n, 32)
    return out.permute(0, 2, 1, 3).reshape(scenes * n, 128)


class SimBackend:
    name = "sim"

    def __init__(self, device=None):
        self.device = device or torch.device("cpu")
        self.counts = {}

    def _step(self, name, fn):
        self.counts[name] = self.counts.get(name, 0) + 1
        return fn

    def run(self, steps, stream):
        with torch.no_grad():
            for s in steps:
                s()

    # ---------------------------------------------------------------- forward
    def fuse_act_ok(self, a, w, out, a2=None, bias=None, preact=None, actgrad_x=None):
        """The product backend fuses activation epilogues only into launches its split kernel takes; the dataflow of BOTH forms is
        exercised here: fused for products with >= `fuse_rows` activation rows, separate launches below."""
        return a.shape[0] >= getattr(self, "fuse_rows", 0)

    def gemm(self, a, w, out, bias=None, a2=None, residual=None, act_out=ACT_NONE, preact=None, actgrad_x=None):
        def f():
            x = torch.cat([a, a2], dim=1) if a2 is not None else a
            y = x.double() @ w.double().t()
            if bias is not None:
                y = y + bias.double()
            if preact is not None:                       # include/diffuscene_hip.h: u also stored, y = act_out(u)
                preact.copy_(y.float())
            if actgrad_x is not None:                    # y = product * act_out'(actgrad_x)
                with torch.enable_grad():
                    u = actgrad_x.double().clone().requires_grad_(True)
                    (d,) = torch.autograd.grad(_act(u, act_out).sum(), u)
                y = y * d
            else:
                y = _act(y, act_out)
            if residual is not None:
                y = y + residual.double()
            out.copy_(y.float())
        return self._step("gemm_fused_act" if (preact is not None or actgrad_x is not None) else "gemm", f)

    def gemm_gn(self, a, w, out, bias, gamma, beta, n_tok, a2=None, ss=None, ss_mode=SS_NONE, residual=None, preact=None):
        def f():
            x = torch.cat([a, a2], dim=1) if a2 is not None else a
            z = x.double() @ w.double().t() + bias.double()
            if preact is not None:
                preact.copy_(z.float())
            rows = _ss_expand(ss.double() if ss is not None else None, ss_mode, z.shape[0], n_tok)
            y = _gn_forward(z, gamma.double(), beta.double(), rows, n_tok)
            if residual is not None:
                y = y + residual.double()
            out.copy_(y.float())
        return self._step("gemm_gn", f)

    def smallk(self, x, w, bias, out, act_out=ACT_NONE):
        def f():
            y = x.double() @ w.double().t()
            if bias is not None:
                y = y + bias.double()
            out.copy_(_act(y, act_out).float())
        return self._step("smallk", f)

    def ws(self, weights, outs):
        def f():
            for w, o in zip(weights, outs):
                wd = w.double()
                mean = wd.mean(dim=1, keepdim=True)
                var = wd.var(dim=1, unbiased=False, keepdim=True)
                o.copy_(((wd - mean) * (var + EPS).rsqrt()).float())
        return self._step("ws", f)

    def time_embedding(self, t, table, freq, out):
        def f():
            out.copy_(table[t])
        return self._step("time_embedding", f)

    def act(self, x, out, kind):
        def f():
            out.copy_(_act(x.double(), kind).float())
        return self._step("act", f)

    def layernorm(self, x, g, out, residual=None):
        def f():
            xd = x.double()
            mean = xd.mean(dim=1, keepdim=True)
            var = xd.var(dim=1, unbiased=False, keepdim=True)
            y = (xd - mean) * (var + EPS).rsqrt() * g.double()
            if residual is not None:
                y = y + residual.double()
            out.copy_(y.float())
        return self._step("layernorm", f)

    def linattn(self, q, k, v, out, scenes, nq, nk, scale):
        def f():
            out.copy_(_linattn(q.double(), k.double(), v.double(), scenes, nq, nk, scale).float())
        return self._step("linattn", f)

    def attn(self, q, k, v, out, scenes, n, scale):
        def f():
            out.copy_(_attn(q.double(), k.double(), v.double(), scenes, n, scale).float())
        return self._step("attn", f)

    def q_sample(self, x0, noise, t, sqrt_ac, sqrt_1mac, xt, v):
        def f():
            a = sqrt_ac[t].view(-1, 1, 1)
            b = sqrt_1mac[t].view(-1, 1, 1)
            xt.copy_(a * x0 + b * noise)
            if v is not None:
                v.copy_(a * noise - b * x0)
        return self._step("q_sample", f)

    def loss(self, target, out, x_t, t, tb, ca, cb, bounds, dims, separate, iou, mean_type, losses, parts, dout, scale):
        from oracle import ref_torch as R

        def f():
            with torch.enable_grad():
                o = out.double().clone().requires_grad_(True)
                tg, xt = target.double(), x_t.double()
                tr, sz, bb = dims["translation_dim"], dims["size_dim"], dims["bbox_dim"]
                nc, no, nf = dims["class_dim"], dims["objectness_dim"], dims["objfeat_dim"]
                B = o.shape[0]

                def mse(a, b):
                    if b <= a:
                        return torch.zeros(B, dtype=torch.float64)
                    return ((tg[:, :, a:b] - o[:, :, a:b]) ** 2).mean(dim=(1, 2))
                arrange = sz == 0 and nc == 0 and no == 0 and nf == 0
                l_trans, l_size, l_angle, l_bbox = mse(0, tr), mse(tr, tr + sz), mse(tr + sz, bb), mse(0, bb)
                l_class = mse(bb, bb + nc)
                l_obj = torch.zeros(B, dtype=torch.float64) if arrange else \
                    (mse(bb + nc - 1, bb + nc) if no == 0 else mse(bb + nc, bb + nc + no))
                l_feat = mse(bb + nc + no, o.shape[-1]) if nf > 0 else torch.zeros(B, dtype=torch.float64)
                if arrange and separate:
                    ls = l_trans + l_angle
                elif separate:
                    ls = l_bbox + l_class
                    if no > 0:
                        ls = ls + l_obj
                    if nf > 0:
                        ls = ls + l_feat
                else:
                    ls = ((tg - o) ** 2).mean(dim=(1, 2))
                lw = ls * tb["loss_weight"].double()[t]
                liou = torch.zeros(B, dtype=torch.float64)
                iou_avg = torch.zeros(B, dtype=torch.float64)
                if iou:
                    if ca is None:                       # 'x0': the network output IS the reconstruction, no coefficient tables
                        xr = o.clamp(-1.0, 1.0)
                    else:
                        A = ca.double()[t].view(-1, 1, 1)
                        Bc = cb.double()[t].view(-1, 1, 1)
                        xr = (A * xt - Bc * o).clamp(-1.0, 1.0)
                    if no > 0:
                        valid = (xr[:, :, bb + nc:bb + nc + no] >= 0).double().squeeze(2)
                    else:
                        valid = (xr[:, :, bb + nc - 1:bb + nc] <= 0).double().squeeze(2)
                    bd = torch.tensor([float(v) for v in bounds], dtype=torch.float64)
                    ctr = R.descale(xr[:, :, :tr], bd[0:3], bd[3:6])
                    siz = R.descale(xr[:, :, tr:tr + sz], bd[6:9], bd[9:12])
                    corners = torch.cat([ctr - siz, ctr + siz], dim=-1)
                    iou_m = R.bbox_iou_3d(corners, corners)
                    mask = valid[:, :, None] * valid[:, None, :]
                    iv = iou_m * mask
                    den = mask.sum(dim=(1, 2)) + 1e-6
                    iou_avg = iv.sum(dim=(1, 2)) / den
                    w = tb["alphas_cumprod"].double()[t].reshape(B, 1, 1)
                    liou = (w * 0.1 * iv).sum(dim=(1, 2)) / den
                    lw = lw + liou
                g, = torch.autograd.grad(lw.sum() * scale, o)
            losses.copy_(lw.detach().float())
            parts.copy_(torch.stack([l_bbox, l_trans, l_size, l_angle, l_class, l_obj, l_feat, liou, iou_avg],
                                    dim=1).detach().float())
            dout.copy_(g.float())
        return self._step("loss", f)

    # ---------------------------------------------------------------- backward
    def gemm_tn(self, a, dy, out, a2=None, kvalid=None, dbias=None):
        def f():
            x = torch.cat([a, a2], dim=1) if a2 is not None else a
            kv = x.shape[1] if kvalid is None else kvalid
            out.copy_((dy.double().t() @ x.double()[:, :kv]).float())
            if dbias is not None:
                dbias.copy_(dy.double().sum(dim=0).float())
        return self._step("gemm_tn", f)

    def gemm_tn_grouped(self, items):
        steps = [self.gemm_tn(it["a"], it["dy"], it["out"], it.get("a2"), it.get("kvalid"), it.get("dbias")) for it in items]
        self.counts["gemm_tn_grouped"] = self.counts.get("gemm_tn_grouped", 0) + 1

        def f():
            for st in steps:
                st()
        return f

    def colsum(self, x, out):
        def f():
            out.copy_(x.double().sum(dim=0).float())
        return self._step("colsum", f)

    def gn_bwd(self, z, dy, gamma, beta, ss, ss_mode, dz, part, dss, scenes, n_tok):
        def f():
            M, Cc = z.shape
            with torch.enable_grad():
                zd = z.double().clone().requires_grad_(True)
                # per-scene copies of the affine parameters so that their gradients come out per scene (the kernel writes
                # per-scene partials, reduced later by dsc_colsum_f32)
                gm = gamma.double()[None, :].expand(scenes, Cc).clone().requires_grad_(True)
                bt = beta.double()[None, :].expand(scenes, Cc).clone().requires_grad_(True)
                rows = None
                ssd = None
                if ss is not None and ss_mode != SS_NONE:
                    ssd = ss.double().clone().requires_grad_(True)
                    rows = _ss_expand(ssd, ss_mode, M, n_tok)
                    if ss_mode in (SS_PER_TOKEN, SS_PER_SLOT):
                        # the kernel reports PER_TOKEN / PER_SLOT gradients per token
                        rows = rows.detach().clone().requires_grad_(True)
                g_rows = gm[:, None, :].expand(scenes, n_tok, Cc).reshape(M, Cc)
                b_rows = bt[:, None, :].expand(scenes, n_tok, Cc).reshape(M, Cc)
                zz = zd.view(scenes, n_tok, 8, Cc // 8)
                mu = zz.mean(dim=(1, 3), keepdim=True)
                var = zz.var(dim=(1, 3), unbiased=False, keepdim=True)
                xh = ((zz - mu) * (var + EPS).rsqrt()).reshape(M, Cc)
                u = xh * g_rows + b_rows
                if rows is not None:
                    u = u * (rows[:, :Cc] + 1.0) + rows[:, Cc:]
                y = F.silu(u)
                wrt = [zd, gm, bt]
                if rows is not None:
                    wrt.append(rows if ss_mode in (SS_PER_TOKEN, SS_PER_SLOT) else ssd)
                grads = torch.autograd.grad(y, wrt, dy.double())
            dz.copy_(grads[0].float())
            dbias = grads[0].view(scenes, n_tok, Cc).sum(dim=1)          # bias sits before the norm: dbias = sum dz
            part.copy_(torch.cat([dbias, grads[1], grads[2]], dim=1).float())
            if rows is not None and dss is not None:
                dss.copy_(grads[3].float())
        return self._step("gn_bwd", f)

    def ws_bwd(self, weights, dws, outs):
        def f():
            for w, g, o in zip(weights, dws, outs):
                w2 = w.view(w.shape[0], -1)
                with torch.enable_grad():
                    wd = w2.double().clone().requires_grad_(True)
                    mean = wd.mean(dim=1, keepdim=True)
                    var = wd.var(dim=1, unbiased=False, keepdim=True)
                    wn = (wd - mean) * (var + EPS).rsqrt()
                    gw, = torch.autograd.grad(wn, wd, g.double())
                o.view(o.shape[0], -1).copy_(gw.float())
        return self._step("ws_bwd", f)

    def layernorm_bwd(self, x, g, dy, dx, dg_part, addend=None):
        def f():
            with torch.enable_grad():
                xd = x.double().clone().requires_grad_(True)
                gd = g.double().clone().requires_grad_(True)
                mean = xd.mean(dim=1, keepdim=True)
                var = xd.var(dim=1, unbiased=False, keepdim=True)
                y = (xd - mean) * (var + EPS).rsqrt() * gd
                gx, gg = torch.autograd.grad(y, [xd, gd], dy.double())
            dx.copy_((gx if addend is None else gx + addend.double()).float())
            dg_part.zero_()
            dg_part[0].copy_(gg.float())
        return self._step("layernorm_bwd", f)

    def linattn_bwd(self, q, k, v, dout, dq, dk, dv, scenes, nq, nk, scale):
        def f():
            with torch.enable_grad():
                qd, kd, vd = (t.double().clone().requires_grad_(True) for t in (q, k, v))
                o = _linattn(qd, kd, vd, scenes, nq, nk, scale)
                gq, gk, gv = torch.autograd.grad(o, [qd, kd, vd], dout.double())
            dq.copy_(gq.float()); dk.copy_(gk.float()); dv.copy_(gv.float())
        return self._step("linattn_bwd", f)

    def attn_bwd(self, q, k, v, dout, dq, dk, dv, scenes, n, scale):
        def f():
            with torch.enable_grad():
                qd, kd, vd = (t.double().clone().requires_grad_(True) for t in (q, k, v))
                o = _attn(qd, kd, vd, scenes, n, scale)
                gq, gk, gv = torch.autograd.grad(o, [qd, kd, vd], dout.double())
            dq.copy_(gq.float()); dk.copy_(gk.float()); dv.copy_(gv.float())
        return self._step("attn_bwd", f)

    def act_bwd(self, x, dy, dx, kind):
        def f():
            with torch.enable_grad():
                xd = x.double().clone().requires_grad_(True)
                gx, = torch.autograd.grad(_act(xd, kind), xd, dy.double())
            dx.copy_(gx.float())
        return self._step("act_bwd", f)

    def transpose_many(self, pairs):
        def f():
            for w, o in pairs:
                o.copy_(w.t())
        return [self._step("transpose_many", f)]

    def copy(self, dst, src):
        def f():
            dst.copy_(src)
        return self._step("copy", f)

    def add(self, dst, src):
        def f():
            dst.add_(src)
        return self._step("add", f)


assert math  # noqa: keep the import (used by callers that extend the sim)
