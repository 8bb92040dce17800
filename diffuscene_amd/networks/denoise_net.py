"""Unet1D denoiser of DiffuScene on MI355X -- drop-in for scene_synthesis/networks/denoise_net.py.

Same constructor kwargs (reference denoise_net.py:336-362, dead ones accepted and ignored), same
``forward(x, beta, context=None, context_cross=None)`` contract ((B,N,C) fp32, (B,) int64 -> (B,N,C)
contiguous), and the same ``state_dict`` keys and shapes so reference checkpoints load
unchanged.  The modules below only HOLD parameters under the reference's names; all arithmetic runs in
hand-written HIP kernels through ``DenoiserEngine`` (inference: static launch plan) or ``autograd_ops``
(autograd over the same kernels).  There is no PyTorch/CPU fallback: a CPU input raises.
"""
import math

import torch
from torch import nn

_SUPPORTED_DIM = 512


class ChannelNormGain(nn.Module):
    """Holder of the gain ``g`` (1, dim, 1) of the reference LayerNorm (denoise_net.py:93-96)."""

    def __init__(self, dim):
        super().__init__()
        self.g = nn.Parameter(torch.ones(1, dim, 1))


class ConvGN(nn.Module):
    """Parameters of reference ``Block`` (:160-165): weight-standardised 1x1 conv + GroupNorm."""

    def __init__(self, dim, dim_out, groups):
        super().__init__()
        self.proj = nn.Conv1d(dim, dim_out, 1)
        self.norm = nn.GroupNorm(groups, dim_out)


class ResBlockParams(nn.Module):
    """Parameters of reference ``ResnetBlock`` (:178-188)."""

    def __init__(self, dim, dim_out, emb_dim, groups=8):
        super().__init__()
        self.mlp = nn.Sequential(nn.SiLU(), nn.Linear(emb_dim, dim_out * 2)) if emb_dim is not None else None
        self.block1 = ConvGN(dim, dim_out, groups)
        self.block2 = ConvGN(dim_out, dim_out, groups)
        self.has_res_conv = dim != dim_out
        self.res_conv = nn.Conv1d(dim, dim_out, 1) if self.has_res_conv else nn.Identity()


class LinearAttnParams(nn.Module):
    """reference LinearAttention (:208-219)"""

    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.scale, self.heads = dim_head ** -0.5, heads
        hidden = heads * dim_head
        self.to_qkv = nn.Conv1d(dim, hidden * 3, 1, bias=False)
        self.to_out = nn.Sequential(nn.Conv1d(hidden, dim, 1), ChannelNormGain(dim))


class AttnParams(nn.Module):
    """reference Attention (:237-245)"""

    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.scale, self.heads = dim_head ** -0.5, heads
        hidden = heads * dim_head
        self.to_qkv = nn.Conv1d(dim, hidden * 3, 1, bias=False)
        self.to_out = nn.Conv1d(hidden, dim, 1)


class CrossLinearAttnParams(nn.Module):
    """reference LinearAttentionCross (:261-276)"""

    def __init__(self, dim, context_dim, heads=4, dim_head=32):
        super().__init__()
        self.scale, self.heads = dim_head ** -0.5, heads
        hidden = heads * dim_head
        self.to_q = nn.Conv1d(dim, hidden, 1, bias=False)
        self.to_kv = nn.Conv1d(context_dim if context_dim is not None else dim, hidden * 2, 1, bias=False)
        self.to_out = nn.Sequential(nn.Conv1d(hidden, dim, 1), ChannelNormGain(dim))


class PreNormed(nn.Module):
    """``fn`` + ``norm`` pair (reference PreNorm / PreNormCross, :104-123)."""

    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = ChannelNormGain(dim)


class Skip(nn.Module):
    """reference Residual / ResidualCross wrapper (:39-53): only contributes the ``fn.`` key level."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn


def _mlp3(sizes):
    (a, b), (c, d), (e, f) = sizes
    return nn.Sequential(nn.Conv1d(a, b, 1), nn.GELU(), nn.Conv1d(c, d, 1), nn.GELU(), nn.Conv1d(e, f, 1))


class Unet1D(nn.Module):
    def __init__(self, dim=256, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=3,
                 self_condition=False, seperate_all=False, merge_bbox=False, objectness_dim=1, class_dim=21,
                 translation_dim=3, size_dim=3, angle_dim=1, objfeat_dim=0, context_dim=256, instanclass_dim=0,
                 modulate_time_context_instanclass=False, text_condition=False, text_dim=256,
                 resnet_block_groups=8, learned_variance=False, learned_sinusoidal_cond=False,
                 random_fourier_features=False, learned_sinusoidal_dim=16, time_table_rows=1000):
        super().__init__()
        dim_mults = tuple(dim_mults)
        if dim != _SUPPORTED_DIM or any(m != 1 for m in dim_mults) or (init_dim not in (None, dim)) \
                or resnet_block_groups != 8 or learned_variance or learned_sinusoidal_cond or random_fourier_features:
            raise NotImplementedError(
                "diffuscene_amd.Unet1D implements the layout every shipped DiffuScene config uses "
                "(dim=512, dim_mults all 1, 8 GroupNorm groups, sinusoidal time embedding); got dim=%s "
                "dim_mults=%s groups=%s" % (dim, dim_mults, resnet_block_groups))
        self.channels = channels
        self.self_condition = self_condition
        self.seperate_all = seperate_all
        self.objectness_dim, self.class_dim = objectness_dim, class_dim
        self.translation_dim, self.size_dim, self.angle_dim = translation_dim, size_dim, angle_dim
        self.bbox_dim = translation_dim + size_dim + angle_dim
        self.objfeat_dim = objfeat_dim
        self.text_condition, self.text_dim = text_condition, text_dim
        self.dim = dim
        d = dim
        if seperate_all:
            if objectness_dim > 0:
                self.objectness_embedf = _mlp3([(objectness_dim, d), (d, 2 * d), (2 * d, d)])
            if objfeat_dim > 0:
                self.objfeat_embedf = _mlp3([(objfeat_dim, d), (d, 2 * d), (2 * d, d)])
            self.class_embedf = _mlp3([(class_dim, d), (d, 2 * d), (2 * d, d)])
            self.bbox_embedf = _mlp3([(self.bbox_dim, d), (d, 2 * d), (2 * d, d)])
            in_ch = d
            if channels != self.bbox_dim + class_dim + objectness_dim + objfeat_dim:
                raise ValueError("channels (%d) != bbox+class+objectness+objfeat dims" % channels)
            print('separate unet1d encoder of objectness/class/translation/size/angle')
        else:
            in_ch = channels
            if channels > 64:
                raise NotImplementedError("non-separate init_conv supports at most 64 input channels")
            print('unet1d encoder of all object properties')
        self.init_conv = nn.Conv1d(in_ch, d, 1)
        time_dim = d * 4
        self.time_mlp = nn.Sequential(nn.Identity(), nn.Linear(d, time_dim), nn.GELU(), nn.Linear(time_dim, time_dim))
        ctx = context_dim + instanclass_dim
        n_res = len(dim_mults)
        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])
        for i in range(n_res):
            last = i == n_res - 1
            self.downs.append(nn.ModuleList([
                ResBlockParams(d, d, ctx), ResBlockParams(d, d, time_dim),
                Skip(PreNormed(d, CrossLinearAttnParams(d, text_dim))) if text_condition else nn.Identity(),
                ResBlockParams(d, d, time_dim), Skip(PreNormed(d, LinearAttnParams(d))),
                nn.Conv1d(d, d, 1) if last else nn.Identity()]))
        self.mid_block0 = ResBlockParams(d, d, ctx)
        self.mid_block1 = ResBlockParams(d, d, time_dim)
        self.mid_attn_cross = Skip(PreNormed(d, CrossLinearAttnParams(d, text_dim))) if text_condition else nn.Identity()
        self.mid_attn = Skip(PreNormed(d, AttnParams(d)))
        self.mid_block2 = ResBlockParams(d, d, time_dim)
        for i in range(n_res):
            last = i == n_res - 1
            self.ups.append(nn.ModuleList([
                ResBlockParams(d, d, ctx), ResBlockParams(2 * d, d, time_dim),
                Skip(PreNormed(d, CrossLinearAttnParams(d, text_dim))) if text_condition else nn.Identity(),
                ResBlockParams(2 * d, d, time_dim), Skip(PreNormed(d, LinearAttnParams(d))),
                nn.Conv1d(d, d, 1) if last else nn.Identity()]))
        self.out_dim = out_dim if out_dim is not None else channels
        self.final_res_block = ResBlockParams(2 * d, d, time_dim)
        if seperate_all:
            if objectness_dim > 0:
                self.objectness_hidden2output = _mlp3([(d, 2 * d), (2 * d, d), (d, objectness_dim)])
            if objfeat_dim > 0:
                self.objfeat_hidden2output = _mlp3([(d, 2 * d), (2 * d, d), (d, objfeat_dim)])
            self.class_hidden2output = _mlp3([(d, 2 * d), (2 * d, d), (d, class_dim)])
            self.bbox_hidden2output = _mlp3([(d, 2 * d), (2 * d, d), (d, self.bbox_dim)])
            print('separate unet1d decoder of objectness/class/translation/size/angle')
        else:
            self.final_conv = nn.Conv1d(d, self.out_dim, 1)
            print('unet1d decoder of all object properties')
        # SinusoidalPosEmb (reference :132-139) tabulated on the host with the reference's own fp32 expression,
        # so the device gathers bit-identical rows for every DDPM timestep (plain attributes: not in state_dict)
        half = d // 2
        freq = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
        arg = torch.arange(time_table_rows)[:, None] * freq[None, :]
        self.time_freq = freq.contiguous()
        self.time_table = torch.cat((arg.sin(), arg.cos()), dim=-1).contiguous()
        self._engine = None
        self._engine_device = None

    # -- structure helpers used by the engines ---------------------------------------------------
    def resblocks_in_order(self):
        """(ResBlockParams, 'c' | 't') in execution order (reference forward, :542-575)."""
        out = []
        for lvl in self.downs:
            out += [(lvl[0], "c"), (lvl[1], "t"), (lvl[3], "t")]
        out += [(self.mid_block0, "c"), (self.mid_block1, "t"), (self.mid_block2, "t")]
        for lvl in self.ups:
            out += [(lvl[0], "c"), (lvl[1], "t"), (lvl[3], "t")]
        out.append((self.final_res_block, "t"))
        return out

    def engine(self, device):
        from ..engine import DenoiserEngine
        from .._lib import gemm_mode
        if self._engine is None or self._engine_device != device or self._engine.mode != gemm_mode():
            self._engine = DenoiserEngine(self, device)
            self._engine_device = device
        return self._engine

    def _apply(self, fn, *a, **k):
        self._engine = None          # parameters may move: plans hold raw pointers
        return super()._apply(fn, *a, **k)

    def forward(self, x, beta, context=None, context_cross=None):
        if not x.is_cuda:
            raise RuntimeError("diffuscene_amd.Unet1D runs on a HIP device only (input is on %s); there is no CPU "
                               "fallback" % x.device)
        if x.dim() != 3 or x.shape[-1] != self.channels:
            raise AssertionError("expected (B, N, %d) input, got %s" % (self.channels, tuple(x.shape)))
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs_grad:
            from ..autograd_ops import unet1d_train_forward
            return unet1d_train_forward(self, x, beta, context, context_cross)
        return self.engine(x.device).forward(x, beta, context, context_cross)
