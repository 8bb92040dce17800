"""FoldingNet KL auto-encoder of the object shapes (reference scene_synthesis/networks/foldingnet_autoencoder.py:56-440;
SURVEY.md 8f-4) -- the network that produces the 32/64-d ``objfeats`` latent codes the scene denoiser diffuses.

Same classes, constructor arguments, parameter names / shapes (``state_dict`` interchangeable with the reference) and
call signatures: ``KLAutoEncoder(latent_dim, kl_weight)`` with ``encode / decode / forward / get_loss``, ``AutoEncoder``,
``Encoder``, ``Decoder``, ``train_on_batch``, ``validate_on_batch``.  Underneath nothing is ATen: point features are
token-major ``[cloud * N + point][channel]`` device tensors and every layer is a HIP kernel with a hand-written backward
(``csrc/foldingnet.hip`` + the fp32 MFMA GEMM):

* ``knn`` (:59-76): xyz -> distances in-kernel; features -> per-cloud Gram matrices by ONE batched GEMM, then a top-16
  selection kernel (one wave per point).  Neighbour order is nearest-first with ties to the lowest index; everything
  downstream (max pooling, covariance) is order-invariant.
* Conv1d(k=1) + BatchNorm1d(+ReLU): GEMM, then column statistics over all points of the batch (shifted sums, fp64 combine),
  apply; running statistics are updated like ``nn.BatchNorm1d`` (momentum 0.1, unbiased variance) so ``eval()`` works.
* GraphLayer max pooling over 16 neighbours, global max pooling over points: arg-max kept for the backward.
* FoldingLayer first convolution on ``cat([grid | codeword])`` (:247-251): the codeword columns are the same for all 2025 grid
  points of a cloud, so ``Wc . codeword + bias`` is one small GEMM per batch and the per-point part is a rank-2 / rank-3 update
  (``dsc_point_affine_f32``) -- the 514- / 515-wide concatenated input is never built (250x fewer FLOPs in that layer).
* Chamfer loss: ``diffuscene_amd.chamfer.chamfer_3DDist`` (deterministic backward).

``posterior.sample()`` draws ``torch.randn(mean.shape)`` on the CPU generator and moves it to the device, exactly as the reference
does (:322), so seeded runs see the same noise."""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.nn.utils import clip_grad_norm_

from .. import _lib, ops
from ..autograd_ops import LinearFn, linear_any
from ..chamfer import chamfer_3DDist
from ..stats_logger import StatsLogger

K = 16


def _check(rc, what):
    _lib.check(rc, what)


def _ws(rows, ch, device):
    n = _lib.fn("dsc_bn_workspace_floats")(rows, ch)
    return torch.empty((max(int(n), 1),), device=device, dtype=torch.float32)


# ------------------------------------------------------------------------------------------------ kernels as ops
def knn16(x, clouds, n):
    """x [clouds*n, d] token-major -> int32 [clouds*n, 16] neighbour indices inside the cloud (reference knn, :59-76)."""
    ops._dev(x)
    d = x.shape[1]
    idx = torch.empty((clouds * n, K), device=x.device, dtype=torch.int32)
    if d == 3:
        _check(_lib.fn("dsc_knn16_f32")(x.data_ptr(), x.stride(0), 3, None, None, clouds, n, idx.data_ptr(), ops.stream_ptr()),
               "dsc_knn16_f32")
        return idx
    if d % 32:
        raise RuntimeError("feature kNN needs a channel count that is a multiple of 32, got %d" % d)
    gram = torch.empty((clouds * n, n), device=x.device, dtype=torch.float32)
    g = ops.make_gemm_args(x[:n], x[:n], gram[:n])                   # per-cloud X . X^T, batched over the clouds
    g.batch, g.sa1, g.sw, g.sy = clouds, n * x.stride(0), n * x.stride(0), n * n
    ops.run_gemm(g)
    sq = torch.empty((clouds * n,), device=x.device, dtype=torch.float32)
    _check(_lib.fn("dsc_rowsq_f32")(x.data_ptr(), x.stride(0), d, clouds * n, sq.data_ptr(), ops.stream_ptr()), "dsc_rowsq_f32")
    _check(_lib.fn("dsc_knn16_f32")(x.data_ptr(), x.stride(0), d, gram.data_ptr(), sq.data_ptr(), clouds, n, idx.data_ptr(),
                                    ops.stream_ptr()), "dsc_knn16_f32")
    return idx


def knn_cov(xyz, idx, clouds, n):
    out = torch.empty((clouds * n, 12), device=xyz.device, dtype=torch.float32)
    _check(_lib.fn("dsc_knn_cov_f32")(xyz.data_ptr(), xyz.stride(0), idx.data_ptr(), clouds, n, out.data_ptr(), 12,
                                      ops.stream_ptr()), "dsc_knn_cov_f32")
    return out


class GatherMaxFn(Function):
    """local max pooling over the 16 neighbours (GraphLayer, :160-165)"""

    @staticmethod
    def forward(ctx, x, idx, clouds, n):
        ch = x.shape[1]
        out = torch.empty((clouds * n, ch), device=x.device, dtype=torch.float32)
        arg = torch.empty((clouds * n, ch), device=x.device, dtype=torch.uint8)
        _check(_lib.fn("dsc_gather_max_f32")(x.data_ptr(), x.stride(0), idx.data_ptr(), clouds, n, ch, out.data_ptr(), ch,
                                             arg.data_ptr(), ops.stream_ptr()), "dsc_gather_max_f32")
        ctx.save_for_backward(idx, arg)
        ctx.dims = (clouds, n, ch)
        return out

    @staticmethod
    def backward(ctx, dy):
        idx, arg = ctx.saved_tensors
        clouds, n, ch = ctx.dims
        dy = dy.contiguous()
        dx = torch.zeros((clouds * n, ch), device=dy.device, dtype=torch.float32)
        _check(_lib.fn("dsc_gather_max_bwd_f32")(dy.data_ptr(), ch, idx.data_ptr(), arg.data_ptr(), clouds, n, ch,
                                                 dx.data_ptr(), ch, ops.stream_ptr()), "dsc_gather_max_bwd_f32")
        return dx, None, None, None


class BatchNormFn(Function):
    """nn.BatchNorm1d in training mode over the rows of x [rows, ch] (+ ReLU)"""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, relu):
        rows, ch = x.shape
        y = torch.empty((rows, ch), device=x.device, dtype=torch.float32)
        xhat = torch.empty_like(y)
        mean = torch.empty((ch,), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        ws = _ws(rows, ch, x.device)
        _check(_lib.fn("dsc_batchnorm_fwd_f32")(x.data_ptr(), x.stride(0), gamma.data_ptr(), beta.data_ptr(), rows, ch, eps,
                                                momentum, 1 if relu else 0, mean.data_ptr(), rstd.data_ptr(),
                                                running_mean.data_ptr() if running_mean is not None else None,
                                                running_var.data_ptr() if running_var is not None else None,
                                                xhat.data_ptr(), y.data_ptr(), ws.data_ptr(), ws.numel(), ops.stream_ptr()),
               "dsc_batchnorm_fwd_f32")
        ctx.save_for_backward(xhat, y, gamma, rstd)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        xhat, y, gamma, rstd = ctx.saved_tensors
        rows, ch = xhat.shape
        dy = dy.contiguous()
        dx = torch.empty_like(xhat)
        dg = torch.empty((ch,), device=dy.device, dtype=torch.float32)
        db = torch.empty_like(dg)
        ws = _ws(rows, ch, dy.device)
        _check(_lib.fn("dsc_batchnorm_bwd_f32")(dy.data_ptr(), xhat.data_ptr(), y.data_ptr(), gamma.data_ptr(), rstd.data_ptr(),
                                                rows, ch, 1 if ctx.relu else 0, dx.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                                ws.data_ptr(), ws.numel(), ops.stream_ptr()), "dsc_batchnorm_bwd_f32")
        return dx, dg, db, None, None, None, None, None


def batchnorm_eval(x, bn, relu):
    rows, ch = x.shape
    y = torch.empty((rows, ch), device=x.device, dtype=torch.float32)
    rstd = torch.rsqrt(bn.running_var + bn.eps)
    _check(_lib.fn("dsc_batchnorm_eval_f32")(x.data_ptr(), x.stride(0), bn.weight.data_ptr(), bn.bias.data_ptr(),
                                             bn.running_mean.data_ptr(), rstd.data_ptr(), rows, ch, 1 if relu else 0,
                                             y.data_ptr(), ops.stream_ptr()), "dsc_batchnorm_eval_f32")
    return y


class RowMaxFn(Function):
    """global max pooling over the points of every cloud (:219)"""

    @staticmethod
    def forward(ctx, x, clouds, n):
        ch = x.shape[1]
        out = torch.empty((clouds, ch), device=x.device, dtype=torch.float32)
        arg = torch.empty((clouds, ch), device=x.device, dtype=torch.int32)
        _check(_lib.fn("dsc_rowmax_f32")(x.data_ptr(), x.stride(0), clouds, n, ch, out.data_ptr(), arg.data_ptr(),
                                         ops.stream_ptr()), "dsc_rowmax_f32")
        ctx.save_for_backward(arg)
        ctx.dims = (clouds, n, ch)
        return out

    @staticmethod
    def backward(ctx, dy):
        arg, = ctx.saved_tensors
        clouds, n, ch = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty((clouds * n, ch), device=dy.device, dtype=torch.float32)
        _check(_lib.fn("dsc_rowmax_bwd_f32")(dy.data_ptr(), arg.data_ptr(), clouds, n, ch, dx.data_ptr(), ops.stream_ptr()),
               "dsc_rowmax_bwd_f32")
        return dx, None, None


class PointAffineFn(Function):
    """y[b*n + p] = wp . x[row] + t[b]  (FoldingLayer first conv without the concatenation, :247-251)"""

    @staticmethod
    def forward(ctx, x, wp, t, clouds, n, per_cloud):
        ch, d = wp.shape
        y = torch.empty((clouds * n, ch), device=t.device, dtype=torch.float32)
        _check(_lib.fn("dsc_point_affine_f32")(x.data_ptr(), x.stride(0), 1 if per_cloud else 0, wp.data_ptr(), wp.stride(0),
                                               t.data_ptr(), clouds, n, ch, d, y.data_ptr(), ops.stream_ptr()),
               "dsc_point_affine_f32")
        ctx.save_for_backward(x, wp)
        ctx.dims = (clouds, n, ch, d, per_cloud)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wp = ctx.saved_tensors
        clouds, n, ch, d, per_cloud = ctx.dims
        dy = dy.contiguous()
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty((clouds * n, d), device=dy.device, dtype=torch.float32) if need_dx else None
        dwp = torch.empty((ch, d), device=dy.device, dtype=torch.float32)
        dt = torch.empty((clouds, ch), device=dy.device, dtype=torch.float32)
        ws = _ws(clouds * n, ch, dy.device)
        _check(_lib.fn("dsc_point_affine_bwd_f32")(dy.data_ptr(), x.data_ptr(), x.stride(0), 1 if per_cloud else 0, wp.data_ptr(),
                                                   wp.stride(0), clouds, n, ch, d, dx.data_ptr() if need_dx else None, d,
                                                   dwp.data_ptr(), d, dt.data_ptr(), ws.data_ptr(), ws.numel(),
                                                   ops.stream_ptr()), "dsc_point_affine_bwd_f32")
        return dx, dwp, dt, None, None, None


# ------------------------------------------------------------------------------------------------ layers
def _conv(x, conv):
    """nn.Conv1d(k=1) on token-major rows"""
    return linear_any(x, conv.weight, conv.bias)


def _bn(x, bn, relu):
    if bn.training:
        if bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
        return BatchNormFn.apply(x.contiguous(), bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                 bn.momentum if bn.momentum is not None else 0.1, relu)
    return batchnorm_eval(x.contiguous(), bn, relu)


class GraphLayer(nn.Module):
    """reference :139-168; operates on token-major features"""

    def __init__(self, in_channel, out_channel, k=16):
        super().__init__()
        if k != K:
            raise NotImplementedError("the kNN kernels are built for k = 16 (the only value the reference uses)")
        self.k = k
        self.conv = nn.Conv1d(in_channel, out_channel, 1)
        self.bn = nn.BatchNorm1d(out_channel)

    def rows(self, x, clouds, n):
        with torch.no_grad():
            idx = knn16(x.detach().contiguous(), clouds, n)
        x = GatherMaxFn.apply(x.contiguous(), idx, clouds, n)
        return _bn(_conv(x, self.conv), self.bn, True)


class Encoder(nn.Module):
    """Graph based encoder, reference :171-220.  forward(x): x (B, 3, N) as in the reference -> (B, 512)."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv1d(12, 64, 1)
        self.conv2 = nn.Conv1d(64, 64, 1)
        self.conv3 = nn.Conv1d(64, 64, 1)
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(64)
        self.bn3 = nn.BatchNorm1d(64)
        self.graph_layer1 = GraphLayer(in_channel=64, out_channel=128, k=16)
        self.graph_layer2 = GraphLayer(in_channel=128, out_channel=1024, k=16)
        self.conv4 = nn.Conv1d(1024, 512, 1)
        self.bn4 = nn.BatchNorm1d(512)

    def forward(self, x):
        b, c, n = x.shape
        return self.rows(x.permute(0, 2, 1).reshape(b * n, c), b, n)

    def rows(self, xyz, clouds, n):
        """xyz token-major [clouds*n, 3] (no copy when the caller already holds (B, N, 3) points)"""
        xyz = xyz.detach().contiguous()
        with torch.no_grad():
            feats = knn_cov(xyz, knn16(xyz, clouds, n), clouds, n)          # [rows, 12] = xyz | covariances (:197-205)
        x = _bn(_conv(feats, self.conv1), self.bn1, True)
        x = _bn(_conv(x, self.conv2), self.bn2, True)
        x = _bn(_conv(x, self.conv3), self.bn3, True)
        x = self.graph_layer1.rows(x, clouds, n)
        x = self.graph_layer2.rows(x, clouds, n)
        x = _bn(_conv(x, self.conv4), self.bn4, False)
        return RowMaxFn.apply(x, clouds, n)


class FoldingLayer(nn.Module):
    """reference :223-254: shared MLP on cat([grids | codewords]); here the concatenation is never built"""

    def __init__(self, in_channel, out_channels):
        super().__init__()
        layers = []
        for oc in out_channels[:-1]:
            layers.extend([nn.Conv1d(in_channel, oc, 1), nn.BatchNorm1d(oc), nn.ReLU(inplace=True)])
            in_channel = oc
        layers.append(nn.Conv1d(in_channel, out_channels[-1], 1))
        self.layers = nn.Sequential(*layers)

    def rows(self, pts, per_cloud, codewords, clouds, n):
        """pts: [n, d] shared grid (per_cloud False) or [clouds*n, d]; codewords [clouds, 512] -> [clouds*n, out]"""
        conv0 = self.layers[0]
        w = conv0.weight.view(conv0.weight.shape[0], conv0.weight.shape[1])
        d = pts.shape[1]
        t = LinearFn.apply(codewords.contiguous(), w[:, d:].contiguous(), conv0.bias, None, None)      # Wc . code + bias, per cloud
        x = PointAffineFn.apply(pts.contiguous(), w[:, :d].contiguous(), t, clouds, n, per_cloud)
        mods = list(self.layers)[1:]
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.BatchNorm1d):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = _bn(x, m, relu)
                i += 2 if relu else 1
            elif isinstance(m, nn.Conv1d):
                x = _conv(x, m)
                i += 1
            else:
                raise NotImplementedError("FoldingLayer: an activation that does not follow a BatchNorm (not in the reference)")
        return x


class Decoder(nn.Module):
    """reference :257-297: two folding operations on a 45 x 45 grid.  forward(x): (B, C) -> (B, 3, 2025)."""

    def __init__(self, in_channel=512):
        super().__init__()
        xx = np.linspace(-0.3, 0.3, 45, dtype=np.float32)
        yy = np.linspace(-0.3, 0.3, 45, dtype=np.float32)
        grid = np.meshgrid(xx, yy)
        self.grid = torch.Tensor(np.array(grid)).view(2, -1)                 # (2, 2025), a plain attribute as in the reference
        self.m = self.grid.shape[1]
        self.fold1 = FoldingLayer(in_channel + 2, [512, 512, 3])
        self.fold2 = FoldingLayer(in_channel + 3, [512, 512, 3])

    def rows(self, codewords):
        clouds = codewords.shape[0]
        grid = self.grid.to(codewords.device).t().contiguous()               # [2025, 2]
        r1 = self.fold1.rows(grid, False, codewords, clouds, self.m)
        return self.fold2.rows(r1, True, codewords, clouds, self.m)          # [clouds*2025, 3]

    def forward(self, x):
        return self.rows(x).view(x.shape[0], self.m, 3).permute(0, 2, 1)


class AutoEncoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = Encoder()
        self.decoder = Decoder()

    def forward(self, x):
        return self.decoder(self.encoder(x))


class DiagonalGaussianDistribution(object):
    """reference :312-349 (elementwise on (B, latent) tensors)"""

    def __init__(self, mean, logvar, deterministic=False):
        self.mean = mean
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self):
        return self.mean + self.std * torch.randn(self.mean.shape).to(device=self.mean.device)

    def kl(self, other=None):
        if self.deterministic:
            return torch.Tensor([0.])
        if other is None:
            return 0.5 * torch.mean(torch.pow(self.mean, 2) + self.var - 1.0 - self.logvar, dim=[1, ])
        return 0.5 * torch.mean(torch.pow(self.mean - other.mean, 2) / other.var + self.var / other.var - 1.0 - self.logvar
                                + other.logvar, dim=[1, 2, 3])

    def nll(self, sample, dims=[1, 2, 3]):
        if self.deterministic:
            return torch.Tensor([0.])
        logtwopi = np.log(2.0 * np.pi)
        return 0.5 * torch.sum(logtwopi + self.logvar + torch.pow(sample - self.mean, 2) / self.var, dim=dims)

    def mode(self):
        return self.mean


class KLAutoEncoder(nn.Module):
    """reference :351-400"""

    def __init__(self, latent_dim=64, kl_weight=0.001):
        super().__init__()
        self.latent_dim = latent_dim
        self.kl_weight = kl_weight
        self.encoder = Encoder()
        self.mean_fc = nn.Linear(512, latent_dim)
        self.logvar_fc = nn.Linear(512, latent_dim)
        self.fc = nn.Linear(latent_dim, 512)
        self.decoder = Decoder()
        self.cd_loss = chamfer_3DDist()

    def _posterior(self, code):
        mean = linear_any(code, self.mean_fc.weight, self.mean_fc.bias)
        logvar = linear_any(code, self.logvar_fc.weight, self.logvar_fc.bias)
        return DiagonalGaussianDistribution(mean, logvar)

    def encode(self, x):
        posterior = self._posterior(self.encoder(x))
        x = posterior.sample()
        return posterior.kl(), x

    def decode(self, lat):
        x = linear_any(lat, self.fc.weight, self.fc.bias)
        return self.decoder(x).permute(0, 2, 1)

    def forward(self, pc):
        b, n, _ = pc.shape
        posterior = self._posterior(self.encoder.rows(pc.reshape(b * n, 3), b, n))      # (B, N, 3) is already token-major
        lat = posterior.sample()
        x = linear_any(lat, self.fc.weight, self.fc.bias)
        return posterior.kl(), lat, self.decoder.rows(x).view(b, self.decoder.m, 3)

    def get_loss(self, samples):
        pc = samples["points"]
        kl, lat, pc_recon = self.forward(pc)
        loss_kl = torch.sum(kl) / kl.shape[0]
        dist1, dist2, idx1, idx2 = self.cd_loss(pc.contiguous(), pc_recon.contiguous())
        loss_cd = (dist1.mean(dim=1) + dist2.mean(dim=1)).mean()
        loss = loss_cd + loss_kl * self.kl_weight
        return loss, {'loss.cd': loss_cd.mean(), 'loss.kl': loss_kl.mean()}


def train_on_batch(model, optimizer, sample_params, config):
    """reference :404-421"""
    optimizer.zero_grad()
    loss, loss_dict = model.get_loss(sample_params)
    loss.backward()
    grad_norm = clip_grad_norm_(model.parameters(), config["training"]["max_grad_norm"])
    keys = list(loss_dict.keys())
    packed = torch.stack([loss.detach(), grad_norm.detach()] + [loss_dict[k].detach() for k in keys]).tolist()
    for k, v in zip(keys, packed[2:]):
        StatsLogger.instance()[k].value = v
    StatsLogger.instance()["gradnorm"].value = packed[1]
    StatsLogger.instance()["lr"].value = optimizer.param_groups[0]['lr']
    optimizer.step()
    return packed[0]


@torch.no_grad()
def validate_on_batch(model, sample_params, config):
    loss, loss_dict = model.get_loss(sample_params)
    keys = list(loss_dict.keys())
    packed = torch.stack([loss.detach()] + [loss_dict[k].detach() for k in keys]).tolist()
    for k, v in zip(keys, packed[1:]):
        StatsLogger.instance()[k].value = v
    return packed[0]
