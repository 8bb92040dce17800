"""DDPM process of DiffuScene on MI355X -- drop-in for scene_synthesis/networks/diffusion_ddpm.py.

Same names, signatures, assertion behaviour and RNG draw order as the reference (``get_betas`` :45-91,
``GaussianDiffusion`` :125-717, ``DiffusionPoint`` :721-804).  Differences are internal:

* the 13 schedule tables are built exactly as the reference builds them (float64 numpy -> fp32 torch) but are
  uploaded ONCE per device instead of on every ``_extract`` call (:220,:232,:284,...);
* ``q_sample`` / v-target and the whole ``p_mean_variance`` + ``p_sample`` chain (:242-352) are single HIP
  kernels (csrc/diffusion.hip), bit-identical to the reference's fp32 expressions;
* the reverse loops run as a replayed hipGraph (default since round 6; ``graph=False`` or env DSC_GRAPH=0 for the eager loop): one captured step
  with a device-resident timestep, replayed T times -- no Python, no launches on the critical path.
"""
import json
import os
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .loss import axis_aligned_bbox_overlaps_3d  # noqa: F401  (re-exported: the reference module exposes it, :16)

ModelPrediction = namedtuple('ModelPrediction', ['pred_noise', 'pred_x_start'])

_MEAN = {"eps": ops.MEAN_EPS, "x0": ops.MEAN_X0, "v": ops.MEAN_V}


def getGradNorm(net):
    """(parameter norm, gradient norm), reference :28-31 -- two fused multi-tensor reductions instead of 2x|params| kernels."""
    ps = [p for p in net.parameters()]
    pNorm = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(ps)))
    gradNorm = torch.linalg.vector_norm(torch.stack(torch._foreach_norm([p.grad for p in ps])))
    return pNorm, gradNorm


def normal_kl(mean1, logvar1, mean2, logvar2):
    """KL( N(mean1, e^logvar1) || N(mean2, e^logvar2) ) per element -- what the variational-bound diagnostics (_vb_terms_bpd,
    _prior_bpd) need of the reference's helper block (:94-99); its other helpers (identity, norm, weights_init, the discretized
    log-likelihood) have no caller on this path and are not carried."""
    dlog = logvar2 - logvar1
    return 0.5 * (dlog - 1.0 + torch.exp(-dlog) + (mean1 - mean2).pow(2) * torch.exp(-logvar2))


def get_betas(schedule_type, b_start, b_end, time_num):
    if schedule_type == 'linear':
        betas = np.linspace(b_start, b_end, time_num)
    elif schedule_type in ('warm0.1', 'warm0.2', 'warm0.5'):
        frac = float(schedule_type[4:])
        betas = b_end * np.ones(time_num, dtype=np.float64)
        warm = int(time_num * frac)
        betas[:warm] = np.linspace(b_start, b_end, warm, dtype=np.float64)
    else:
        # the reference's 'cosine' branch never assigns betas (diffusion_ddpm.py:84-87, UnboundLocalError)
        raise NotImplementedError(schedule_type)
    return betas


class GaussianDiffusion:
    def __init__(self, config, betas, loss_type, model_mean_type, model_var_type, loss_separate, loss_iou,
                 train_stats_file):
        self.objectness_dim = config.get("objectness_dim", 1)
        self.class_dim = config.get("class_dim", 21)
        self.translation_dim = config.get("translation_dim", 3)
        self.size_dim = config.get("size_dim", 3)
        self.angle_dim = config.get("angle_dim", 1)
        self.bbox_dim = self.translation_dim + self.size_dim + self.angle_dim
        self.objfeat_dim = config.get("objfeat_dim", 0)
        self.loss_separate = loss_separate
        self.loss_iou = loss_iou
        if self.loss_iou:
            with open(train_stats_file, "r") as f:
                train_stats = json.load(f)
            c = train_stats["bounds_translations"]
            self._centroids = (np.array(c[:3]), np.array(c[3:]))
            self._centroids_min = torch.from_numpy(self._centroids[0]).float()
            self._centroids_max = torch.from_numpy(self._centroids[1]).float()
            print('load centriods min {} and max {} in Gausssion Diffusion'.format(*self._centroids))
            s = train_stats["bounds_sizes"]
            self._sizes = (np.array(s[:3]), np.array(s[3:]))
            self._sizes_min = torch.from_numpy(self._sizes[0]).float()
            self._sizes_max = torch.from_numpy(self._sizes[1]).float()
            print('load sizes min {} and max {} in Gausssion Diffusion'.format(*self._sizes))
            a = train_stats["bounds_angles"]
            self._angles = (np.array(a[0]), np.array(a[1]))
        self.room_partial_condition = config.get("room_partial_condition", False)
        self.room_arrange_condition = config.get("room_arrange_condition", False)
        self.loss_type = loss_type
        self.model_mean_type = model_mean_type
        self.model_var_type = model_var_type
        assert isinstance(betas, np.ndarray)
        self.np_betas = betas = betas.astype(np.float64)
        assert (betas > 0).all() and (betas <= 1).all()
        timesteps, = betas.shape
        self.num_timesteps = int(timesteps)

        # float64 numpy -> fp32 torch, op for op as the reference (:168-203) so the tables are bit-identical
        alphas = 1. - betas
        alphas_cumprod = torch.from_numpy(np.cumprod(alphas, axis=0)).float()
        alphas_cumprod_prev = torch.from_numpy(np.append(1., alphas_cumprod[:-1])).float()
        self.betas = torch.from_numpy(betas).float()
        self.alphas_cumprod = alphas_cumprod.float()
        self.alphas_cumprod_prev = alphas_cumprod_prev.float()
        self.sqrt_alphas_cumprod = torch.sqrt(alphas_cumprod).float()
        self.sqrt_one_minus_alphas_cumprod = torch.sqrt(1. - alphas_cumprod).float()
        self.log_one_minus_alphas_cumprod = torch.log(1. - alphas_cumprod).float()
        self.sqrt_recip_alphas_cumprod = torch.sqrt(1. / alphas_cumprod).float()
        self.sqrt_recipm1_alphas_cumprod = torch.sqrt(1. / alphas_cumprod - 1).float()
        betas_t = torch.from_numpy(betas).float()
        alphas_t = torch.from_numpy(alphas).float()
        posterior_variance = betas_t * (1. - alphas_cumprod_prev) / (1. - alphas_cumprod)
        self.posterior_variance = posterior_variance
        self.posterior_log_variance_clipped = torch.log(
            torch.max(posterior_variance, 1e-20 * torch.ones_like(posterior_variance)))
        self.posterior_mean_coef1 = betas_t * torch.sqrt(alphas_cumprod_prev) / (1. - alphas_cumprod)
        self.posterior_mean_coef2 = (1. - alphas_cumprod_prev) * torch.sqrt(alphas_t) / (1. - alphas_cumprod)
        snr = alphas_cumprod / (1 - alphas_cumprod)
        if model_mean_type == 'eps':
            loss_weight = torch.ones_like(snr)
        elif model_mean_type == 'x0':
            loss_weight = snr
        elif model_mean_type == 'v':
            loss_weight = snr / (snr + 1)
        self.loss_weight = loss_weight
        # sigma_t = exp(0.5 * log-variance) of p_sample (:350), tabulated with the same fp32 torch ops
        self._sigma_small = torch.exp(0.5 * self.posterior_log_variance_clipped)
        self._logvar_large = torch.log(torch.cat([self.posterior_variance[1:2], self.betas[1:]]))
        self._sigma_large = torch.exp(0.5 * self._logvar_large)
        self._dev = {}
        self._graphs = {}

    # ------------------------------------------------------------------ device-resident tables
    _TABLE_NAMES = ("betas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                    "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
                    "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2", "loss_weight",
                    "_sigma_small", "_sigma_large", "_logvar_large", "log_one_minus_alphas_cumprod")

    def tables(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("diffuscene_amd.GaussianDiffusion runs on a HIP device only (got %s); there is no CPU "
                               "fallback" % device)
        tb = self._dev.get(device)
        if tb is None:
            flat = torch.stack([getattr(self, n) for n in self._TABLE_NAMES]).to(device)
            tb = {n: flat[i] for i, n in enumerate(self._TABLE_NAMES)}
            self._dev[device] = tb
        return tb

    @staticmethod
    def _extract(a, t, x_shape):
        bs, = t.shape
        assert x_shape[0] == bs
        out = torch.gather(a, 0, t)
        assert out.shape == torch.Size([bs])
        return torch.reshape(out, [bs] + ((len(x_shape) - 1) * [1]))

    # -- parameterisation conversions (not on the fused fast path; device torch ops on resident tables) --
    def _predict_xstart_from_eps(self, x_t, t, eps):
        assert x_t.shape == eps.shape
        tb = self.tables(x_t.device)
        return (self._extract(tb["sqrt_recip_alphas_cumprod"], t, x_t.shape) * x_t -
                self._extract(tb["sqrt_recipm1_alphas_cumprod"], t, x_t.shape) * eps)

    def _predict_eps_from_start(self, x_t, t, x0):
        tb = self.tables(x_t.device)
        return ((self._extract(tb["sqrt_recip_alphas_cumprod"], t, x_t.shape) * x_t - x0) /
                self._extract(tb["sqrt_recipm1_alphas_cumprod"], t, x_t.shape))

    def _predict_v(self, x0, t, eps):
        tb = self.tables(x0.device)
        return (self._extract(tb["sqrt_alphas_cumprod"], t, x0.shape) * eps -
                self._extract(tb["sqrt_one_minus_alphas_cumprod"], t, x0.shape) * x0)

    def _predict_start_from_v(self, x_t, t, v):
        tb = self.tables(x_t.device)
        return (self._extract(tb["sqrt_alphas_cumprod"], t, x_t.shape) * x_t -
                self._extract(tb["sqrt_one_minus_alphas_cumprod"], t, x_t.shape) * v)

    def _coeffs(self, tb):
        """(ca, cb) of dsc_p_sample_f32 for the configured mean type."""
        if self.model_mean_type == 'v':
            return tb["sqrt_alphas_cumprod"], tb["sqrt_one_minus_alphas_cumprod"]
        if self.model_mean_type == 'eps':
            return tb["sqrt_recip_alphas_cumprod"], tb["sqrt_recipm1_alphas_cumprod"]
        return None, None

    def _sigma(self, tb):
        if self.model_var_type == 'fixedsmall':
            return tb["_sigma_small"]
        if self.model_var_type == 'fixedlarge':
            return tb["_sigma_large"]
        raise NotImplementedError(self.model_var_type)

    def model_predictions(self, denoise_fn, x_t, t, condition, condition_cross, x_self_cond=None,
                          clip_x_start=False, rederive_pred_noise=False):
        model_output = denoise_fn(x_t, t, condition, condition_cross)
        clip = (lambda z: torch.clamp(z, min=-1., max=1.)) if clip_x_start else (lambda z: z)
        if self.model_mean_type == 'eps':
            pred_noise = model_output
            x_start = clip(self._predict_xstart_from_eps(x_t, t, pred_noise))
            if clip_x_start and rederive_pred_noise:
                pred_noise = self._predict_eps_from_start(x_t, t, x_start)
        elif self.model_mean_type == 'x0':
            x_start = clip(model_output)
            pred_noise = self._predict_eps_from_start(x_t, t, x_start)
        elif self.model_mean_type == 'v':
            x_start = clip(self._predict_start_from_v(x_t, t, model_output))
            pred_noise = self._predict_eps_from_start(x_t, t, x_start)
        return ModelPrediction(pred_noise, x_start)

    def q_mean_variance(self, x_start, t):
        tb = self.tables(x_start.device)
        mean = self._extract(tb["sqrt_alphas_cumprod"], t, x_start.shape) * x_start
        variance = self._extract(1. - tb["alphas_cumprod"], t, x_start.shape)
        log_variance = self._extract(tb["log_one_minus_alphas_cumprod"], t, x_start.shape)
        return mean, variance, log_variance

    def q_sample(self, x_start, t, noise=None):
        """q(x_t | x_0), reference :276-286 -- one HIP kernel."""
        if noise is None:
            noise = torch.randn(x_start.shape, device=x_start.device)
        assert noise.shape == x_start.shape
        tb = self.tables(x_start.device)
        return ops.q_sample(x_start.contiguous(), noise.contiguous(), t, tb["sqrt_alphas_cumprod"],
                            tb["sqrt_one_minus_alphas_cumprod"])

    def q_posterior_mean_variance(self, x_start, x_t, t):
        assert x_start.shape == x_t.shape
        tb = self.tables(x_start.device)
        posterior_mean = (self._extract(tb["posterior_mean_coef1"], t, x_t.shape) * x_start +
                          self._extract(tb["posterior_mean_coef2"], t, x_t.shape) * x_t)
        posterior_variance = self._extract(tb["posterior_variance"], t, x_t.shape)
        posterior_log_variance_clipped = self._extract(tb["posterior_log_variance_clipped"], t, x_t.shape)
        assert (posterior_mean.shape[0] == posterior_variance.shape[0] == posterior_log_variance_clipped.shape[0] ==
                x_start.shape[0])
        return posterior_mean, posterior_variance, posterior_log_variance_clipped

    def p_mean_variance(self, denoise_fn, data, t, condition, condition_cross, clip_denoised: bool,
                        return_pred_xstart: bool):
        """reference :305-335.  Mean and x_recon come from the fused kernel (zero noise)."""
        if self.model_var_type not in ('fixedsmall', 'fixedlarge'):
            raise NotImplementedError(self.model_var_type)
        model_output = denoise_fn(data, t, condition, condition_cross)
        tb = self.tables(data.device)
        ca, cb = self._coeffs(tb)
        if torch.is_grad_enabled() and model_output.requires_grad:
            # loss_type 'kl' training (:657-660): the KL back-propagates through model_mean to the denoiser, so this
            # branch stays on differentiable device ops over the same tables (the fused kernel has no autograd)
            if self.model_mean_type == 'eps':
                x_recon = self._predict_xstart_from_eps(data, t, eps=model_output)
            elif self.model_mean_type == 'x0':
                x_recon = model_output
            else:
                x_recon = self._predict_start_from_v(data, t, v=model_output)
            if clip_denoised:
                x_recon = torch.clamp(x_recon, -1.0, 1.0)
            model_mean, _, _ = self.q_posterior_mean_variance(x_start=x_recon, x_t=data, t=t)
        else:
            x_recon = torch.empty_like(data)
            model_mean = ops.p_sample(data.contiguous(), model_output.contiguous(), torch.zeros_like(data), t, ca, cb,
                                      tb["posterior_mean_coef1"], tb["posterior_mean_coef2"], self._sigma(tb),
                                      _MEAN[self.model_mean_type], clip_denoised, x0_out=x_recon)
        var_tab = tb["posterior_variance"] if self.model_var_type == 'fixedsmall' else tb["betas"]
        logvar_tab = (tb["posterior_log_variance_clipped"] if self.model_var_type == 'fixedsmall'
                      else tb["_logvar_large"])
        model_variance = self._extract(var_tab, t, data.shape) * torch.ones_like(data)
        model_log_variance = self._extract(logvar_tab, t, data.shape) * torch.ones_like(data)
        assert model_mean.shape == x_recon.shape == data.shape
        assert model_variance.shape == model_log_variance.shape == data.shape
        if return_pred_xstart:
            return model_mean, model_variance, model_log_variance, x_recon
        return model_mean, model_variance, model_log_variance

    # ------------------------------------------------------------------ sampling
    def p_sample(self, denoise_fn, data, t, condition, condition_cross, noise_fn, clip_denoised=False,
                 return_pred_xstart=False):
        """One reverse step, reference :339-352: model call, then ONE noise draw (also at t == 0), then the fused
        x0-from-output / clamp / posterior-mean / masked noise-add kernel."""
        model_output = denoise_fn(data, t, condition, condition_cross)
        noise = noise_fn(size=data.shape, dtype=data.dtype, device=data.device)
        assert noise.shape == data.shape
        tb = self.tables(data.device)
        ca, cb = self._coeffs(tb)
        pred_xstart = torch.empty_like(data) if return_pred_xstart else None
        sample = ops.p_sample(data.contiguous(), model_output.contiguous(), noise.contiguous(), t, ca, cb,
                              tb["posterior_mean_coef1"], tb["posterior_mean_coef2"], self._sigma(tb),
                              _MEAN[self.model_mean_type], clip_denoised, x0_out=pred_xstart)
        assert sample.shape == data.shape
        return (sample, pred_xstart) if return_pred_xstart else sample

    def _total_steps(self, keep_running):
        return self.num_timesteps if not keep_running else len(self.betas)

    def p_sample_loop(self, denoise_fn, shape, device, condition, condition_cross, noise_fn=torch.randn,
                      clip_denoised=True, keep_running=False, graph=None):
        """Generate samples, reference :355-371 (draw order: x_T, then one draw per step)."""
        assert isinstance(shape, (tuple, list))
        if _use_graph(graph, noise_fn, denoise_fn):
            from ..sampler import graph_sample_loop
            return graph_sample_loop(self, denoise_fn, tuple(shape), device, condition, condition_cross,
                                     clip_denoised, self._total_steps(keep_running), noise_fn)
        img_t = noise_fn(size=shape, dtype=torch.float, device=device)
        for t in reversed(range(0, self._total_steps(keep_running))):
            t_ = torch.empty(shape[0], dtype=torch.int64, device=device).fill_(t)
            img_t = self.p_sample(denoise_fn=denoise_fn, data=img_t, t=t_, condition=condition,
                                  condition_cross=condition_cross, noise_fn=noise_fn,
                                  clip_denoised=clip_denoised, return_pred_xstart=False)
        assert img_t.shape == shape
        return img_t

    def p_sample_loop_trajectory(self, denoise_fn, shape, device, freq, condition, condition_cross,
                                 noise_fn=torch.randn, clip_denoised=True, keep_running=False):
        """reference :373-398"""
        assert isinstance(shape, (tuple, list))
        total_steps = self._total_steps(keep_running)
        img_t = noise_fn(size=shape, dtype=torch.float, device=device)
        imgs = [img_t]
        for t in reversed(range(0, total_steps)):
            t_ = torch.empty(shape[0], dtype=torch.int64, device=device).fill_(t)
            img_t = self.p_sample(denoise_fn=denoise_fn, data=img_t, t=t_, condition=condition,
                                  condition_cross=condition_cross, noise_fn=noise_fn,
                                  clip_denoised=clip_denoised, return_pred_xstart=False)
            if t % freq == 0 or t == total_steps - 1:
                imgs.append(img_t)
        assert imgs[-1].shape == shape
        return imgs

    def ddim_sample_loop(self, *args, **kwargs):
        # the reference implementation (:401-444) reads an undefined self.self_condition and calls
        # model_predictions without denoise_fn: it cannot run; no shipped script reaches it.
        raise NotImplementedError("ddim_sample_loop is dead code in the reference (diffusion_ddpm.py:419-420)")

    def p_sample_loop_complete(self, denoise_fn, shape, device, condition, condition_cross, noise_fn=torch.randn,
                               clip_denoised=True, keep_running=False, partial_boxes=None, graph=None):
        """Scene completion, reference :447-476: every step re-noises the given objects (noise drawn BEFORE the
        model call) and overwrites the first P rows of x_t in place; at t == 0 the clean objects are restored."""
        assert isinstance(shape, (tuple, list))
        if _use_graph(graph, noise_fn, denoise_fn):
            from ..sampler import graph_sample_loop
            print('last:', 0, self.num_timesteps, len(self.betas))
            return graph_sample_loop(self, denoise_fn, tuple(shape), device, condition, condition_cross, clip_denoised,
                                     self._total_steps(keep_running), noise_fn, partial_boxes=partial_boxes.contiguous())
        tb = self.tables(device)
        img_t = noise_fn(size=shape, dtype=torch.float, device=device).clone()   # overwritten in place below
        partial_boxes = partial_boxes.contiguous()
        num_partial = partial_boxes.shape[1]
        for t in reversed(range(0, self._total_steps(keep_running))):
            t_ = torch.empty(shape[0], dtype=torch.int64, device=device).fill_(t)
            noise = noise_fn(size=partial_boxes.shape, dtype=torch.float, device=device)
            ops.complete_overwrite(img_t, partial_boxes, noise.contiguous(), t_, tb["sqrt_alphas_cumprod"],
                                   tb["sqrt_one_minus_alphas_cumprod"])
            img_t = self.p_sample(denoise_fn=denoise_fn, data=img_t, t=t_, condition=condition,
                                  condition_cross=condition_cross, noise_fn=noise_fn,
                                  clip_denoised=clip_denoised, return_pred_xstart=False)
            if t == 0:
                print('last:', t, self.num_timesteps, len(self.betas))
                img_t[:, :num_partial, :] = partial_boxes
        assert img_t.shape == shape
        return img_t

    def p_sample_loop_arrange(self, denoise_fn, shape, device, condition, condition_cross, noise_fn=torch.randn,
                              clip_denoised=True, keep_running=False, input_boxes=None, graph=None):
        """Re-arrangement, reference :478-506: diffuse [translation | angle] only, re-assemble at t == 0."""
        assert isinstance(shape, (tuple, list))
        if _use_graph(graph, noise_fn, denoise_fn):
            from ..sampler import graph_sample_loop
            sub = (shape[0], shape[1], self.translation_dim + self.angle_dim)
            img_t = graph_sample_loop(self, denoise_fn, sub, device, condition, condition_cross, clip_denoised,
                                      self._total_steps(keep_running), noise_fn)
            print('last:', 0, self.num_timesteps, len(self.betas))
            tr, sz, bb = self.translation_dim, self.size_dim, self.bbox_dim
            img_t = torch.cat([img_t[:, :, 0:tr], input_boxes[:, :, tr:tr + sz], img_t[:, :, tr:],
                               input_boxes[:, :, bb:]], dim=-1).contiguous()
            assert img_t.shape == shape
            return img_t
        img_t = noise_fn(size=(shape[0], shape[1], self.translation_dim + self.angle_dim), dtype=torch.float,
                         device=device)
        for t in reversed(range(0, self._total_steps(keep_running))):
            t_ = torch.empty(shape[0], dtype=torch.int64, device=device).fill_(t)
            img_t = self.p_sample(denoise_fn=denoise_fn, data=img_t, t=t_, condition=condition,
                                  condition_cross=condition_cross, noise_fn=noise_fn,
                                  clip_denoised=clip_denoised, return_pred_xstart=False)
            if t == 0:
                print('last:', t, self.num_timesteps, len(self.betas))
                tr, sz, bb = self.translation_dim, self.size_dim, self.bbox_dim
                img_t = torch.cat([img_t[:, :, 0:tr], input_boxes[:, :, tr:tr + sz], img_t[:, :, tr:],
                                   input_boxes[:, :, bb:]], dim=-1).contiguous()
        assert img_t.shape == shape
        return img_t

    # ------------------------------------------------------------------ losses
    def p_losses(self, denoise_fn, data_start, t, noise=None, condition=None, condition_cross=None):
        """Training loss, reference :520-665 (loss_type 'mse')."""
        if len(data_start.shape) == 3:
            B, D, N = data_start.shape
        elif len(data_start.shape) == 4:
            B, D, M, N = data_start.shape
        assert t.shape == torch.Size([B])
        if noise is None:
            noise = torch.randn(data_start.shape, dtype=data_start.dtype, device=data_start.device)
        assert noise.shape == data_start.shape and noise.dtype == data_start.dtype
        if self.loss_type not in ('mse', 'kl'):
            raise NotImplementedError(self.loss_type)
        tb = self.tables(data_start.device)
        if self.loss_type == 'kl':
            # reference :657-660 -- a (B,) tensor only, no loss dict
            data_t = ops.q_sample(data_start.contiguous(), noise.contiguous(), t, tb["sqrt_alphas_cumprod"],
                                  tb["sqrt_one_minus_alphas_cumprod"])
            losses = self._vb_terms_bpd(denoise_fn=denoise_fn, data_start=data_start, data_t=data_t, t=t,
                                        condition=condition, condition_cross=condition_cross, clip_denoised=False,
                                        return_pred_xstart=False)
            assert losses.shape == torch.Size([B])
            return losses
        with torch.no_grad():
            data_t, v_target = ops.q_sample(data_start.contiguous(), noise.contiguous(), t, tb["sqrt_alphas_cumprod"],
                                            tb["sqrt_one_minus_alphas_cumprod"], want_v=True)
        if self.model_mean_type == 'eps':
            target = noise
        elif self.model_mean_type == 'x0':
            target = data_start
        elif self.model_mean_type == 'v':
            target = v_target
        else:
            raise NotImplementedError
        denoise_out = denoise_fn(data_t, t, condition, condition_cross)
        assert data_t.shape == data_start.shape
        assert denoise_out.shape == data_start.shape
        from ..train_loss import diffusion_losses
        return diffusion_losses(self, tb, data_start, data_t, target, denoise_out, t)

    def _vb_terms_bpd(self, denoise_fn, data_start, data_t, t, condition, condition_cross, clip_denoised: bool,
                      return_pred_xstart: bool):
        """KL(q(x_{t-1}|x_t,x_0) || p(x_{t-1}|x_t)) per scene in bits, reference :511-518."""
        true_mean, _, true_log_variance_clipped = self.q_posterior_mean_variance(x_start=data_start, x_t=data_t, t=t)
        model_mean, _, model_log_variance, pred_xstart = self.p_mean_variance(
            denoise_fn, data=data_t, t=t, condition=condition, condition_cross=condition_cross,
            clip_denoised=clip_denoised, return_pred_xstart=True)
        kl = normal_kl(true_mean, true_log_variance_clipped, model_mean, model_log_variance)
        kl = kl.mean(dim=list(range(1, len(data_start.shape)))) / np.log(2.)
        return (kl, pred_xstart) if return_pred_xstart else kl

    def descale_to_origin(self, x, minimum, maximum):
        x = (x + 1) / 2
        x = x * (maximum - minimum)[None, None, :] + minimum[None, None, :]
        return x

    # ------------------------------------------------------------------ diagnostics
    def _prior_bpd(self, x_start):
        """KL(q(x_T|x_0) || N(0,I)) per scene in bits, reference :679-688."""
        with torch.no_grad():
            B, T = x_start.shape[0], self.num_timesteps
            t_ = torch.empty(B, dtype=torch.int64, device=x_start.device).fill_(T - 1)
            qt_mean, _, qt_log_variance = self.q_mean_variance(x_start, t=t_)
            kl_prior = normal_kl(mean1=qt_mean, logvar1=qt_log_variance,
                                 mean2=torch.tensor([0.]).to(qt_mean), logvar2=torch.tensor([0.]).to(qt_log_variance))
            assert kl_prior.shape == x_start.shape
            return kl_prior.mean(dim=list(range(1, len(kl_prior.shape)))) / np.log(2.)

    def calc_bpd_loop(self, denoise_fn, x_start, condition, condition_cross, clip_denoised=True):
        """Variational bound over all T timesteps, reference :690-717.  Same draw order (one q_sample draw per
        timestep, T-1 first); the (B,T) tables are filled by column instead of the reference's mask arithmetic."""
        with torch.no_grad():
            B, T = x_start.shape[0], self.num_timesteps
            vals_bt_ = torch.zeros([B, T], device=x_start.device)
            mse_bt_ = torch.zeros([B, T], device=x_start.device)
            for t in reversed(range(T)):
                t_b = torch.empty(B, dtype=torch.int64, device=x_start.device).fill_(t)
                new_vals_b, pred_xstart = self._vb_terms_bpd(
                    denoise_fn, data_start=x_start, data_t=self.q_sample(x_start=x_start, t=t_b), t=t_b,
                    condition=condition, condition_cross=condition_cross, clip_denoised=clip_denoised,
                    return_pred_xstart=True)
                assert pred_xstart.shape == x_start.shape
                new_mse_b = ((pred_xstart - x_start) ** 2).mean(dim=list(range(1, len(x_start.shape))))
                assert new_vals_b.shape == new_mse_b.shape == torch.Size([B])
                vals_bt_[:, t] = new_vals_b
                mse_bt_[:, t] = new_mse_b
            prior_bpd_b = self._prior_bpd(x_start)
            total_bpd_b = vals_bt_.sum(dim=1) + prior_bpd_b
            assert vals_bt_.shape == mse_bt_.shape == torch.Size([B, T]) and \
                total_bpd_b.shape == prior_bpd_b.shape == torch.Size([B])
            return total_bpd_b.mean(), vals_bt_.mean(), prior_bpd_b.mean(), mse_bt_.mean()


def _use_graph(graph, noise_fn, denoise_fn=None):
    """Does this reverse loop run as the replayed hipGraph step (sampler.py)?  ``graph=True`` / ``False`` decide; None (what the
    reference's call sites pass) = yes by default since round 6 -- the captured loop is bit-identical to the eager one and not
    host-bound at small batches -- unless DSC_GRAPH=0, a custom ``noise_fn`` (its draws cannot be captured) or a ``denoise_fn`` that
    is not DiffusionPoint._denoise over this package's Unet1D."""
    from ..sampler import NoiseReplay
    capturable = noise_fn is torch.randn or isinstance(noise_fn, NoiseReplay)
    if graph is None:
        if os.environ.get("DSC_GRAPH", "1") == "0" or not capturable:
            return False
        from .denoise_net import Unet1D
        return isinstance(getattr(getattr(denoise_fn, "__self__", None), "model", None), Unet1D)
    return bool(graph) and capturable


class DiffusionPoint(nn.Module):
    def __init__(self, denoise_net, config, schedule_type='linear', beta_start=0.0001, beta_end=0.02, time_num=1000,
                 loss_type='mse', model_mean_type='eps', model_var_type='fixedsmall', loss_separate=False,
                 loss_iou=False, train_stats_file=None):
        super(DiffusionPoint, self).__init__()
        betas = get_betas(schedule_type, beta_start, beta_end, time_num)
        self.diffusion = GaussianDiffusion(config, betas, loss_type, model_mean_type, model_var_type, loss_separate,
                                           loss_iou, train_stats_file)
        self.model = denoise_net

    def prior_kl(self, x0):
        return self.diffusion._prior_bpd(x0)

    def all_kl(self, x0, condition, condition_cross, clip_denoised=True):
        total_bpd_b, vals_bt, prior_bpd_b, mse_bt = self.diffusion.calc_bpd_loop(self._denoise, x0, condition,
                                                                                 condition_cross, clip_denoised)
        return {'total_bpd_b': total_bpd_b, 'terms_bpd': vals_bt, 'prior_bpd_b': prior_bpd_b, 'mse_bt': mse_bt}

    def _denoise(self, data, t, condition, condition_cross):
        B, D, N = data.shape
        assert data.dtype == torch.float
        assert t.shape == torch.Size([B]) and t.dtype == torch.int64
        out = self.model(data, t, condition, condition_cross)
        assert out.shape == torch.Size([B, D, N])
        return out

    def get_loss_iter(self, data, noises=None, condition=None, condition_cross=None):
        """reference :758-772: draws t (RNG draw #1), p_losses draws the noise (draw #2)."""
        if len(data.shape) == 3:
            B, D, N = data.shape
        elif len(data.shape) == 4:
            B, D, M, N = data.shape
        t = torch.randint(0, self.diffusion.num_timesteps, size=(B,), device=data.device)
        if noises is not None:
            noises[t != 0] = torch.randn((t != 0).sum(), *noises.shape[1:]).to(noises)
        losses, loss_dict = self.diffusion.p_losses(denoise_fn=self._denoise, data_start=data, t=t, noise=noises,
                                                    condition=condition, condition_cross=condition_cross)
        assert losses.shape == t.shape == torch.Size([B])
        return losses.mean(), loss_dict

    def gen_samples(self, shape, device, condition=None, condition_cross=None, noise_fn=torch.randn,
                    clip_denoised=True, keep_running=False, graph=None):
        return self.diffusion.p_sample_loop(self._denoise, shape=shape, device=device, condition=condition,
                                            condition_cross=condition_cross, noise_fn=noise_fn,
                                            clip_denoised=clip_denoised, keep_running=keep_running, graph=graph)

    def gen_sample_traj(self, shape, device, freq, condition=None, condition_cross=None, noise_fn=torch.randn,
                        clip_denoised=True, keep_running=False):
        return self.diffusion.p_sample_loop_trajectory(self._denoise, shape=shape, device=device, condition=condition,
                                                       condition_cross=condition_cross, noise_fn=noise_fn, freq=freq,
                                                       clip_denoised=clip_denoised, keep_running=keep_running)

    def gen_samples_ddim(self, *args, **kwargs):
        return self.diffusion.ddim_sample_loop(*args, **kwargs)

    def complete_samples(self, shape, device, condition=None, condition_cross=None, noise_fn=torch.randn,
                         clip_denoised=True, keep_running=False, partial_boxes=None, graph=None):
        return self.diffusion.p_sample_loop_complete(self._denoise, shape=shape, device=device, condition=condition,
                                                     condition_cross=condition_cross, noise_fn=noise_fn,
                                                     clip_denoised=clip_denoised, keep_running=keep_running,
                                                     partial_boxes=partial_boxes, graph=graph)

    def arrange_samples(self, shape, device, condition=None, condition_cross=None, noise_fn=torch.randn,
                        clip_denoised=True, keep_running=False, input_boxes=None, graph=None):
        return self.diffusion.p_sample_loop_arrange(self._denoise, shape=shape, device=device, condition=condition,
                                                    condition_cross=condition_cross, noise_fn=noise_fn,
                                                    clip_denoised=clip_denoised, keep_running=keep_running,
                                                    input_boxes=input_boxes, graph=graph)
