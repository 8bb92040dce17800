"""The training objective of reference p_losses (diffusion_ddpm.py:556-652) for the AUTOGRAD path (configurations the static
training plan does not cover, and the tests' reference of the fused kernel): ``DdpmLossFn`` wraps the fused loss kernel
(all loss terms + d loss / d denoise_out in one launch) as an autograd Function; ``diffusion_losses`` dispatches to it.
The captured training graph itself lives in train_plan.py / train_step.py."""
import torch

from .networks.loss import axis_aligned_bbox_overlaps_3d


class DdpmLossFn(torch.autograd.Function):
    """losses_weight (B,) and the 9 logged per-scene terms from ONE kernel that also produces d loss / d denoise_out."""

    @staticmethod
    def forward(ctx, denoise_out, target, data_t, t, diff, tb):
        from . import ops
        ca, cb = diff._coeffs(tb)
        bounds = None
        if diff.loss_iou:
            bounds = list(diff._centroids[0]) + list(diff._centroids[1]) + list(diff._sizes[0]) + list(diff._sizes[1])
        dims = dict(translation_dim=diff.translation_dim, size_dim=diff.size_dim, bbox_dim=diff.bbox_dim,
                    class_dim=diff.class_dim, objectness_dim=diff.objectness_dim, objfeat_dim=diff.objfeat_dim)
        losses, parts, dout = ops.ddpm_loss(target.contiguous(), denoise_out.contiguous(), data_t.contiguous(), t,
                                            tb["loss_weight"], ca, cb, tb["alphas_cumprod"], bounds, dims,
                                            diff.loss_separate, diff.loss_iou,
                                            {"eps": ops.MEAN_EPS, "x0": ops.MEAN_X0, "v": ops.MEAN_V}[diff.model_mean_type])
        ctx.save_for_backward(dout)
        ctx.mark_non_differentiable(parts)
        return losses, parts

    @staticmethod
    def backward(ctx, g_losses, _g_parts):
        dout, = ctx.saved_tensors
        return dout * g_losses.reshape(-1, 1, 1), None, None, None, None, None


_PART_KEYS = ('loss.bbox', 'loss.trans', 'loss.size', 'loss.angle', 'loss.class', 'loss.object', 'loss.objfeat',
              'loss.liou', 'loss.bbox_iou')


def _mse(target, out, a, b):
    return ((target[:, :, a:b] - out[:, :, a:b]) ** 2).mean(dim=(1, 2))


def diffusion_losses(diff, tb, data_start, data_t, target, denoise_out, t, fused=True):
    """Separated MSE terms, loss_weight[t] scaling and the 3-D IoU regulariser of reference p_losses
    (diffusion_ddpm.py:558-652).  The shipped attribute layout goes through the fused HIP kernel (DdpmLossFn); the
    re-arrangement loss and exotic layouts use device torch ops under autograd (also the kernel's test reference)."""
    B = data_start.shape[0]
    tr, sz, bb = diff.translation_dim, diff.size_dim, diff.bbox_dim
    nc, no, nf = diff.class_dim, diff.objectness_dim, diff.objfeat_dim
    lw_t = tb["loss_weight"][t]
    if diff.room_arrange_condition:
        assert data_start.shape[-1] == tr + diff.angle_dim
        loss_trans = _mse(target, denoise_out, 0, tr)
        loss_angle = _mse(target, denoise_out, tr, data_start.shape[-1])
        if diff.loss_separate:
            losses = loss_trans + loss_angle
        else:
            losses = ((target - denoise_out) ** 2).mean(dim=(1, 2))
        return losses * lw_t, {'loss.trans': loss_trans.mean(), 'loss.angle': loss_angle.mean()}
    if data_start.shape[-1] != no + nc + bb + nf:
        print('unimplement point dim is: ', data_start.shape[-1])
        raise NotImplementedError
    if fused and tr == 3 and sz == 3 and data_start.shape[1] <= 160:
        # the shipped layout: one fused HIP kernel for all terms and for d loss / d denoise_out
        losses_weight, parts = DdpmLossFn.apply(denoise_out, target, data_t, t, diff, tb)
        means = parts.mean(dim=0)
        return losses_weight, {k: means[i] for i, k in enumerate(_PART_KEYS)}
    loss_trans = _mse(target, denoise_out, 0, tr)
    loss_size = _mse(target, denoise_out, tr, tr + sz)
    loss_angle = _mse(target, denoise_out, tr + sz, bb)
    loss_bbox = _mse(target, denoise_out, 0, bb)
    loss_class = _mse(target, denoise_out, bb, bb + nc)
    loss_object = _mse(target, denoise_out, bb + nc - 1, bb + nc) if no == 0 else \
        _mse(target, denoise_out, bb + nc, bb + nc + no)
    loss_objfeat = torch.zeros(B, device=data_start.device) if nf == 0 else \
        _mse(target, denoise_out, bb + nc + no, data_start.shape[-1])
    if diff.loss_separate:
        losses = loss_bbox + loss_class
        if no > 0:
            losses = losses + loss_object
        if nf > 0:
            losses = losses + loss_objfeat
    else:
        losses = ((target - denoise_out) ** 2).mean(dim=(1, 2))
    losses_weight = losses * lw_t
    if diff.loss_iou:
        if diff.model_mean_type == 'eps':
            x_recon = diff._predict_xstart_from_eps(data_t, t, eps=denoise_out)
        elif diff.model_mean_type == 'x0':
            x_recon = denoise_out
        else:
            x_recon = diff._predict_start_from_v(data_t, t, v=denoise_out)
        x_recon = torch.clamp(x_recon, -1.0, 1.0)
        if no > 0:
            valid = (x_recon[:, :, bb + nc:bb + nc + no] >= 0).float().squeeze(2)
        else:
            valid = (x_recon[:, :, bb + nc - 1:bb + nc] <= 0).float().squeeze(2)
        dev = data_start.device
        ctr = diff.descale_to_origin(x_recon[:, :, 0:tr], diff._centroids_min.to(dev), diff._centroids_max.to(dev))
        siz = diff.descale_to_origin(x_recon[:, :, tr:tr + sz], diff._sizes_min.to(dev), diff._sizes_max.to(dev))
        corners = torch.cat([ctr - siz, ctr + siz], dim=-1)
        assert corners.shape[-1] == 6
        bbox_iou = axis_aligned_bbox_overlaps_3d(corners, corners)
        mask = valid[:, :, None] * valid[:, None, :]
        iou_valid = bbox_iou * mask
        denom = mask.sum(dim=(1, 2)) + 1e-6
        bbox_iou_valid_avg = iou_valid.sum(dim=(1, 2)) / denom
        w_iou = tb["alphas_cumprod"][t].reshape(B, 1, 1)
        loss_iou_valid_avg = (w_iou * 0.1 * iou_valid).sum(dim=(1, 2)) / denom
        losses_weight = losses_weight + loss_iou_valid_avg
    else:
        loss_iou_valid_avg = torch.zeros(B, device=data_start.device)
        bbox_iou_valid_avg = torch.zeros(B, device=data_start.device)
    return losses_weight, {
        'loss.bbox': loss_bbox.mean(), 'loss.trans': loss_trans.mean(), 'loss.size': loss_size.mean(),
        'loss.angle': loss_angle.mean(), 'loss.class': loss_class.mean(), 'loss.object': loss_object.mean(),
        'loss.objfeat': loss_objfeat.mean(), 'loss.liou': loss_iou_valid_avg.mean(),
        'loss.bbox_iou': bbox_iou_valid_avg.mean(),
    }
