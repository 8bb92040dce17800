"""Pairwise axis-aligned 3-D IoU used by the DDPM IoU loss term -- drop-in for
scene_synthesis/networks/loss.py:7-102 (mode 'iou' and 'giou', aligned or pairwise).

Device torch ops on (B, N, 6) / (B, N, N) tensors: this is the autograd-visible definition; the fused HIP
loss kernel (csrc/train.hip) evaluates the same expressions for the training hot path.
"""
import torch


def _volume(b):
    return (b[..., 3] - b[..., 0]) * (b[..., 4] - b[..., 1]) * (b[..., 5] - b[..., 2])


def axis_aligned_bbox_overlaps_3d(bboxes1, bboxes2, mode='iou', is_aligned=False, eps=1e-6):
    """bboxes (..., m, 6) / (..., n, 6) as <x1, y1, z1, x2, y2, z2> -> (..., m, n) (or (..., m) if aligned)."""
    assert mode in ['iou', 'giou'], f'Unsupported mode {mode}'
    assert (bboxes1.size(-1) == 6 or bboxes1.size(0) == 0)
    assert (bboxes2.size(-1) == 6 or bboxes2.size(0) == 0)
    assert bboxes1.shape[:-2] == bboxes2.shape[:-2]
    batch_shape = bboxes1.shape[:-2]
    rows, cols = bboxes1.size(-2), bboxes2.size(-2)
    if is_aligned:
        assert rows == cols
    if rows * cols == 0:
        return bboxes1.new(batch_shape + ((rows,) if is_aligned else (rows, cols)))
    vol1, vol2 = _volume(bboxes1), _volume(bboxes2)
    if is_aligned:
        lo1, hi1, lo2, hi2 = bboxes1[..., :3], bboxes1[..., 3:], bboxes2[..., :3], bboxes2[..., 3:]
        both = vol1 + vol2
    else:
        lo1, hi1 = bboxes1[..., :, None, :3], bboxes1[..., :, None, 3:]
        lo2, hi2 = bboxes2[..., None, :, :3], bboxes2[..., None, :, 3:]
        both = vol1[..., None] + vol2[..., None, :]
    edge = (torch.min(hi1, hi2) - torch.max(lo1, lo2)).clamp(min=0)
    overlap = edge[..., 0] * edge[..., 1] * edge[..., 2]
    floor = overlap.new_tensor([eps])
    union = torch.max(both - overlap, floor)
    ious = overlap / union
    if mode == 'iou':
        return ious
    hull = (torch.max(hi1, hi2) - torch.min(lo1, lo2)).clamp(min=0)
    hull_vol = torch.max(hull[..., 0] * hull[..., 1] * hull[..., 2], floor)
    return ious - (hull_vol - union) / hull_vol
