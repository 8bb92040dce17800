"""3-D Chamfer distance on MI355X -- drop-in for ChamferDistancePytorch/chamfer3D/dist_chamfer_3D.py
(``chamfer_3DFunction`` :27-64, ``chamfer_3DDist`` :67-74) and fscore.py, the loss of the FoldingNet shape auto-encoder
(scene_synthesis/networks/foldingnet_autoencoder.py:9-10,381-383).  GPU tensors only, as the reference."""
import torch
from torch import nn
from torch.autograd import Function

from . import _lib


def _check(t, name):
    if not t.is_cuda:
        raise RuntimeError("diffuscene_amd.chamfer: %s must be a GPU tensor (no CPU fallback)" % name)
    if t.dtype != torch.float32 or t.dim() != 3 or t.shape[-1] != 3:
        raise RuntimeError("diffuscene_amd.chamfer: %s must be float32 (B, n, 3), got %s %s" % (name, t.dtype, tuple(t.shape)))


class chamfer_3DFunction(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        _check(xyz1, "xyz1")
        _check(xyz2, "xyz2")
        xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
        batchsize, n, _ = xyz1.size()
        _, m, _ = xyz2.size()
        assert xyz2.size(0) == batchsize
        device = xyz1.device
        dist1 = torch.empty(batchsize, n, device=device)
        dist2 = torch.empty(batchsize, m, device=device)
        idx1 = torch.empty(batchsize, n, device=device, dtype=torch.int32)
        idx2 = torch.empty(batchsize, m, device=device, dtype=torch.int32)
        with torch.cuda.device(device):
            rc = _lib.fn("dsc_chamfer3d_forward_f32")(xyz1.data_ptr(), xyz2.data_ptr(), dist1.data_ptr(), dist2.data_ptr(),
                                                      idx1.data_ptr(), idx2.data_ptr(), batchsize, n, m,
                                                      torch.cuda.current_stream(device).cuda_stream)
        _lib.check(rc, "dsc_chamfer3d_forward_f32")
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, dist2, idx1, idx2

    @staticmethod
    def backward(ctx, graddist1, graddist2, gradidx1=None, gradidx2=None):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        graddist1, graddist2 = graddist1.contiguous(), graddist2.contiguous()
        gradxyz1, gradxyz2 = torch.empty_like(xyz1), torch.empty_like(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        with torch.cuda.device(xyz1.device):
            rc = _lib.fn("dsc_chamfer3d_backward_f32")(xyz1.data_ptr(), xyz2.data_ptr(), graddist1.data_ptr(),
                                                       graddist2.data_ptr(), idx1.data_ptr(), idx2.data_ptr(),
                                                       gradxyz1.data_ptr(), gradxyz2.data_ptr(), b, n, m,
                                                       torch.cuda.current_stream(xyz1.device).cuda_stream)
        _lib.check(rc, "dsc_chamfer3d_backward_f32")
        return gradxyz1, gradxyz2


class chamfer_3DDist(nn.Module):
    def __init__(self):
        super(chamfer_3DDist, self).__init__()

    def forward(self, input1, input2):
        input1 = input1.contiguous()
        input2 = input2.contiguous()
        return chamfer_3DFunction.apply(input1, input2)


def fscore(dist1, dist2, threshold=0.001):
    """F-score between two point clouds from their squared NN distances (ChamferDistancePytorch/fscore.py)."""
    precision_1 = torch.mean((dist1 < threshold).float(), dim=1)
    precision_2 = torch.mean((dist2 < threshold).float(), dim=1)
    f = 2 * precision_1 * precision_2 / (precision_1 + precision_2)
    f[torch.isnan(f)] = 0
    return f, precision_1, precision_2
