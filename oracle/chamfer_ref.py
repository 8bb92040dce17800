"""CPU restatement of the reference's 3-D Chamfer distance (TEST INFRASTRUCTURE; only tests/ and tools/ bench baselines
import this).

Reference: ChamferDistancePytorch/chamfer3D/chamfer3D.cu:12-131 (NmDistanceKernel: for every point of cloud A the
squared distance to, and index of, the nearest point of cloud B; strict '<' while scanning B in index order, so ties go
to the LOWEST index) and :149-171 (NmDistanceGradKernel: d dist1[j] / d xyz1[j] = 2 (p1_j - p2_idx), the negative is
scattered onto xyz2[idx]).  The CUDA sources cannot be built here (no CUDA); the restatement is pinned against the
reference's own pure-torch implementation ChamferDistancePytorch/chamfer_python.py:18-40, which is what the reference's
unit test (unit_test.py:15-36) checks the CUDA kernels against (mean squared distance error < 1e-8, identical indices).

Arithmetic: fp32, d = ((dx*dx + dy*dy) + dz*dz) with separately rounded products (the product kernel is compiled with
-ffp-contract=off so both sides agree bit for bit).
"""
import numpy as np


def chamfer_forward(xyz1, xyz2):
    """xyz1 (B,n,3), xyz2 (B,m,3) float32 -> dist1 (B,n), dist2 (B,m) float32, idx1, idx2 int32."""
    xyz1 = np.asarray(xyz1, np.float32)
    xyz2 = np.asarray(xyz2, np.float32)

    def one_way(a, b):
        d = a[:, :, None, :] - b[:, None, :, :]                      # (B,n,m,3) fp32
        sq = d * d
        dist = (sq[..., 0] + sq[..., 1]) + sq[..., 2]
        idx = dist.argmin(axis=2).astype(np.int32)                   # first minimum = lowest index
        return np.take_along_axis(dist, idx[..., None].astype(np.int64), 2)[..., 0], idx
    d1, i1 = one_way(xyz1, xyz2)
    d2, i2 = one_way(xyz2, xyz1)
    return d1, d2, i1, i2


def chamfer_backward(xyz1, xyz2, g1, g2, idx1, idx2):
    """Gradients of sum(g1*dist1) + sum(g2*dist2) w.r.t. both clouds (chamfer3D.cu:149-171), accumulated in float64 and
    rounded once (the reference accumulates with fp32 atomics in arbitrary order)."""
    xyz1, xyz2 = np.asarray(xyz1, np.float64), np.asarray(xyz2, np.float64)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    gx1, gx2 = np.zeros_like(xyz1), np.zeros_like(xyz2)
    for b in range(B):
        t = 2.0 * np.asarray(g1[b], np.float64)[:, None] * (xyz1[b] - xyz2[b][idx1[b]])
        gx1[b] += t
        np.add.at(gx2[b], idx1[b], -t)
        t = 2.0 * np.asarray(g2[b], np.float64)[:, None] * (xyz2[b] - xyz1[b][idx2[b]])
        gx2[b] += t
        np.add.at(gx1[b], idx2[b], -t)
    return gx1.astype(np.float32), gx2.astype(np.float32)


def synth_clouds(B, n, m, seed, dup=False):
    rng = np.random.RandomState(seed)
    a = rng.rand(B, n, 3).astype(np.float32)
    b = rng.rand(B, m, 3).astype(np.float32)
    if dup:                          # exact duplicates -> ties, lowest index must win
        b[:, m // 2:] = b[:, :m - m // 2]
        a[:, ::7] = b[:, :len(a[0, ::7])] if m >= len(a[0, ::7]) else a[:, ::7]
    return a, b
