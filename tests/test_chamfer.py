"""Chamfer row.  CPU: oracle restatement vs the reference's own pure-torch implementation (golden, the
checker of the reference's unit_test.py: mean squared distance error < 1e-8, identical indices).  GPU: HIP kernels vs
the oracle -- distances and indices BIT-EXACT (same fp32 expression, ties -> lowest index), gradients 1e-6."""
import os

import numpy as np
import pytest
import torch

from oracle import chamfer_ref as CR
from oracle.make_golden_chamfer import CASES


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_matches_reference_python(golden_dir, case):
    name, B, n, m, seed = case
    g = np.load(os.path.join(golden_dir, "chamfer.npz"))
    a, b = CR.synth_clouds(B, n, m, seed)
    d1, d2, i1, i2 = CR.chamfer_forward(a, b)
    assert np.mean((d1 - g[name + ".dist1"]) ** 2) + np.mean((d2 - g[name + ".dist2"]) ** 2) < 1e-8     # unit_test.py:23-25
    assert np.array_equal(i1, g[name + ".idx1"]) and np.array_equal(i2, g[name + ".idx2"])                # :27-31
    g1 = np.full((B, n), 1.0 / (n * B), np.float32)
    g2 = np.full((B, m), 1.0 / (m * B), np.float32)
    gx1, gx2 = CR.chamfer_backward(a, b, g1, g2, i1, i2)
    assert np.abs(gx1 - g[name + ".grad1"]).max() < 1e-6 and np.abs(gx2 - g[name + ".grad2"]).max() < 1e-6


def test_oracle_ties_go_to_lowest_index():
    a, b = CR.synth_clouds(1, 20, 8, 5, dup=True)
    _, _, i1, _ = CR.chamfer_forward(a, b)
    assert i1.max() < 4            # second half of b duplicates the first half: the first copy must win


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES + [("b32_2000_1000", 32, 2000, 1000, 9), ("dup", 2, 300, 64, 7)],
                         ids=[c[0] for c in CASES] + ["b32_2000_1000", "dup"])
def test_hip_chamfer_bit_exact_vs_oracle(case):
    from diffuscene_amd.chamfer import chamfer_3DDist
    name, B, n, m, seed = case
    a, b = CR.synth_clouds(B, n, m, seed, dup=(name == "dup"))
    ta = torch.from_numpy(a).cuda().requires_grad_(True)
    tb = torch.from_numpy(b).cuda().requires_grad_(True)
    d1, d2, i1, i2 = chamfer_3DDist()(ta, tb)
    if B * n * m <= 4 * 2048 * 2048:
        r1, r2, ri1, ri2 = CR.chamfer_forward(a, b)
    else:                                   # big case: oracle per batch item to bound memory
        parts = [CR.chamfer_forward(a[i:i + 1], b[i:i + 1]) for i in range(B)]
        r1, r2, ri1, ri2 = [np.concatenate([p[k] for p in parts], 0) for k in range(4)]
    assert np.array_equal(d1.detach().cpu().numpy(), r1) and np.array_equal(d2.detach().cpu().numpy(), r2)
    assert np.array_equal(i1.cpu().numpy(), ri1) and np.array_equal(i2.cpu().numpy(), ri2)
    assert i1.dtype == torch.int32 and not i1.requires_grad
    # loss of foldingnet_autoencoder.py:381-383
    loss = (d1.mean(dim=1) + d2.mean(dim=1)).mean()
    loss.backward()
    g1 = np.full((B, n), 1.0 / (n * B), np.float32)
    g2 = np.full((B, m), 1.0 / (m * B), np.float32)
    gx1, gx2 = CR.chamfer_backward(a, b, g1, g2, ri1, ri2)
    s = max(np.abs(gx1).max(), np.abs(gx2).max())
    assert np.abs(ta.grad.cpu().numpy() - gx1).max() <= 1e-6 * max(s, 1) and np.abs(tb.grad.cpu().numpy() - gx2).max() <= 1e-6 * max(s, 1)


@pytest.mark.gpu
def test_hip_chamfer_matches_reference_golden_and_is_deterministic(golden_dir):
    from diffuscene_amd.chamfer import chamfer_3DFunction, fscore
    g = np.load(os.path.join(golden_dir, "chamfer.npz"))
    name, B, n, m, seed = CASES[1]
    a, b = CR.synth_clouds(B, n, m, seed)
    ta, tb = torch.from_numpy(a).cuda().requires_grad_(True), torch.from_numpy(b).cuda().requires_grad_(True)
    d1, d2, i1, i2 = chamfer_3DFunction.apply(ta, tb)
    assert float(((d1.cpu() - torch.from_numpy(g[name + ".dist1"])) ** 2).mean()) < 1e-8
    assert np.array_equal(i1.cpu().numpy(), g[name + ".idx1"]) and np.array_equal(i2.cpu().numpy(), g[name + ".idx2"])
    (d1.mean(dim=1) + d2.mean(dim=1)).mean().backward()
    assert np.abs(ta.grad.cpu().numpy() - g[name + ".grad1"]).max() < 1e-6
    g_first = ta.grad.clone()
    ta.grad = None
    d1b, d2b, _, _ = chamfer_3DFunction.apply(ta, tb)
    (d1b.mean(dim=1) + d2b.mean(dim=1)).mean().backward()
    assert torch.equal(ta.grad, g_first)                 # no atomics: bitwise reproducible
    f, p1, p2 = fscore(d1.detach(), d2.detach(), threshold=0.001)
    assert f.shape == (B,) and float(f.min()) >= 0
    with pytest.raises(RuntimeError):
        chamfer_3DFunction.apply(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3))
