"""FoldingNet KL auto-encoder on the HIP kernels vs the REAL reference (tests/golden/foldingnet.npz, produced by
oracle/make_golden_foldingnet.py from scene_synthesis/networks/foldingnet_autoencoder.py with the same seeded weights and
clouds).  Tolerances: 1e-4 relative on activations / losses (the north-star tolerance), 2e-3 on gradient norms and 1e-3 relative
L2 on whole gradient tensors.  Why gradients are not held to 1e-4: the network routes gradients through discrete choices (kNN
neighbour sets, max-pool arg-max, Chamfer arg-min) and BatchNorm statistics over B*N values, and the reference's OWN fp32 gradients
are 3.5e-2 (B=4, N=256) / 7e-3 (B=32, N=2048) away from an fp64 evaluation of the same reference module on the same inputs
(`*_fp64` entries of tests/golden/foldingnet_grads.json, oracle/make_golden_foldingnet.py; asserted below) -- 2e-3 against the fp32
reference is 17x tighter than the reference's own rounding sensitivity.  The second golden is the reference's training shape,
32 clouds x 2048 points -> 2025-point folds (foldingnet_autoencoder.py:337-390, :425)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import weights as W

B, N, LATENT = 4, 256, 32


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu().flatten(), torch.as_tensor(b).double().cpu().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def build(train=True):
    from diffuscene_amd.networks.foldingnet_autoencoder import KLAutoEncoder
    m = KLAutoEncoder(latent_dim=LATENT, kl_weight=0.001)
    m.load_state_dict(W.synth_module_state(m, seed=3))
    m = m.to("cuda:0")
    return m.train() if train else m.eval()


def test_state_dict_layout_is_the_reference_layout(golden_dir):
    from diffuscene_amd.networks.foldingnet_autoencoder import KLAutoEncoder
    ref = json.load(open(os.path.join(golden_dir, "foldingnet_grads.json")))["state_dict"]
    mine = {k: list(v.shape) for k, v in KLAutoEncoder(latent_dim=LATENT).state_dict().items()}
    assert mine == ref


@pytest.mark.gpu
def test_knn_and_encoder_match_reference(golden_dir):
    from diffuscene_amd.networks.foldingnet_autoencoder import knn16
    g = np.load(os.path.join(golden_dir, "foldingnet.npz"))
    pc = W.synth_point_clouds(B, N, seed=5).to("cuda:0")
    idx = knn16(pc.reshape(B * N, 3).contiguous(), B, N).view(B, N, 16).cpu().numpy()
    assert (idx[:, :, 0] == np.arange(N)[None]).all()                    # nearest-first: a point is its own nearest neighbour
    same = (np.sort(idx, axis=-1) == g["knn_xyz"]).all(axis=-1).mean()
    assert same == 1.0, same
    m = build()
    code = m.encoder(pc.permute(0, 2, 1))
    assert rel(code, g["code"]) < 1e-4
    # feature kNN (Gram matrix path) against a direct torch evaluation of reference knn() on the same features
    x = torch.randn(2 * 64, 64, device="cuda:0")
    xi = x.view(2, 64, 64).permute(0, 2, 1)
    pd = -(xi ** 2).sum(1, keepdim=True) + 2 * xi.transpose(2, 1) @ xi - (xi ** 2).sum(1, keepdim=True).transpose(2, 1)
    want = np.sort(pd.topk(16, dim=-1)[1].cpu().numpy(), axis=-1)
    got = np.sort(knn16(x, 2, 64).view(2, 64, 16).cpu().numpy(), axis=-1)
    assert (want == got).all(axis=-1).mean() > 0.98                      # near-ties may swap the 16th neighbour


@pytest.mark.gpu
def test_forward_loss_and_gradients_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "foldingnet.npz"))
    gn = json.load(open(os.path.join(golden_dir, "foldingnet_grads.json")))
    pc = W.synth_point_clouds(B, N, seed=5).to("cuda:0")
    # ---- get_loss: Chamfer + KL, all gradient norms
    m = build()
    torch.manual_seed(11)
    loss, ld = m.get_loss({"points": pc})
    loss.backward()
    got = [float(loss.detach()), float(ld["loss.cd"].detach()), float(ld["loss.kl"].detach())]
    for a, b in zip(got, g["loss"]):
        assert abs(a - b) <= 1e-4 * abs(b), (got, g["loss"])
    big = max(gn["get_loss"].values())
    for name, p in m.named_parameters():
        want = gn["get_loss"][name]
        if want > 1e-4 * big:
            assert abs(float(p.grad.norm()) - want) <= 2e-3 * want, (name, float(p.grad.norm()), want)
    # ---- forward + a second objective: whole gradient tensors
    m = build()
    torch.manual_seed(11)
    kl, lat, rec = m(pc)
    assert rel(kl, g["kl"]) < 1e-4 and rel(lat, g["lat"]) < 1e-4 and rel(rec, g["recon"]) < 1e-4
    ((rec ** 2).mean() + kl.mean()).backward()
    assert rel(m.encoder.bn1.running_mean, g["running_mean_bn1"]) < 1e-5
    assert rel(m.encoder.bn1.running_var, g["running_var_bn1"]) < 1e-5
    assert rel_l2(m.encoder.conv1.weight.grad, g["grad_conv1"]) < 1e-3
    assert rel_l2(m.decoder.fold2.layers[0].weight.grad[:, :35, 0], g["grad_fold2_first"]) < 1e-3
    assert rel_l2(m.fc.weight.grad, g["grad_fc"]) < 1e-3
    big = max(gn["recon_sq_plus_kl"].values())
    for name, p in m.named_parameters():
        want = gn["recon_sq_plus_kl"][name]
        if want > 1e-4 * big:
            assert abs(float(p.grad.norm()) - want) <= 2e-3 * want, (name, float(p.grad.norm()), want)
    # ---- evaluation mode (running statistics)
    me = build(train=False)
    with torch.no_grad():
        torch.manual_seed(11)
        assert rel(me(pc)[2], g["recon_eval"]) < 1e-4


def _worst(a, b):
    big = max(b.values())
    return max(abs(a[n] - b[n]) / b[n] for n in a if b[n] > 1e-4 * big)


def test_reference_fp32_gradients_are_themselves_percent_level(golden_dir):
    """The yardstick behind the 2e-3 gradient tolerance: reference fp32 vs reference fp64, same module, same inputs."""
    gn = json.load(open(os.path.join(golden_dir, "foldingnet_grads.json")))
    assert 5e-3 < _worst(gn["get_loss"], gn["get_loss_fp64"]) < 1e-1
    assert 2e-3 < _worst(gn["recon_sq_plus_kl"], gn["recon_sq_plus_kl_fp64"]) < 1e-1
    assert 2e-3 < _worst(gn["b32_n2048"]["get_loss"], gn["b32_n2048"]["get_loss_fp64"]) < 1e-1


@pytest.mark.gpu
def test_training_shape_32_clouds_of_2048_points(golden_dir):
    """get_loss (Chamfer + KL) and every gradient norm at the reference's own training shape.  Losses: 2e-4 (the reference's fp32
    loss is 1.5e-4 from its fp64 evaluation at this size); gradient norms: within the reference's own fp32-vs-fp64 distance."""
    big = json.load(open(os.path.join(golden_dir, "foldingnet_grads.json")))["b32_n2048"]
    pc = W.synth_point_clouds(32, 2048, seed=6).to("cuda:0")
    m = build()
    torch.manual_seed(12)
    loss, ld = m.get_loss({"points": pc})
    loss.backward()
    got = [float(loss.detach()), float(ld["loss.cd"].detach()), float(ld["loss.kl"].detach())]
    for a, b in zip(got, big["loss"]):
        assert abs(a - b) <= 2e-4 * abs(b), (got, big["loss"])
    mine = {n: float(p.grad.norm()) for n, p in m.named_parameters()}
    e32, e64, r = _worst(mine, big["get_loss"]), _worst(mine, big["get_loss_fp64"]), _worst(big["get_loss"], big["get_loss_fp64"])
    print("B=32 N=2048 gradient norms: vs reference fp32 %.3g, vs reference fp64 %.3g (reference fp32 vs fp64: %.3g)" % (e32, e64, r))
    assert e32 <= r and e64 <= 1.5 * r, (e32, e64, r)
