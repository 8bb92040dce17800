"""Golden outputs of the REAL reference FoldingNet KL auto-encoder (build container only; TEST INFRASTRUCTURE).

    python -m oracle.make_golden_foldingnet   ->  tests/golden/foldingnet.npz (+ foldingnet_grads.json)

Runs scene_synthesis/networks/foldingnet_autoencoder.py (KLAutoEncoder, latent 32, training-mode BatchNorm) of the reference on
seeded clouds (B=4, N=256) with seeded weights (oracle/weights.synth_module_state, keyed by parameter name): encoder code, kNN
neighbour sets of the input cloud, kl, latent, reconstruction, get_loss (Chamfer through the reference's own chamfer_python
stand-in for its CUDA extension) and the gradient norm of every parameter; one evaluation-mode reconstruction as well."""
import json
import os

import numpy as np
import torch

from . import weights as W
from .make_golden import GOLDEN
from .ref_loader import load_reference_package

B, N, LATENT = 4, 256, 32


def main():
    torch.set_num_threads(8)
    ref = load_reference_package()["foldingnet_autoencoder"]
    model = ref.KLAutoEncoder(latent_dim=LATENT, kl_weight=0.001)
    model.load_state_dict(W.synth_module_state(model, seed=3))
    pc = W.synth_point_clouds(B, N, seed=5)
    out = {}
    # neighbour sets of the raw cloud
    out["knn_xyz"] = np.sort(ref.knn(pc.permute(0, 2, 1), k=16).numpy(), axis=-1).astype(np.int32)
    model.train()
    torch.manual_seed(11)                        # posterior.sample(): torch.randn on the CPU generator
    code = model.encoder(pc.permute(0, 2, 1))
    out["code"] = code.detach().numpy()
    model.load_state_dict(W.synth_module_state(model, seed=3))          # undo the running-stat update of the probe above
    torch.manual_seed(11)
    loss, ld = model.get_loss({"points": pc})
    loss.backward()
    out["loss"] = np.array([float(loss.detach()), float(ld["loss.cd"].detach()), float(ld["loss.kl"].detach())], dtype=np.float64)
    grads_loss = {n: float(p.grad.norm()) for n, p in model.named_parameters()}
    torch.manual_seed(11)
    model.zero_grad()
    model.load_state_dict(W.synth_module_state(model, seed=3))
    kl, lat, rec = model(pc)
    out["kl"], out["lat"], out["recon"] = kl.detach().numpy(), lat.detach().numpy(), rec.detach().numpy()
    loss2 = (rec ** 2).mean() + kl.mean()
    loss2.backward()
    grads = {n: float(p.grad.norm()) for n, p in model.named_parameters()}
    out["running_mean_bn1"] = model.encoder.bn1.running_mean.numpy().copy()
    out["running_var_bn1"] = model.encoder.bn1.running_var.numpy().copy()
    out["grad_conv1"] = model.encoder.conv1.weight.grad.numpy().copy()
    out["grad_fold2_first"] = model.decoder.fold2.layers[0].weight.grad.numpy()[:, :35, 0].copy()    # point part + 32 codeword columns
    out["grad_fc"] = model.fc.weight.grad.numpy().copy()
    model.eval()
    model.load_state_dict(W.synth_module_state(model, seed=3))
    with torch.no_grad():
        torch.manual_seed(11)
        out["recon_eval"] = model(pc)[2].numpy()
    # ---- the same two objectives evaluated by the SAME reference module in fp64: how far the reference's own fp32 gradients are from
    #      the exact ones (BatchNorm statistics over B*N values, max-pool arg-max routing, Chamfer arg-mins) -- the yardstick for the
    #      tolerance of the gradient checks (tests/test_gpu_foldingnet.py)
    def grads64(objective, clouds):
        m64 = ref.KLAutoEncoder(latent_dim=LATENT, kl_weight=0.001).double()
        m64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in W.synth_module_state(model, seed=3).items()})
        m64.train()
        torch.manual_seed(11)
        if objective == "get_loss":
            l64, _ = m64.get_loss({"points": clouds.double()})
        else:
            kl64, _, rec64 = m64(clouds.double())
            l64 = (rec64 ** 2).mean() + kl64.mean()
        l64.backward()
        return {n: float(p.grad.norm()) for n, p in m64.named_parameters()}, float(l64.detach())

    g64_loss, _ = grads64("get_loss", pc)
    g64_sq, _ = grads64("recon_sq_plus_kl", pc)
    np.savez_compressed(os.path.join(GOLDEN, "foldingnet.npz"), **out)

    # ---- the reference's own training shape: 32 clouds x 2048 points -> 2025-point folds (foldingnet_autoencoder.py:337-390, :425)
    B2, N2 = 32, 2048
    pc2 = W.synth_point_clouds(B2, N2, seed=6)
    model.train()
    model.zero_grad()
    model.load_state_dict(W.synth_module_state(model, seed=3))
    torch.manual_seed(12)
    loss2k, ld2k = model.get_loss({"points": pc2})
    loss2k.backward()
    big = {"loss": [float(loss2k.detach()), float(ld2k["loss.cd"].detach()), float(ld2k["loss.kl"].detach())],
           "get_loss": {n: float(p.grad.norm()) for n, p in model.named_parameters()}}
    m64 = ref.KLAutoEncoder(latent_dim=LATENT, kl_weight=0.001).double()
    m64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in W.synth_module_state(model, seed=3).items()})
    m64.train()
    torch.manual_seed(12)
    l64, _ = m64.get_loss({"points": pc2.double()})
    l64.backward()
    big["get_loss_fp64"] = {n: float(p.grad.norm()) for n, p in m64.named_parameters()}
    big["loss_fp64"] = float(l64.detach())
    with open(os.path.join(GOLDEN, "foldingnet_grads.json"), "w") as f:
        json.dump({"recon_sq_plus_kl": grads, "get_loss": grads_loss, "recon_sq_plus_kl_fp64": g64_sq, "get_loss_fp64": g64_loss,
                   "b32_n2048": big,
                   "state_dict": {k: list(v.shape) for k, v in model.state_dict().items()}}, f, indent=0)
    worst = max(abs(grads_loss[n] - g64_loss[n]) / g64_loss[n] for n in grads_loss if g64_loss[n] > 1e-4 * max(g64_loss.values()))
    worst2 = max(abs(big["get_loss"][n] - big["get_loss_fp64"][n]) / big["get_loss_fp64"][n] for n in big["get_loss"]
                 if big["get_loss_fp64"][n] > 1e-4 * max(big["get_loss_fp64"].values()))
    print("reference fp32 vs fp64 gradient norms: worst relative difference %.3g (B=4, N=256), %.3g (B=32, N=2048)" % (worst, worst2))
    print("wrote foldingnet.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
