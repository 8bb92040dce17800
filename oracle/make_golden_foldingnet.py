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
    np.savez_compressed(os.path.join(GOLDEN, "foldingnet.npz"), **out)
    with open(os.path.join(GOLDEN, "foldingnet_grads.json"), "w") as f:
        json.dump({"recon_sq_plus_kl": grads, "get_loss": grads_loss,
                   "state_dict": {k: list(v.shape) for k, v in model.state_dict().items()}}, f, indent=0)
    print("wrote foldingnet.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
