"""Golden vectors for the Chamfer row from the reference's own pure-torch implementation
(ChamferDistancePytorch/chamfer_python.py, the checker of its unit test):  python -m oracle.make_golden_chamfer"""
import importlib.util
import os

import numpy as np
import torch

from .chamfer_ref import synth_clouds
from .ref_loader import REF_ROOT

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
CASES = [("b4_100_200", 4, 100, 200, 0), ("b2_2048_2025", 2, 2048, 2025, 1), ("b3_1_5", 3, 1, 5, 2), ("b1_513_64", 1, 513, 64, 3)]


def main():
    spec = importlib.util.spec_from_file_location("dsc_ref_chamfer_python",
                                                  os.path.join(REF_ROOT, "ChamferDistancePytorch", "chamfer_python.py"))
    cp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cp)
    out = {}
    for name, B, n, m, seed in CASES:
        a, b = synth_clouds(B, n, m, seed)
        ta, tb = torch.from_numpy(a).requires_grad_(True), torch.from_numpy(b).requires_grad_(True)
        d1, d2, i1, i2 = cp.distChamfer(ta, tb)
        # loss of foldingnet_autoencoder.py:381-383 (means over points) -> gradients
        loss = (d1.mean(dim=1) + d2.mean(dim=1)).mean()
        loss.backward()
        out[name + ".dist1"], out[name + ".dist2"] = d1.detach().numpy(), d2.detach().numpy()
        out[name + ".idx1"], out[name + ".idx2"] = i1.numpy(), i2.numpy()
        out[name + ".grad1"], out[name + ".grad2"] = ta.grad.numpy(), tb.grad.numpy()
        print(name, float(loss))
    np.savez_compressed(os.path.join(GOLDEN, "chamfer.npz"), **out)


if __name__ == "__main__":
    main()
