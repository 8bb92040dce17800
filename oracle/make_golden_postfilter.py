"""Golden outputs of the REAL reference post-filter (build container only; TEST INFRASTRUCTURE).

    python -m oracle.make_golden_postfilter   ->  tests/golden/postfilter.npz

Calls DiffusionSceneLayout_DDPM.delete_empty_from_network_samples / delete_empty_boxes of the reference
(diffusion_scene_layout_ddpm.py:351-452) -- unbound, on a stand-in ``self`` that carries the dimension attributes the methods
read -- for seeded (B, N, C) sample tensors: uncond bedroom (C=62), living (C=65), no-objfeat layout, keep_empty on/off.  The
reference method only works for batch_size 1 (its accumulators are (1, 0, .) tensors, :367-374: torch.cat fails for B > 1), so
every scene of a batch is passed on its own; the arrays are stored per scene -- that IS the per-scene mode of the device op,
and scene 0 alone is what the drop-in method returns for B = 1."""
import os
import types

import numpy as np
import torch

from . import weights as W
from .make_golden import GOLDEN
from .ref_loader import load_reference_package

CASES = {"bedroom_b3": (3, 12, 22, 32, 7), "living_b2": (2, 21, 25, 32, 8), "bedroom_b1_noobjfeat": (1, 12, 22, 0, 9)}


def case_samples(name):
    B, N, nc, nf, seed = CASES[name]
    x = W.synth_noise((B, N, 8 + nc + nf), seed, "postfilter")
    return x, nc, nf


def fake_self(nc, nf):
    return types.SimpleNamespace(translation_dim=3, size_dim=3, angle_dim=2, bbox_dim=8, class_dim=nc, objfeat_dim=nf,
                                 n_classes=nc + 1)


def main():
    ref = load_reference_package()["diffusion_scene_layout_ddpm"].DiffusionSceneLayout_DDPM
    out = {}
    for name in CASES:
        x, nc, nf = case_samples(name)
        for keep in (False, True):
            for b in range(x.shape[0]):
                r = ref.delete_empty_from_network_samples(fake_self(nc, nf), x[b:b + 1], device="cpu", keep_empty=keep)
                for k, v in r.items():
                    out["%s.keep%d.scene%d.%s" % (name, int(keep), b, k)] = v.numpy()
    np.savez_compressed(os.path.join(GOLDEN, "postfilter.npz"), **out)
    print("wrote postfilter.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
