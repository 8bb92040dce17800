"""diffuscene_amd -- MI355X-native DDPM training / sampling path of DiffuScene.

Drop-in for the hot path of ``scene_synthesis.networks`` (Unet1D, GaussianDiffusion, DiffusionPoint,
DiffusionSceneLayout_DDPM, build_network ...) running on hand-written HIP kernels (csrc/) through the C ABI of
include/diffuscene_hip.h.  ``install_as_scene_synthesis()`` (see compat.py / INTEGRATION.md) makes the reference
scripts pick this implementation up unchanged.
"""
__version__ = "0.1.0"

from .compat import install_as_scene_synthesis  # noqa: F401
