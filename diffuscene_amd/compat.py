"""Make the reference scripts (scripts/train_diffusion.py, generate_diffusion.py, completion_rearrange.py) use this
implementation without editing them: they import ``scene_synthesis.networks`` and ``scene_synthesis.stats_logger``;
``install_as_scene_synthesis()`` registers our modules under those names (before the reference package is
imported) and leaves every other ``scene_synthesis.*`` sub-package (datasets, utils) to the reference tree."""
import importlib
import sys
import types


def install_as_scene_synthesis(reference_root=None):
    """Alias diffuscene_amd.networks / .stats_logger as scene_synthesis.networks / .stats_logger.

    reference_root: optional path of the DiffuScene checkout whose ``scene_synthesis/datasets`` etc. should keep
    resolving (its directory is put on the package __path__ so only the aliased sub-modules are replaced)."""
    import diffuscene_amd.networks as nets
    import diffuscene_amd.stats_logger as sl
    pkg = sys.modules.get("scene_synthesis")
    if pkg is None:
        pkg = types.ModuleType("scene_synthesis")
        pkg.__path__ = []
        sys.modules["scene_synthesis"] = pkg
    if reference_root is not None:
        import os
        p = os.path.join(reference_root, "scene_synthesis")
        if p not in pkg.__path__:
            pkg.__path__.append(p)
    sys.modules["scene_synthesis.networks"] = nets
    sys.modules["scene_synthesis.stats_logger"] = sl
    pkg.networks, pkg.stats_logger = nets, sl
    for sub in ("denoise_net", "diffusion_ddpm", "diffusion_scene_layout_ddpm", "loss"):
        sys.modules["scene_synthesis.networks." + sub] = importlib.import_module("diffuscene_amd.networks." + sub)
    return pkg
