"""Import the REAL reference modules (build container only).

TEST INFRASTRUCTURE -- see oracle/__init__.py.

The reference package cannot be imported as a package here (tkinter, torchvision,
clip, wandb are absent), but the three hot-path files load by path once the two
unused tkinter imports of denoise_net.py:6-7 are stubbed.
``/root/reference`` does not exist on the GPU box; callers must guard with
``reference_available()``.
"""
import importlib.machinery
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("DSC_REFERENCE_ROOT", "/root/reference")
_NET_DIR = os.path.join(REF_ROOT, "scene_synthesis", "networks")
_PKG = "dsc_refnet"


def reference_available():
    return os.path.isfile(os.path.join(_NET_DIR, "denoise_net.py"))


def load_reference():
    """Return (loss_mod, denoise_net_mod, diffusion_ddpm_mod) of the reference."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    if _PKG + ".diffusion_ddpm" in sys.modules:
        return (sys.modules[_PKG + ".loss"], sys.modules[_PKG + ".denoise_net"],
                sys.modules[_PKG + ".diffusion_ddpm"])
    stubs = {"tkinter": {"E": "e"}, "tkinter.messagebox": {"NO": "no"},
             "tkinter.tix": {"Tree": object}}
    for name, attrs in stubs.items():
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
    pkg = types.ModuleType(_PKG)
    pkg.__path__ = [_NET_DIR]
    sys.modules[_PKG] = pkg

    def _load(mod, fname):
        spec = importlib.util.spec_from_file_location(
            "%s.%s" % (_PKG, mod), os.path.join(_NET_DIR, fname))
        m = importlib.util.module_from_spec(spec)
        sys.modules["%s.%s" % (_PKG, mod)] = m
        spec.loader.exec_module(m)
        return m

    loss = _load("loss", "loss.py")
    dn = _load("denoise_net", "denoise_net.py")
    dd = _load("diffusion_ddpm", "diffusion_ddpm.py")
    return loss, dn, dd


_PARENT = "dsc_refpkg"


def load_reference_package():
    """Load the reference's wrapper-level modules -- networks/diffusion_scene_layout_ddpm.py and
    networks/foldingnet_autoencoder.py -- under a synthetic parent package so that their relative imports
    (``..stats_logger``) resolve.  Absent third-party imports are stubbed: ``clip`` and ``wandb`` (never called on the pinned
    paths), and ``ChamferDistancePytorch.chamfer3D.dist_chamfer_3D.chamfer_3DDist`` -> the reference's own pure-torch
    ``chamfer_python.distChamfer`` (same outputs as its CUDA extension per its unit_test.py).  Returns a dict of modules."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    key = _PARENT + ".networks.diffusion_scene_layout_ddpm"
    if key in sys.modules:
        return {n: sys.modules[_PARENT + ".networks." + n] for n in
                ("loss", "denoise_net", "diffusion_ddpm", "diffusion_scene_layout_ddpm", "foldingnet_autoencoder")}
    load_reference()                       # tkinter stubs
    from transformers import BertModel, BertTokenizer  # noqa: F401  (import it before wandb is stubbed: accelerate probes for wandb)
    for name in ("clip", "wandb"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)

    def _load_file(modname, path):
        spec = importlib.util.spec_from_file_location(modname, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[modname] = m
        spec.loader.exec_module(m)
        return m

    cp = _load_file("dsc_ref_chamfer_python", os.path.join(REF_ROOT, "ChamferDistancePytorch", "chamfer_python.py"))
    import torch

    class chamfer_3DDist(torch.nn.Module):
        def forward(self, a, b):
            return cp.distChamfer(a, b)

    for name in ("ChamferDistancePytorch", "ChamferDistancePytorch.chamfer3D", "ChamferDistancePytorch.chamfer3D.dist_chamfer_3D"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules["ChamferDistancePytorch.chamfer3D.dist_chamfer_3D"].chamfer_3DDist = chamfer_3DDist
    parent = types.ModuleType(_PARENT)
    parent.__path__ = [os.path.join(REF_ROOT, "scene_synthesis")]
    sys.modules[_PARENT] = parent
    _load_file(_PARENT + ".stats_logger", os.path.join(REF_ROOT, "scene_synthesis", "stats_logger.py"))
    nets = types.ModuleType(_PARENT + ".networks")
    nets.__path__ = [_NET_DIR]
    sys.modules[_PARENT + ".networks"] = nets
    out = {}
    for mod in ("loss", "denoise_net", "diffusion_ddpm", "diffusion_scene_layout_ddpm", "foldingnet_autoencoder"):
        out[mod] = _load_file("%s.networks.%s" % (_PARENT, mod), os.path.join(_NET_DIR, mod + ".py"))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# executing the reference's SCRIPTS (scripts/train_diffusion.py ...) in the build container
# ---------------------------------------------------------------------------------------------------------------------
class _AnyMeta(type):
    def __getattr__(cls, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Any


class _Any(metaclass=_AnyMeta):
    """Stand-in for anything a missing third-party module exports at import time (class, function, table)."""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Any()

    def __call__(self, *a, **k):
        return _Any()

    def __iter__(self):
        return iter(())

    def __contains__(self, x):
        return False

    def __getitem__(self, k):
        return _Any()


def _stub_getattr(n):
    if n.startswith("__"):                             # inspect.getmodule() probes __file__ of every module in sys.modules
        raise AttributeError(n)
    return _Any


# imported by scene_synthesis/datasets (rendering, raw 3D-FRONT meshes, text generation) and absent here; none of them is touched by
# the cached-dataset training path the script takes
_SCRIPT_STUBS = ("trimesh", "simple_3dviz", "simple_3dviz.renderables", "simple_3dviz.renderables.textured_mesh", "simple_3dviz.behaviours",
                 "simple_3dviz.behaviours.keyboard", "simple_3dviz.behaviours.misc", "torchtext", "num2words", "nltk", "nltk.tokenize",
                 "nltk.corpus", "wandb",
                 # scripts/generate_diffusion.py + scripts/utils.py + scene_synthesis/utils.py: rendering / mesh IO (out of scope)
                 "simple_3dviz.utils", "simple_3dviz.behaviours.movements", "simple_3dviz.behaviours.trajectory", "simple_3dviz.behaviours.io",
                 "pyrr", "open3d", "pyvista", "seaborn", "turtle")


def prepare_reference_script_imports():
    """Make ``import train_diffusion`` (the reference's scripts/train_diffusion.py, UNCHANGED) work in the build container with
    ``scene_synthesis.networks`` / ``.stats_logger`` resolving to diffuscene_amd (diffuscene_amd.compat -- the documented drop-in
    switch) and ``scene_synthesis.datasets`` to the REFERENCE's own package.  Returns the list of stubbed third-party modules."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    load_reference()                                   # tkinter stubs
    made = []
    for name in _SCRIPT_STUBS:
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except ImportError:
                m = types.ModuleType(name)
                m.__path__ = []
                m.__spec__ = importlib.machinery.ModuleSpec(name, None)      # importlib.util.find_spec(name) must not choke on the stub
                m.__getattr__ = _stub_getattr
                sys.modules[name] = m
                made.append(name)
    from diffuscene_amd.compat import install_as_scene_synthesis
    install_as_scene_synthesis(REF_ROOT)
    scripts = os.path.join(REF_ROOT, "scripts")
    if scripts not in sys.path:
        sys.path.insert(0, scripts)
    return made
