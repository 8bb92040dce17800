"""Import the REAL reference modules (build container only).

TEST INFRASTRUCTURE -- see oracle/__init__.py.

The reference package cannot be imported as a package here (tkinter, torchvision,
clip, wandb are absent), but the three hot-path files load by path once the two
unused tkinter imports of denoise_net.py:6-7 are stubbed.
``/root/reference`` does not exist on the GPU box; callers must guard with
``reference_available()``.
"""
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
