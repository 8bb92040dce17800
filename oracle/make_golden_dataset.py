"""Golden batches from the REAL reference dataset classes (build container only):  python -m oracle.make_golden_dataset

Runs CachedThreedFront + dataset_encoding_factory (scene_synthesis/datasets/threed_front.py:275-373,
threed_front_dataset.py:942-1060) over a synthetic directory in the reference's cached format
(oracle/dataset_ref.write_synth_cached_dataset) and stores the collated batches -> tests/golden/dataset.npz.
torchtext / num2words / nltk / trimesh are absent here; the modules that need them (text descriptions, raw 3D-FRONT
parsing) are stubbed -- the cached diffusion encodings never call into them.
"""
import contextlib
import importlib.util
import io
import os
import sys
import tempfile
import types

import numpy as np

from .dataset_ref import write_synth_cached_dataset
from .ref_loader import REF_ROOT

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# (name, encoding_type, augmentations, numpy seed, scene indices of the batch, max_length)
BATCHES = [
    ("bedroom_fixedrot", "cached_diffusion_cosin_angle_objfeatsnorm_lat32_wocm", ["fixed_rotations"], 5, list(range(16)), 12),
    ("bedroom_rot_jitter", "cached_diffusion_cosin_angle_objfeatsnorm_lat32_wocm", ["rotations", "jitter"], 6,
     [3, 1, 4, 1, 5, 9, 2, 6], 12),
    ("bedroom_noperm", "cached_diffusion_cosin_angle_objfeatsnorm_lat32_wocm_no_prm", None, 7, [0, 2, 7], 12),
    ("living_nofeat", "cached_diffusion_cosin_angle_wocm", ["fixed_rotations"], 8, list(range(8, 20)), 21),
]
N_SCENES = 24


def load_reference_datasets():
    ds_dir = os.path.join(REF_ROOT, "scene_synthesis", "datasets")
    stubs = {"tkinter": {"E": "e"}, "torchtext": {}, "num2words": {"num2words": lambda *a, **k: ""}, "nltk": {},
             "nltk.tokenize": {"word_tokenize": lambda s: s.split()}}
    for name, attrs in stubs.items():
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
    pkg = types.ModuleType("dsc_refds2")
    pkg.__path__ = [ds_dir]
    sys.modules["dsc_refds2"] = pkg
    for name, attrs in {"threed_front_scene": {"Room": object, "Asset": object, "ModelInfo": object},
                        "utils": {"parse_threed_front_scenes": None, "parse_threed_future_models": None},
                        "utils_text": {"compute_rel": None, "get_article": None}}.items():
        m = types.ModuleType("dsc_refds2." + name)
        m.__dict__.update(attrs)
        sys.modules["dsc_refds2." + name] = m

    def load(mod):
        spec = importlib.util.spec_from_file_location("dsc_refds2." + mod, os.path.join(ds_dir, mod + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules["dsc_refds2." + mod] = m
        spec.loader.exec_module(m)
        return m
    load("common")
    return load("threed_front"), load("threed_front_dataset")


def main():
    tf, tfd = load_reference_datasets()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, enc, augs, seed, idx, max_len in BATCHES:
            root = os.path.join(tmp, "ds_%d" % max_len)
            ids = write_synth_cached_dataset(root, N_SCENES, seed=0, max_length=max_len)
            cfg = {"train_stats": "dataset_stats.txt", "room_layout_size": "64,64", "max_length": max_len}
            with contextlib.redirect_stdout(io.StringIO()):
                raw = tf.CachedThreedFront(root, config=cfg, scene_ids=set(ids))
                ds = tfd.dataset_encoding_factory(enc, raw, augs, None)
            np.random.seed(seed)
            batch = ds.collate_fn([ds[i] for i in idx])
            for k, v in batch.items():
                if k == "room_layout":
                    continue
                out["%s.%s" % (name, k)] = v.numpy()
            if name == "bedroom_fixedrot":
                post = ds.post_process({k: v.numpy() for k, v in batch.items() if k not in ("room_layout", "length")})
                for k, v in post.items():
                    out["post.%s" % k] = np.asarray(v)
            print(name, {k: tuple(v.shape) for k, v in batch.items()})
    np.savez_compressed(os.path.join(GOLDEN, "dataset.npz"), **out)


if __name__ == "__main__":
    main()
