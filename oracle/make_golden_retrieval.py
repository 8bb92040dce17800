"""Golden vectors for shape retrieval from the REAL reference class (build container only):
python -m oracle.make_golden_retrieval"""
import importlib.util
import os
import sys
import types

import numpy as np

from .ref_loader import REF_ROOT
from .retrieval_ref import synth_objects, synth_queries

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load_reference_class():
    ds_dir = os.path.join(REF_ROOT, "scene_synthesis", "datasets")
    pkg = types.ModuleType("dsc_refds")
    pkg.__path__ = [ds_dir]
    sys.modules["dsc_refds"] = pkg
    utils = types.ModuleType("dsc_refds.utils")          # threed_future_dataset.py:4 imports a parser we never call
    utils.parse_threed_future_models = lambda *a, **k: None
    sys.modules["dsc_refds.utils"] = utils
    spec = importlib.util.spec_from_file_location("dsc_refds.threed_future_dataset",
                                                  os.path.join(ds_dir, "threed_future_dataset.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["dsc_refds.threed_future_dataset"] = m
    spec.loader.exec_module(m)
    return m.ThreedFutureDataset


def main():
    cls = load_reference_class()
    objs = synth_objects()
    ds = cls(objs)
    labels, feats, sizes = synth_queries(objs)
    index_of = {id(o): i for i, o in enumerate(objs)}
    a = np.array([index_of[id(ds.get_closest_furniture_to_objfeats(l, f))] for l, f in zip(labels, feats)], dtype=np.int32)
    b = np.array([index_of[id(ds.get_closest_furniture_to_objfeats_and_size(l, f, s))]
                  for l, f, s in zip(labels, feats, sizes)], dtype=np.int32)
    np.savez_compressed(os.path.join(GOLDEN, "retrieval.npz"), by_feat=a, by_feat_and_size=b)
    print("retrieval golden:", a[:10], b[:10])


if __name__ == "__main__":
    main()
