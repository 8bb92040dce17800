"""Shape retrieval: oracle vs golden indices from the real reference class (CPU), HIP kernel vs oracle
(GPU, index-exact incl. ties and duplicate codes)."""
import os

import numpy as np
import pytest

from oracle import retrieval_ref as RR


def test_oracle_matches_reference_class(golden_dir):
    g = np.load(os.path.join(golden_dir, "retrieval.npz"))
    objs = RR.synth_objects()
    labels, feats, sizes = RR.synth_queries(objs)
    a = [RR.closest_to_objfeats(objs, l, f) for l, f in zip(labels, feats)]
    b = [RR.closest_to_objfeats_and_size(objs, l, f, s) for l, f, s in zip(labels, feats, sizes)]
    assert np.array_equal(np.array(a), g["by_feat"]) and np.array_equal(np.array(b), g["by_feat_and_size"])


@pytest.mark.gpu
def test_hip_retrieval_index_exact(golden_dir):
    from diffuscene_amd.retrieval import ShapeCodeIndex
    g = np.load(os.path.join(golden_dir, "retrieval.npz"))
    objs = RR.synth_objects()
    labels, feats, sizes = RR.synth_queries(objs)
    idx = ShapeCodeIndex(objs, "cuda:0")
    a = idx.closest(labels, feats).cpu().numpy()
    b = idx.closest(labels, feats, sizes).cpu().numpy()
    assert np.array_equal(a, g["by_feat"]) and np.array_equal(b, g["by_feat_and_size"])
    assert idx.get_closest_furniture_to_objfeats(labels[3], feats[3]) is objs[int(g["by_feat"][3])]
    assert idx.get_closest_furniture_to_objfeats_and_size(labels[5], feats[5], sizes[5]) is objs[int(g["by_feat_and_size"][5])]
    with pytest.raises(IndexError):
        idx.get_closest_furniture_to_objfeats("no_such_class", feats[0])


@pytest.mark.gpu
def test_hip_retrieval_large_database_vs_oracle():
    """16k objects (3D-FUTURE scale), 5120 queries (256 scenes x 20 boxes): spot-check 64 queries against the oracle."""
    from diffuscene_amd.retrieval import ShapeCodeIndex
    objs = RR.synth_objects(n=16000, n_labels=30, seed=3)
    labels, feats, sizes = RR.synth_queries(objs, q=5120, seed=4)
    idx = ShapeCodeIndex(objs, "cuda:0")
    a = idx.closest(labels, feats).cpu().numpy()
    b = idx.closest(labels, feats, sizes).cpu().numpy()
    for k in range(0, 5120, 80):
        assert a[k] == RR.closest_to_objfeats(objs, labels[k], feats[k])
        assert b[k] == RR.closest_to_objfeats_and_size(objs, labels[k], feats[k], sizes[k])
    assert (a >= 0).all()
