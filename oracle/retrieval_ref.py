"""Numpy restatement of the shape retrieval of the reference (TEST INFRASTRUCTURE, see oracle/__init__.py):
ThreedFutureDataset.get_closest_furniture_to_objfeats / ..._and_size,
scene_synthesis/datasets/threed_future_dataset.py:49-77.  Pinned against the real class by tests/golden/retrieval.npz
(oracle/make_golden_retrieval.py)."""
import numpy as np


class SynthObject:
    """Stand-in for ThreedFutureModel: label, size (float64 (3,)), 32-d latent code (float32)."""

    def __init__(self, label, size, lat32):
        self.label, self.size, self._lat = label, size, lat32

    def raw_model_norm_pc_lat32(self):
        return self._lat


def synth_objects(n=500, n_labels=7, seed=0, duplicates=True):
    rng = np.random.RandomState(seed)
    objs = []
    for i in range(n):
        lat = rng.uniform(-1, 1, 32).astype(np.float32)
        size = rng.uniform(0.05, 1.5, 3)
        if duplicates and i % 50 == 49:          # exact duplicates exercise the tie rule (lowest index wins)
            lat, size = objs[i - 7]._lat.copy(), objs[i - 7].size.copy()
            lab = objs[i - 7].label
        else:
            lab = "class_%d" % rng.randint(n_labels)
        objs.append(SynthObject(lab, size, lat))
    return objs


def synth_queries(objs, q=200, seed=1):
    rng = np.random.RandomState(seed)
    labels = [objs[rng.randint(len(objs))].label for _ in range(q)]
    feats = rng.uniform(-1, 1, (q, 32)).astype(np.float32)
    sizes = rng.uniform(0.05, 1.5, (q, 3)).astype(np.float32)
    for k in range(0, q, 10):                    # some queries sit exactly on a database object (distance 0, duplicates)
        o = objs[rng.randint(len(objs))]
        labels[k], feats[k], sizes[k] = o.label, o._lat, o.size.astype(np.float32)
    return labels, feats, sizes


def closest_to_objfeats(objs, query_label, query_objfeat):
    """threed_future_dataset.py:49-59: dict of mse -> stable sort -> first."""
    cand = [(i, o) for i, o in enumerate(objs) if o.label == query_label]
    mses = [np.sum((o.raw_model_norm_pc_lat32() - query_objfeat) ** 2, axis=-1) for _, o in cand]
    order = sorted(range(len(cand)), key=lambda k: mses[k])
    return cand[order[0]][0]


def closest_to_objfeats_and_size(objs, query_label, query_objfeat, query_size):
    """threed_future_dataset.py:61-77: np.lexsort((mses_feat, mses_size))[0]."""
    cand = [(i, o) for i, o in enumerate(objs) if o.label == query_label]
    mf = [np.sum((o.raw_model_norm_pc_lat32() - query_objfeat) ** 2, axis=-1) for _, o in cand]
    ms = [np.sum((o.size - query_size) ** 2, axis=-1) for _, o in cand]
    return cand[np.lexsort((mf, ms))[0]][0]
