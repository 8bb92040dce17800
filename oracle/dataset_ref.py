"""CPU restatement of the reference's training input pipeline for the diffusion encodings.

TEST INFRASTRUCTURE (see oracle/__init__.py): only tests/ may import this.  Pinned against the REAL reference classes
by oracle/make_golden_dataset.py -> tests/golden/dataset.npz.

Pipeline restated (scene_synthesis/datasets/, encoding "cached_diffusion_cosin_angle[_objfeatsnorm_lat32]_wocm"):

    CachedThreedFront.get_room_params      threed_front.py:349-373   (boxes.npz -> dict of per-object arrays)
    RotationAugmentation                   threed_front_dataset.py:313-371
    Jitter                                 threed_front_dataset.py:559-567
    Scale_CosinAngle_ObjfeatsNorm          threed_front_dataset.py:481-513  (Scale.scale :377-382)
    Permutation                            threed_front_dataset.py:570-584
    Diffusion (padding wrapper)            threed_front_dataset.py:888-925
    default_collate                        threed_front_dataset.py:927-936

numpy dtype note: the reference mixes float32 arrays with float64 bounds; under numpy 2 (this container) the clip /
normalise arithmetic runs in float64 and the final Diffusion wrapper casts to float32.  Under numpy 1.x (the
reference's pinned environment) value-based casting keeps parts in float32; the two differ by <= 1 float32 ulp, which is
the tolerance (1e-6 absolute on values in [-1, 1]) the dataset parity tests state.
"""
import json
import os

import numpy as np

N_OBJECT_TYPES = 21          # bedroom: 21 furniture classes + start + end = 23 one-hot columns (class_dim 22 after Diffusion)


def synth_stats(n_object_types=N_OBJECT_TYPES):
    names = ["type%02d" % i for i in range(n_object_types)]
    return {
        "bounds_translations": [-2.76, 0.045, -2.75, 2.78, 3.62, 2.82],
        "bounds_sizes": [0.04, 0.02, 0.01, 2.87, 1.77, 1.70],
        "bounds_angles": [-3.1415927, 3.1415927],
        "bounds_objfeats_32": [1.37, -4.8, 5.1],          # (std, min, max), preprocess_data.py:180-206
        "class_labels": names + ["start", "end"],
        "object_types": names,
        "class_frequencies": {n: 1.0 / n_object_types for n in names},
        "class_order": {n: i for i, n in enumerate(names)},
        "count_furniture": {n: 10 for n in names},
    }


def synth_scene(i, seed=0, n_object_types=N_OBJECT_TYPES, max_length=12, with_objfeats=True):
    """One cached room (the arrays CachedThreedFront.get_room_params returns), deterministic in (seed, i)."""
    rng = np.random.RandomState(1000003 * seed + i)
    L = int(rng.randint(3, max_length + 1))
    cls = np.zeros((L, n_object_types + 2), dtype=np.float32)
    cls[np.arange(L), rng.randint(0, n_object_types, size=L)] = 1.0
    s = synth_stats(n_object_types)
    lo, hi = np.array(s["bounds_translations"][:3]), np.array(s["bounds_translations"][3:])
    # a few values outside the bounds so the clip in Scale.scale is exercised
    tr = (lo + (hi - lo) * rng.uniform(-0.05, 1.05, size=(L, 3))).astype(np.float32)
    lo, hi = np.array(s["bounds_sizes"][:3]), np.array(s["bounds_sizes"][3:])
    sz = (lo + (hi - lo) * rng.uniform(-0.05, 1.05, size=(L, 3))).astype(np.float32)
    ang = rng.uniform(-np.pi, np.pi, size=(L, 1)).astype(np.float32)
    d = {"class_labels": cls, "translations": tr, "sizes": sz, "angles": ang}
    if with_objfeats:
        d["objfeats_32"] = rng.normal(0, 1.37, size=(L, 32)).astype(np.float32)
    return d


def write_synth_cached_dataset(root, n_scenes, seed=0, max_length=12, with_objfeats=True):
    """Directory in the reference's cached format: <root>/<tag>/boxes.npz + <root>/dataset_stats.txt.
    Returns the list of scene ids (the middle token of each tag, threed_front.py:283-287)."""
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, "dataset_stats.txt"), "w") as f:
        json.dump(synth_stats(), f)
    ids = []
    for i in range(n_scenes):
        sid = "SCENE%05d" % i
        tag = "Room-%05d_%s_0" % (i, sid)
        os.makedirs(os.path.join(root, tag), exist_ok=True)
        d = synth_scene(i, seed, max_length=max_length, with_objfeats=with_objfeats)
        rng = np.random.RandomState(77 + i)
        np.savez(os.path.join(root, tag, "boxes.npz"), scene_id=sid,
                 room_layout=(rng.rand(64, 64, 1) > 0.5).astype(np.uint8) * 255,
                 floor_plan_vertices=np.zeros((4, 3), np.float32), floor_plan_faces=np.zeros((2, 3), np.int64),
                 floor_plan_centroid=np.zeros(3, np.float32), **d)
        ids.append(sid)
    return ids


# ---------------------------------------------------------------------------------------------------------------
def draw_rot_angle(fixed, min_rad=0.174533, max_rad=5.06145):
    """RotationAugmentation.rot_angle / fixed_rot_angle (threed_front_dataset.py:330-346): draws from the GLOBAL numpy
    RNG in the reference's (quirky, cascading) order."""
    if fixed:
        if np.random.rand() < 0.25:
            return np.pi * 1.5
        elif np.random.rand() < 0.50:
            return np.pi
        elif np.random.rand() < 0.75:
            return np.pi * 0.5
        return 0.0
    if np.random.rand() < 0.5:
        return np.random.uniform(min_rad, max_rad)
    return 0.0


def scale(x, minimum, maximum):
    """Scale.scale, threed_front_dataset.py:377-382."""
    X = x.astype(np.float32)
    X = np.clip(X, minimum, maximum)
    X = (X - minimum) / (maximum - minimum)
    return 2 * X - 1


def encode_sample(room, stats, max_length, rot_angle=None, jitter=None, ordering=None, permute_objfeats=True):
    """One training sample after RotationAugmentation -> Jitter -> Scale_CosinAngle_ObjfeatsNorm -> Permutation ->
    Diffusion.  ``rot_angle`` None = no rotation wrapper; ``jitter`` = (d_trans, d_size, d_angle) scalars or None;
    ``ordering`` = permutation of the objects or None ('wocm_no_prm')."""
    tr, sz, ang, cls = room["translations"], room["sizes"], room["angles"], room["class_labels"]
    feats = room.get("objfeats_32")
    amin = np.array(stats["bounds_angles"][0])
    if rot_angle is not None:
        R = np.zeros((3, 3))
        R[0, 0] = np.cos(rot_angle)
        R[0, 2] = -np.sin(rot_angle)
        R[2, 0] = np.sin(rot_angle)
        R[2, 2] = np.cos(rot_angle)
        R[1, 1] = 1.
        tr = tr.dot(R)
        ang = (ang + rot_angle - amin) % (2 * np.pi) + amin
    if jitter is not None:
        tr, sz, ang = tr + jitter[0], sz + jitter[1], ang + jitter[2]
    bt, bs = stats["bounds_translations"], stats["bounds_sizes"]
    tr = scale(tr, np.array(bt[:3]), np.array(bt[3:]))
    sz = scale(sz, np.array(bs[:3]), np.array(bs[3:]))
    ang = np.concatenate([np.cos(ang), np.sin(ang)], axis=-1)
    if feats is not None:
        bf = stats.get("bounds_objfeats_32", [1, -1, 1])
        feats = scale(feats, np.array([bf[1]]), np.array([bf[2]]))
    if ordering is not None:
        tr, sz, ang, cls = tr[ordering], sz[ordering], ang[ordering], cls[ordering]
        if feats is not None and permute_objfeats:
            feats = feats[ordering]
    L = cls.shape[0]
    new_cls = np.concatenate([cls[:, :-2], cls[:, -1:]], axis=-1)
    C = new_cls.shape[1]
    end_label = np.eye(C)[-1]
    out = {"class_labels": np.vstack([new_cls, np.tile(end_label[None, :], [max_length - L, 1])]).astype(np.float32) * 2.0 - 1.0,
           "length": L}

    def pad(p):
        return np.vstack([p, np.zeros((max_length - L, p.shape[1]))]).astype(np.float32)
    out["translations"], out["sizes"], out["angles"] = pad(tr), pad(sz), pad(ang)
    if feats is not None:
        out["objfeats_32"] = pad(feats)
    return out


def encode_batch(rooms, stats, max_length, augmentations=("fixed_rotations",), permute=True, permute_objfeats=True):
    """Batch exactly as DataLoader(num_workers=0) + Diffusion.collate_fn produce it: samples are encoded one after the
    other, each consuming the global numpy RNG in wrapper order (rotation draws, jitter draws, then the permutation)."""
    outs = []
    for room in rooms:
        rot = jit = None
        for aug in augmentations or ():
            if aug == "rotations":
                rot = draw_rot_angle(False)
            elif aug == "fixed_rotations":
                rot = draw_rot_angle(True)
            elif aug == "jitter":
                jit = (np.random.normal(0, 0.01), np.random.normal(0, 0.01), np.random.normal(0, 0.01))
        order = np.random.permutation(room["class_labels"].shape[0]) if permute else None
        outs.append(encode_sample(room, stats, max_length, rot, jit, order, permute_objfeats))
    batch = {k: np.stack([o[k] for o in outs], 0) for k in outs[0] if k != "length"}
    batch["length"] = np.array([o["length"] for o in outs], dtype=np.int64)
    return batch
