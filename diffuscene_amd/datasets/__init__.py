"""Training input pipeline with the dataset resident in HBM.

Mirrors the slice of ``scene_synthesis.datasets`` the diffusion scripts use for the *cached* 3D-FRONT format:

    CachedThreedFront(base_dir, config, scene_ids)          threed_front.py:275-470
    dataset_encoding_factory(name, dataset, augmentations)  threed_front_dataset.py:942-1060
    encoded.post_process / bounds / max_length / n_classes / feature_size / collate_fn

but instead of a per-sample numpy decorator chain executed in DataLoader workers, the whole dataset (a few thousand
rooms x <=21 objects x <=65 floats -- a few MB) is uploaded ONCE as ragged arrays, and one HIP launch per step
(csrc/dataset.hip) produces the padded, scaled, rotated, permuted (B, N, C) batch directly in device memory:

    ds  = dataset_encoding_factory(cfg["encoding_type"], CachedThreedFront(dir, cfg, ids), cfg["augmentations"])
    for sample in ds.loader(batch_size, shuffle=True, device="cuda"):     # replaces torch DataLoader + .to(device)
        train_on_batch(network, optimizer, sample, config)

The random draws (rotation angle, jitter, object permutation, shuffle order) are made on the host with the SAME generators
in the SAME order as the reference with ``num_workers=0`` (numpy global RNG per sample, torch RandomSampler for the
order), so a seeded run sees the same batches.  Text descriptions (Add_Text: nltk/num2words), raw (non-cached) 3D-FRONT
parsing, ``box_ordering`` and the autoregressive encodings are out of scope and raise NotImplementedError.
"""
import json
import os

import numpy as np
import torch

from .. import _lib


class CachedThreedFront:
    """The cached rooms of the reference (one ``<tag>/boxes.npz`` per room + ``dataset_stats.txt``), loaded into ragged
    arrays.  Constructor arguments as threed_front.py:275-310; ``scene_ids=None`` keeps every room."""

    def __init__(self, base_dir, config, scene_ids=None):
        self._base_dir = base_dir
        self.config = config
        self._parse_train_stats(config["train_stats"])
        self._tags = sorted(oi for oi in os.listdir(base_dir)
                            if os.path.isdir(os.path.join(base_dir, oi)) and
                            (scene_ids is None or oi.split("_")[1] in scene_ids))
        self._path_to_rooms = [os.path.join(base_dir, t, "boxes.npz") for t in self._tags]
        cls, tr, sz, ang, f32, f64, lay, off = [], [], [], [], [], [], [], [0]
        for p in self._path_to_rooms:
            D = np.load(p)
            L = D["class_labels"].shape[0]
            cls.append(np.asarray(D["class_labels"], np.float32))
            tr.append(np.asarray(D["translations"], np.float32).reshape(L, 3))
            sz.append(np.asarray(D["sizes"], np.float32).reshape(L, 3))
            ang.append(np.asarray(D["angles"], np.float32).reshape(L))
            if "objfeats_32" in D.keys():
                f32.append(np.asarray(D["objfeats_32"], np.float32).reshape(L, -1))
            if "objfeats" in D.keys():
                f64.append(np.asarray(D["objfeats"], np.float32).reshape(L, -1))
            lay.append(D["room_layout"])
            off.append(off[-1] + L)
        if not cls:
            raise RuntimeError("no cached rooms under %s" % base_dir)
        n = len(cls)
        self.offsets = np.asarray(off, np.int64)
        self.class_labels_onehot = np.concatenate(cls, 0)
        self.translations = np.concatenate(tr, 0)
        self.sizes_ = np.concatenate(sz, 0)
        self.angles_ = np.concatenate(ang, 0)
        self.objfeats_32_ = np.concatenate(f32, 0) if len(f32) == n else None
        self.objfeats_ = np.concatenate(f64, 0) if len(f64) == n else None
        self._room_layouts = lay
        self._device_store = {}

    # ---- statistics (threed_front.py:386-419) ----
    def _parse_train_stats(self, train_stats):
        with open(os.path.join(self._base_dir, train_stats), "r") as f:
            st = json.load(f)
        self._centroids = (np.array(st["bounds_translations"][:3]), np.array(st["bounds_translations"][3:]))
        self._sizes = (np.array(st["bounds_sizes"][:3]), np.array(st["bounds_sizes"][3:]))
        self._angles = (np.array(st["bounds_angles"][0]), np.array(st["bounds_angles"][1]))

        def feat_bounds(key):
            if key in st:
                b = st[key]
                return (np.array([b[0]]), np.array([b[1]]), np.array([b[2]]))
            return (np.array([1]), np.array([-1]), np.array([1]))
        self._objfeats = feat_bounds("bounds_objfeats")
        self._objfeats_32 = feat_bounds("bounds_objfeats_32")
        self._class_labels = st["class_labels"]
        self._object_types = st["object_types"]
        self._class_frequencies = st["class_frequencies"]
        self._class_order = st["class_order"]
        self._count_furniture = st["count_furniture"]
        self._max_length = self.config.get("max_length", 12)

    class_labels = property(lambda self: self._class_labels)
    object_types = property(lambda self: self._object_types)
    class_frequencies = property(lambda self: self._class_frequencies)
    class_order = property(lambda self: self._class_order)
    count_furniture = property(lambda self: self._count_furniture)
    max_length = property(lambda self: self._max_length)
    n_classes = property(lambda self: len(self._class_labels))
    n_object_types = property(lambda self: len(self._object_types))
    centroids = property(lambda self: self._centroids)
    sizes = property(lambda self: self._sizes)
    angles = property(lambda self: self._angles)
    objfeats = property(lambda self: self._objfeats)
    objfeats_32 = property(lambda self: self._objfeats_32)

    @property
    def bounds(self):
        return {"translations": self._centroids, "sizes": self._sizes, "angles": self._angles,
                "objfeats": self._objfeats, "objfeats_32": self._objfeats_32}

    def __len__(self):
        return len(self._path_to_rooms)

    def __str__(self):
        return "Dataset contains {} scenes with {} discrete types".format(len(self), self.n_object_types)

    def post_process(self, s):
        return s

    def _get_room_layout(self, room_layout):
        from PIL import Image                       # same resize as threed_front.py:311-319
        img = Image.fromarray(room_layout[:, :, 0])
        img = img.resize(tuple(map(int, self.config["room_layout_size"].split(","))), resample=Image.BILINEAR)
        return np.asarray(img).astype(np.float32) / np.float32(255)

    def get_room_params(self, i):
        """Raw per-room arrays, threed_front.py:349-373 (host numpy; the training path never calls this)."""
        a, b = self.offsets[i], self.offsets[i + 1]
        room = self._get_room_layout(self._room_layouts[i])
        d = {"room_layout": np.transpose(room[:, :, None], (2, 0, 1)),
             "class_labels": self.class_labels_onehot[a:b], "translations": self.translations[a:b],
             "sizes": self.sizes_[a:b], "angles": self.angles_[a:b, None]}
        if self.objfeats_ is not None:
            d["objfeats"] = self.objfeats_[a:b]
        if self.objfeats_32_ is not None:
            d["objfeats_32"] = self.objfeats_32_[a:b]
        return d

    def device_store(self, device, feat_key):
        """Upload the ragged arrays once per (device, feature kind)."""
        device = torch.device(device)
        key = (str(device), feat_key)
        if key not in self._device_store:
            feats = {"objfeats_32": self.objfeats_32_, "objfeats": self.objfeats_, None: None}[feat_key]
            if feat_key is not None and feats is None:
                raise RuntimeError("the cached rooms hold no %s arrays" % feat_key)
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
            self._device_store[key] = {
                "offsets": up(self.offsets), "class_labels": up(self.class_labels_onehot),
                "translations": up(self.translations), "sizes": up(self.sizes_), "angles": up(self.angles_),
                "objfeats": up(feats) if feats is not None else None,
            }
        return self._device_store[key]


# -----------------------------------------------------------------------------------------------------------------
def draw_rot_angle(fixed, min_rad=0.174533, max_rad=5.06145):
    """RotationAugmentation.rot_angle / fixed_rot_angle (threed_front_dataset.py:330-346), incl. the reference's
    cascade of independent rand() draws, on the global numpy RNG."""
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


class DeviceEncodedScenes:
    """What ``dataset_encoding_factory`` returns for the cached diffusion encodings: the reference's decorator stack
    (RotationAugmentation -> Jitter -> Scale_CosinAngle_ObjfeatsNorm -> Permutation -> Diffusion) as one HIP launch."""

    def __init__(self, dataset, name, augmentations=None):
        self._dataset = dataset
        self.name = name
        augs = list(augmentations) if isinstance(augmentations, list) else []
        rot = [a for a in augs if a in ("rotations", "fixed_rotations")]
        if len(rot) > 1 or any(a not in ("rotations", "fixed_rotations", "jitter") for a in augs):
            raise NotImplementedError("augmentations %r" % (augmentations,))
        if "jitter" in augs and rot and augs.index("jitter") < augs.index(rot[0]):
            raise NotImplementedError("jitter before rotation is not supported (shipped configs rotate first)")
        self._rotation = rot[0] if rot else None
        self._jitter = "jitter" in augs
        self._eval = "eval" in name
        self._permute = ("wocm" in name) and ("wocm_no_prm" not in name) and not self._eval
        self._feat_key = None
        self._permute_feats = False
        if "objfeats" in name:
            self._feat_key = "objfeats_32" if "lat32" in name else "objfeats"
            self._permute_feats = True
        elif dataset.objfeats_32_ is not None:
            self._feat_key = "objfeats_32"          # present in the room dict, scaled but not permuted (:1030-1037)
        b = dataset.bounds
        fb = b[self._feat_key] if self._feat_key else (np.array([1.]), np.array([-1.]), np.array([1.]))
        self._bounds = np.concatenate([b["translations"][0], b["translations"][1], b["sizes"][0], b["sizes"][1],
                                       [float(b["angles"][0])], [float(fb[1][0])], [float(fb[2][0])]]).astype(np.float64)
        lengths = np.diff(dataset.offsets)
        if not self._eval and lengths.max() > dataset.max_length:
            raise RuntimeError("a cached room holds %d objects > max_length %d" % (lengths.max(), dataset.max_length))

    # ---- reference surface ----
    bounds = property(lambda self: self._dataset.bounds)
    n_classes = property(lambda self: self._dataset.n_classes)
    class_labels = property(lambda self: self._dataset.class_labels)
    class_frequencies = property(lambda self: self._dataset.class_frequencies)
    n_object_types = property(lambda self: self._dataset.n_object_types)
    object_types = property(lambda self: self._dataset.object_types)
    max_length = property(lambda self: self._dataset.max_length)
    bbox_dims = property(lambda self: 7 if not self._eval else 3 + 3 + 2)      # Diffusion.bbox_dims :938-939
    feature_size = property(lambda self: self.bbox_dims + self.n_classes)

    def __len__(self):
        return len(self._dataset)

    def post_process(self, s):
        """Scale_CosinAngle_ObjfeatsNorm.post_process (:515-536): network range -> metres / radians (host numpy, as the
        reference; runs once per generated batch, after the sampler)."""
        bounds = self.bounds
        out = {}
        for k, v in s.items():
            if k in ("room_layout", "class_labels", "relations", "description", "desc_emb"):
                out[k] = v
            elif k == "angles":
                out[k] = np.arctan2(v[:, :, 1:2], v[:, :, 0:1])
            elif k in ("objfeats", "objfeats_32"):
                out[k] = (v + 1) / 2 * (bounds[k][2] - bounds[k][1]) + bounds[k][1]
            else:
                out[k] = (v + 1) / 2 * (bounds[k][1] - bounds[k][0]) + bounds[k][0]
        return self._dataset.post_process(out)

    @staticmethod
    def collate_fn(samples):
        from torch.utils.data import dataloader
        return dataloader.default_collate([s for s in samples if s is not None])

    # ---- device encoding ----
    def draw(self, indices):
        """Host draws for one batch in the reference's per-sample order (rotation, jitter, then permutation)."""
        B, N = len(indices), self.max_length
        off = self._dataset.offsets
        rot = np.zeros(B, np.float64) if self._rotation else None
        jit = np.zeros((B, 3), np.float64) if self._jitter else None
        order = np.zeros((B, N), np.int32) if self._permute else None
        for b, i in enumerate(indices):
            if self._rotation:
                rot[b] = draw_rot_angle(self._rotation == "fixed_rotations")
            if self._jitter:
                jit[b] = (np.random.normal(0, 0.01), np.random.normal(0, 0.01), np.random.normal(0, 0.01))
            if self._permute:
                L = int(off[i + 1] - off[i])
                order[b, :L] = np.random.permutation(L)
        return rot, jit, order

    def encode(self, indices, device=None, draws=None):
        """-> sample_params dict of DEVICE tensors for the given room indices (keys/shape/dtype of
        Diffusion.collate_fn, :927-936, minus room_layout).  ``draws`` injects (rot, jitter, order) for tests."""
        if self._eval:
            raise NotImplementedError("'eval' encodings are only used for post_process / statistics")
        lib = _lib.load()
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("diffuscene_amd.datasets: batches are encoded on the GPU; got device %s" % device)
        st = self._dataset.device_store(device, self._feat_key)
        indices = [int(i) for i in indices]
        rot, jit, order = self.draw(indices) if draws is None else draws
        B, N = len(indices), self.max_length
        n_cls_in = st["class_labels"].shape[1]
        fd = st["objfeats"].shape[1] if st["objfeats"] is not None else 0
        C = 8 + n_cls_in - 1 + fd
        up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(device, non_blocking=True)
        scene = up(np.asarray(indices), np.int64)
        rot_d = up(rot, np.float64) if rot is not None else None
        jit_d = up(jit, np.float64) if jit is not None else None
        ord_d = up(order, np.int32) if order is not None else None
        out = torch.empty((B, N, C), device=device, dtype=torch.float32)
        length = torch.empty((B,), device=device, dtype=torch.int64)
        import ctypes as C_
        bounds = (C_.c_double * 15)(*self._bounds.tolist())
        ptr = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(device):
            rc = lib.dsc_encode_scene_batch_f32(
                ptr(st["offsets"]), ptr(st["class_labels"]), ptr(st["translations"]), ptr(st["sizes"]), ptr(st["angles"]),
                ptr(st["objfeats"]), n_cls_in, fd, ptr(scene), ptr(ord_d), ptr(rot_d), ptr(jit_d),
                int(self._permute_feats), bounds, out.data_ptr(), C, length.data_ptr(), B, N,
                torch.cuda.current_stream(device).cuda_stream)
        _lib.check(rc, "dsc_encode_scene_batch_f32")
        nc = n_cls_in - 1
        sample = {"translations": out[:, :, 0:3], "sizes": out[:, :, 3:6], "angles": out[:, :, 6:8],
                  "class_labels": out[:, :, 8:8 + nc], "length": length}
        if fd:
            sample[self._feat_key] = out[:, :, 8 + nc:]
        sample["_packed"] = out                      # [trans|size|angle|class|objfeat], the denoiser's channel order
        return sample

    def __getitem__(self, idx):
        s = self.encode([idx])
        return {k: (v[0].cpu().numpy() if k != "length" else int(v[0])) for k, v in s.items() if k != "_packed"}

    def loader(self, batch_size, shuffle=True, drop_last=False, device=None, rank=0, world_size=1):
        return SceneBatchLoader(self, batch_size, shuffle, drop_last, device, rank, world_size)


class SceneBatchLoader:
    """Replaces ``DataLoader(dataset, batch_size, collate_fn=dataset.collate_fn, shuffle=...)`` of train_diffusion.py:150-168
    + the ``.to(device)`` loop (:229-231).  Same batch composition as the reference with num_workers=0 under the same
    torch / numpy seeds; ``world_size > 1`` gives rank ``r`` every world_size-th batch of the common order."""

    def __init__(self, encoded, batch_size, shuffle=True, drop_last=False, device=None, rank=0, world_size=1):
        self.encoded, self.batch_size, self.shuffle, self.drop_last = encoded, int(batch_size), shuffle, drop_last
        self.device, self.rank, self.world_size = device, int(rank), int(world_size)

    def index_batches(self):
        from torch.utils.data import BatchSampler, RandomSampler, SequentialSampler
        n = len(self.encoded)
        # DataLoader draws its base seed from the default torch generator before the sampler draws its own
        torch.empty((), dtype=torch.int64).random_()
        sampler = RandomSampler(range(n)) if self.shuffle else SequentialSampler(range(n))
        batches = list(BatchSampler(sampler, self.batch_size, self.drop_last))
        if self.world_size > 1:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() == self.world_size:
                # every rank drew an epoch order from ITS generator (keeps each rank's RNG stream where a single-GPU run
                # would leave it); the shards must come from ONE order, so rank 0's is broadcast
                box = [batches]
                dist.broadcast_object_list(box, src=0)
                batches = box[0]
            usable = len(batches) // self.world_size * self.world_size
            batches = batches[self.rank:usable:self.world_size]
        return batches

    def __len__(self):
        n = len(self.encoded)
        nb = n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size
        return nb // self.world_size if self.world_size > 1 else nb

    def __iter__(self):
        for idx in self.index_batches():
            yield self.encoded.encode(idx, self.device)


def dataset_encoding_factory(name, dataset, augmentations=None, box_ordering=None):
    """threed_front_dataset.py:942-1060 for the cached diffusion encodings."""
    if box_ordering is not None:
        raise NotImplementedError("box_ordering")
    if "cached" not in name or "diffusion" not in name:
        raise NotImplementedError("only the cached diffusion encodings run on the device pipeline: %r" % name)
    if "text" in name:
        raise NotImplementedError("Add_Text (nltk / num2words sentence generation) is not part of the device pipeline")
    if not ("cosin_angle" in name or "objfeatsnorm" in name):
        raise NotImplementedError("plain Scale encodings (angle_dim=1) are not used by the shipped diffusion configs")
    if not ("eval" in name or "wocm" in name):
        raise NotImplementedError(name)
    return DeviceEncodedScenes(dataset, name, augmentations)


def get_encoded_dataset(config, filter_fn=lambda s: s, path_to_bounds=None, augmentations=None, split=("train", "val"),
                        scene_ids=None):
    """datasets/__init__.py:58-68 for ``dataset_type`` containing 'cached'.  ``scene_ids`` replaces the CSV split
    lookup when given (CSVSplitsBuilder needs the 3D-FRONT annotation file)."""
    if "cached" not in config["dataset_type"]:
        raise NotImplementedError("raw 3D-FRONT parsing is outside the device pipeline")
    if scene_ids is None and "annotation_file" in config:
        import csv
        with open(config["annotation_file"]) as f:
            scene_ids = set(r[0] for r in csv.reader(f) if len(r) > 1 and r[1] in split)
    raw = CachedThreedFront(config["dataset_directory"], config=config, scene_ids=scene_ids)
    return dataset_encoding_factory(config.get("encoding_type"), raw, augmentations, config.get("box_ordering", None))
