"""Training input pipeline: oracle restatement vs the real reference classes (golden), host logic of the
device loader, and the HIP batch encoder vs both.  Tolerance 1e-6 absolute on the encoded attributes (values in [-1,1];
see the numpy dtype note in oracle/dataset_ref.py); class labels, lengths and padding are exact."""
import os

import numpy as np
import pytest
import torch

from oracle import dataset_ref as DR
from oracle.make_golden_dataset import BATCHES, N_SCENES

ATOL = 1e-6
KEYS = ("class_labels", "translations", "sizes", "angles", "objfeats_32")


def _rooms(idx, max_len):
    return [DR.synth_scene(i, 0, max_length=max_len) for i in idx]


def _args(enc, augs):
    return dict(augmentations=augs, permute="no_prm" not in enc, permute_objfeats="objfeats" in enc)


@pytest.mark.parametrize("case", BATCHES, ids=[b[0] for b in BATCHES])
def test_oracle_pipeline_matches_reference(golden_dir, case):
    name, enc, augs, seed, idx, max_len = case
    g = np.load(os.path.join(golden_dir, "dataset.npz"))
    np.random.seed(seed)
    batch = DR.encode_batch(_rooms(idx, max_len), DR.synth_stats(), max_len, **_args(enc, augs))
    assert np.array_equal(batch["length"], g[name + ".length"])
    assert np.array_equal(batch["class_labels"], g[name + ".class_labels"])
    for k in KEYS[1:]:
        assert batch[k].dtype == np.float32 and batch[k].shape == g[name + "." + k].shape
        assert np.abs(batch[k] - g[name + "." + k]).max() <= 1e-7, k


# ------------------------------------------------------------------------------------------------ host logic (CPU)
@pytest.fixture(scope="module")
def cached_dirs(tmp_path_factory):
    out = {}
    for max_len in (12, 21):
        root = str(tmp_path_factory.mktemp("cached_%d" % max_len))
        ids = DR.write_synth_cached_dataset(root, N_SCENES, seed=0, max_length=max_len)
        out[max_len] = (root, ids)
    return out


def _encoded(cached_dirs, enc, augs, max_len):
    from diffuscene_amd.datasets import CachedThreedFront, dataset_encoding_factory
    root, ids = cached_dirs[max_len]
    cfg = {"train_stats": "dataset_stats.txt", "room_layout_size": "64,64", "max_length": max_len}
    raw = CachedThreedFront(root, config=cfg, scene_ids=set(ids))
    return raw, dataset_encoding_factory(enc, raw, augs, None)


def test_cached_store_loads_reference_format(cached_dirs):
    raw, enc = _encoded(cached_dirs, BATCHES[0][1], BATCHES[0][2], 12)
    assert len(raw) == N_SCENES and raw.n_classes == DR.N_OBJECT_TYPES + 2 and raw.max_length == 12
    for i in (0, 5, N_SCENES - 1):
        ref = DR.synth_scene(i, 0, max_length=12)
        got = raw.get_room_params(i)
        for k in ("class_labels", "translations", "sizes", "angles", "objfeats_32"):
            assert np.array_equal(got[k], ref[k]), k
        assert got["room_layout"].shape == (1, 64, 64) and got["room_layout"].dtype == np.float32
    assert enc.feature_size == 7 + raw.n_classes and enc.max_length == 12
    # only a subset of scene ids
    from diffuscene_amd.datasets import CachedThreedFront
    sub = CachedThreedFront(cached_dirs[12][0], config=raw.config, scene_ids={"SCENE00003", "SCENE00007"})
    assert len(sub) == 2 and np.array_equal(sub.get_room_params(1)["sizes"], DR.synth_scene(7, 0, max_length=12)["sizes"])


@pytest.mark.parametrize("case", BATCHES, ids=[b[0] for b in BATCHES])
def test_host_draws_consume_rng_like_reference(cached_dirs, case):
    name, enc, augs, seed, idx, max_len = case
    _, ds = _encoded(cached_dirs, enc, augs, max_len)
    np.random.seed(seed)
    rot, jit, order = ds.draw(idx)
    after_mine = np.random.rand()
    np.random.seed(seed)
    DR.encode_batch(_rooms(idx, max_len), DR.synth_stats(), max_len, **_args(enc, augs))
    assert after_mine == np.random.rand()
    assert (order is None) == ("no_prm" in enc)
    if order is not None:
        for b, i in enumerate(idx):
            L = DR.synth_scene(i, 0, max_length=max_len)["class_labels"].shape[0]
            assert sorted(order[b, :L].tolist()) == list(range(L))


def test_loader_order_matches_torch_dataloader(cached_dirs):
    from torch.utils.data import DataLoader
    _, ds = _encoded(cached_dirs, BATCHES[0][1], BATCHES[0][2], 12)
    for shuffle in (True, False):
        torch.manual_seed(123)
        want = [b.tolist() for b in DataLoader(list(range(N_SCENES)), batch_size=5, shuffle=shuffle)]
        torch.manual_seed(123)
        got = ds.loader(5, shuffle=shuffle).index_batches()
        assert got == want
    # two ranks split the common order batch-wise, no overlap
    torch.manual_seed(9)
    full = ds.loader(4, shuffle=True).index_batches()
    parts = []
    for r in range(2):
        torch.manual_seed(9)
        parts.append(ds.loader(4, shuffle=True, rank=r, world_size=2).index_batches())
    assert parts[0] == full[0::2] and parts[1] == full[1::2] and len(ds.loader(4, rank=0, world_size=2)) == 3


def test_post_process_matches_reference(golden_dir, cached_dirs):
    g = np.load(os.path.join(golden_dir, "dataset.npz"))
    _, ds = _encoded(cached_dirs, BATCHES[0][1], BATCHES[0][2], 12)
    name = BATCHES[0][0]
    post = ds.post_process({k: g["%s.%s" % (name, k)] for k in KEYS})
    for k in KEYS:
        assert np.allclose(post[k], g["post." + k], rtol=0, atol=1e-12), k


def test_unsupported_encodings_raise(cached_dirs):
    from diffuscene_amd.datasets import dataset_encoding_factory
    raw, _ = _encoded(cached_dirs, BATCHES[0][1], None, 12)
    for bad in ("cached_autoregressive_wocm", "cached_diffusion_text_cosin_angle_objfeatsnorm_lat32_wocm", "cached_diffusion_wocm"):
        with pytest.raises(NotImplementedError):
            dataset_encoding_factory(bad, raw, None, None)
    with pytest.raises(NotImplementedError):
        dataset_encoding_factory(BATCHES[0][1], raw, ["jitter", "rotations"], None)
    _, ds = _encoded(cached_dirs, BATCHES[0][1], None, 12)
    with pytest.raises(RuntimeError):
        ds.encode([0], device="cpu")


# ------------------------------------------------------------------------------------------------ HIP encoder (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize("case", BATCHES, ids=[b[0] for b in BATCHES])
def test_device_batches_match_reference_golden(golden_dir, cached_dirs, case):
    name, enc, augs, seed, idx, max_len = case
    g = np.load(os.path.join(golden_dir, "dataset.npz"))
    _, ds = _encoded(cached_dirs, enc, augs, max_len)
    np.random.seed(seed)
    s = ds.encode(idx, device="cuda:0")
    assert np.array_equal(s["length"].cpu().numpy(), g[name + ".length"])
    assert np.array_equal(s["class_labels"].cpu().numpy(), g[name + ".class_labels"])
    for k in KEYS[1:]:
        got = s[k].cpu().numpy()
        assert got.dtype == np.float32 and got.shape == g[name + "." + k].shape
        err = np.abs(got - g[name + "." + k]).max()
        print(name, k, "max abs err vs reference:", err)
        assert err <= ATOL, k
    # packed tensor = the denoiser's channel order (diffusion_scene_layout_ddpm.py:148-154)
    want = np.concatenate([g[name + "." + k] for k in ("translations", "sizes", "angles", "class_labels", "objfeats_32")], -1)
    assert np.abs(s["_packed"].cpu().numpy() - want).max() <= ATOL
    # padded rows are exactly the end symbol
    L = int(s["length"][0])
    row = s["_packed"][0, L:].cpu().numpy()
    if row.size:
        nc = g[name + ".class_labels"].shape[-1]
        assert np.all(row[:, :8] == 0) and np.all(row[:, 8 + nc:] == 0)
        assert np.all(row[:, 8:8 + nc - 1] == -1) and np.all(row[:, 8 + nc - 1] == 1)


@pytest.mark.gpu
def test_device_loader_epoch_and_train_step(cached_dirs):
    """One epoch through the device loader feeds the training wrapper end to end (keys / shapes of train_diffusion.py)."""
    _, ds = _encoded(cached_dirs, BATCHES[0][1], BATCHES[0][2], 12)
    torch.manual_seed(1)
    np.random.seed(1)
    seen = 0
    for sample in ds.loader(10, shuffle=True, device="cuda:0"):
        assert sample["class_labels"].shape[1:] == (12, 22) and sample["objfeats_32"].shape[1:] == (12, 32)
        assert sample["translations"].is_cuda and float(sample["translations"].abs().max()) <= 1.0
        seen += sample["length"].numel()
    assert seen == N_SCENES
    one = ds[3]
    ref_len = DR.synth_scene(3, 0, max_length=12)["class_labels"].shape[0]
    assert one["length"] == ref_len and one["sizes"].shape == (12, 3)
