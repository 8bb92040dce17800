"""The reference's OWN training script -- /root/reference/scripts/train_diffusion.py, unchanged -- executed against this package on the
CPU of the build container (missing item 6).  ``scene_synthesis.networks`` and ``.stats_logger`` resolve to diffuscene_amd
through the documented switch (diffuscene_amd.compat.install_as_scene_synthesis); ``scene_synthesis.datasets`` is the REFERENCE's own
package (its rendering / text / wandb imports stubbed, oracle/ref_loader.prepare_reference_script_imports) reading a synthetic cached
3D-FRONT directory; the config is the reference's shipped YAML with only the site-specific paths edited.

There is no GPU here and the product has no CPU path, so the two step functions ``build_network`` hands to the script are replaced by
recorders (what the script passes them is checked against what the HIP step consumes: keys, shapes, dtypes, the target / condition the
wrapper assembles from the batch); everything else the script touches is the product: argument parsing -> ``build_network`` signature and
return value -> parameter counts -> ``optimizer_factory`` -> ``load_checkpoints`` -> ``schedule_factory`` -> ``StatsLogger`` ->
``adjust_learning_rate`` -> the epoch loop -> ``save_checkpoints`` -> validation -> resuming from the checkpoints it wrote.
The same sequence with the real step functions runs on the GPU in tests/test_gpu_plan.py (call for call, device input pipeline).
Skipped where /root/reference does not exist (the GPU box)."""
import copy
import json
import os
import sys

import pytest
import torch
import yaml

from oracle import dataset_ref as DR
from oracle.ref_loader import prepare_reference_script_imports, reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="the reference tree is only present in the build container")


@pytest.fixture
def script_imports():
    """Stubs and aliases only live for this test: later tests (transformers probing for wandb, the reference loaders of the golden
    generators) must see the interpreter as they would without it."""
    before = set(sys.modules)
    path = list(sys.path)
    made = prepare_reference_script_imports()
    yield made
    for name in set(sys.modules) - before:
        if name.split(".")[0] in {m.split(".")[0] for m in made} or name.startswith(("scene_synthesis", "train_diffusion", "training_utils")):
            sys.modules.pop(name, None)
    sys.path[:] = path


def test_real_train_diffusion_script_drives_the_package(tmp_path, golden_dir, monkeypatch, script_imports):
    import scene_synthesis.networks as nets
    import diffuscene_amd.networks as ours
    assert nets is ours
    root = str(tmp_path / "cached")
    ids = DR.write_synth_cached_dataset(root, 40, seed=0, max_length=12)
    with open(tmp_path / "splits.csv", "w") as f:
        for i, sid in enumerate(ids):
            f.write("%s,%s\n" % (sid, "test" if i % 5 == 4 else "train"))
    cfgs = json.load(open(os.path.join(golden_dir, "reference_configs.json")))
    config = copy.deepcopy(cfgs["uncond/diffusion_bedrooms_instancond_lat32_v.yaml"])
    config["data"].update(dataset_directory=root, annotation_file=str(tmp_path / "splits.csv"), filter_fn="no_filtering")
    config["network"]["diffusion_kwargs"]["train_stats_file"] = os.path.join(root, "dataset_stats.txt")
    nc = DR.N_OBJECT_TYPES + 1                         # the synthetic store's object types + 'end' (the YAML's 22 = 21 + 1)
    assert config["network"]["class_dim"] == nc        # the bedroom YAML as shipped
    config["training"].update(epochs=2, batch_size=16, save_frequency=1)
    config["validation"].update(frequency=1, batch_size=8)
    cfg_path = tmp_path / "config.yaml"
    cfg_path.write_text(yaml.safe_dump(config))

    calls = {"train": [], "val": [], "models": []}

    def rec_train(model, optimizer, sample_params, cfg):
        assert cfg["training"]["max_grad_norm"] == 10 and optimizer.param_groups[0]["lr"] == 0.0002
        target, condition, cross = model._loss_inputs(sample_params)           # the wrapper's batch assembly, on the script's batch
        B = sample_params["class_labels"].shape[0]
        assert tuple(target.shape) == (B, 12, 62) and target.dtype == torch.float32 and cross is None
        assert tuple(condition.shape) == (B, 12, 128) and condition.stride(0) == 0
        assert sample_params["class_labels"].shape[-1] == nc and sample_params["objfeats_32"].shape[-1] == 32
        assert float(target[:, :, 8:8 + nc].abs().max()) == 1.0              # class labels arrive as -1 / +1
        calls["train"].append(B)
        calls["models"].append((model, model.training))
        from scene_synthesis.stats_logger import StatsLogger
        StatsLogger.instance()["loss.bbox"].value = 0.5
        return 0.25

    def rec_val(model, sample_params, cfg):
        calls["val"].append(sample_params["class_labels"].shape[0])
        return 0.125
    monkeypatch.setattr(ours, "_train_on_batch", rec_train)
    monkeypatch.setattr(ours, "_validate_on_batch", rec_val)
    import train_diffusion
    out = tmp_path / "out"
    train_diffusion.main([str(cfg_path), str(out), "--experiment_tag", "exp", "--seed", "3"])
    exp = out / "exp"
    n_train = sum(1 for i in range(40) if i % 5 != 4)
    assert calls["train"] == [16, 16] * 2 and sum(calls["train"][:2]) == n_train      # 2 epochs x (32 scenes / 16)
    assert calls["val"] == [8]                                                          # epoch 1 validates (frequency 1, i > 0)
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import DiffusionSceneLayout_DDPM
    assert isinstance(calls["models"][0][0], DiffusionSceneLayout_DDPM) and all(tr for _, tr in calls["models"])
    for name in ("params.json", "bounds.npz", "stats.txt", "model_00000", "opt_00000", "model_00001", "opt_00001"):
        assert (exp / name).exists(), name
    stats = (exp / "stats.txt").read_text()
    assert "epoch: 1 - batch: 2 - loss: 0.25000 - loss.bbox: 0.50000" in stats and "epoch: -1 - batch: 1 - loss: 0.12500" in stats
    sd = torch.load(str(exp / "model_00001"))
    assert "positional_embedding" in sd and "diffusion.model.init_conv.weight" in sd
    # a second run in the same experiment directory resumes from the checkpoints the first one wrote (load_checkpoints:
    # model.load_state_dict + optimizer.load_state_dict with OUR optimizer's state_dict format) and has nothing left to do
    calls["train"].clear()
    train_diffusion.main([str(cfg_path), str(out), "--experiment_tag", "exp", "--seed", "3"])
    assert calls["train"] == []


def test_real_generate_diffusion_script_drives_the_package(tmp_path, golden_dir, monkeypatch, script_imports):
    """/root/reference/scripts/generate_diffusion.py, unchanged: argument parsing -> the reference's own test-split dataset ->
    ``build_network(feature_size, n_classes, config, weight_file, device)`` loading a checkpoint OUR model wrote -> ``network.eval()`` ->
    ``network.generate_layout(room_mask=, num_points=, point_dim=, text=, device=, clip_denoised=, batch_seeds=)`` (our wrapper's real
    method: only the reverse loop under it, ``sample``, is a recorder -- there is no GPU here) -> OUR post-filter dict through the REFERENCE's
    ``dataset.post_process`` -> the tensors the script hands to shape retrieval.  Rendering and mesh export (out of scope) are
    stubs; retrieval itself is tests/test_gpu_retrieval.py."""
    import scene_synthesis.networks as nets
    import diffuscene_amd.networks as ours
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import DiffusionSceneLayout_DDPM
    assert nets is ours
    root = str(tmp_path / "cached")
    ids = DR.write_synth_cached_dataset(root, 20, seed=1, max_length=12)
    with open(tmp_path / "splits.csv", "w") as f:
        for i, sid in enumerate(ids):
            f.write("%s,%s\n" % (sid, "test" if i % 4 == 3 else "train"))
    cfgs = json.load(open(os.path.join(golden_dir, "reference_configs.json")))
    config = copy.deepcopy(cfgs["uncond/diffusion_bedrooms_instancond_lat32_v.yaml"])
    config["data"].update(dataset_directory=root, annotation_file=str(tmp_path / "splits.csv"), filter_fn="no_filtering")
    config["network"]["diffusion_kwargs"]["train_stats_file"] = os.path.join(root, "dataset_stats.txt")
    nc = DR.N_OBJECT_TYPES + 1
    N, C = config["network"]["sample_num_points"], config["network"]["point_dim"]
    assert (N, C) == (12, 8 + nc + 32)
    cfg_path = tmp_path / "config.yaml"
    cfg_path.write_text(yaml.safe_dump(config))
    # a checkpoint written by OUR model, as scripts/train_diffusion.py saves it (state_dict through torch.save)
    torch.manual_seed(5)
    trained, _, _ = ours.build_network(None, nc, config, None, device="cpu")
    weights = tmp_path / "model_00010"
    torch.save(trained.state_dict(), str(weights))

    import generate_diffusion
    calls = {"sample": [], "retrieval": [], "onlysize": []}
    futures = ["the pickled 3D-FUTURE models"]
    monkeypatch.setattr(generate_diffusion.ThreedFutureDataset, "from_pickled_dataset", staticmethod(lambda path: futures))
    monkeypatch.setattr(generate_diffusion, "floor_plan_from_scene",
                        lambda scene, textures, no_texture=False: (["floor"], ["floor mesh"], torch.zeros(1, 1, 64, 64)))

    def rec_sample(self, room_mask, num_points, point_dim, batch_size=1, text=None, partial_boxes=None, input_boxes=None, ret_traj=False,
                   ddim=False, clip_denoised=False, freq=40, batch_seeds=None):
        # what the script's call reaches the reverse loop with
        assert isinstance(self, DiffusionSceneLayout_DDPM) and not self.training
        assert tuple(room_mask.shape) == (1, 1, 64, 64) and (num_points, point_dim, batch_size) == (N, C, 1)
        assert text is None and partial_boxes is None and input_boxes is None and clip_denoised is True and not ret_traj
        assert torch.equal(batch_seeds, torch.arange(len(calls["sample"]), len(calls["sample"]) + 1))
        sd = self.state_dict()
        assert all(torch.equal(sd[k], v) for k, v in trained.state_dict().items())           # the checkpoint really was loaded
        g = torch.Generator().manual_seed(len(calls["sample"]))
        x = torch.rand(1, N, C, generator=g) * 2 - 1
        x[0, 5:, 8 + nc - 1] = 1.0                     # slots 5.. are 'empty' (logit >= 0): the post-filter drops them
        x[0, :5, 8 + nc - 1] = -1.0
        calls["sample"].append(x.clone())
        return x
    monkeypatch.setattr(DiffusionSceneLayout_DDPM, "sample", rec_sample)

    def rec_retrieve(bbox_params_t, objects_dataset, classes, diffusion=False, no_texture=False, query_objfeats=None):
        assert objects_dataset is futures and diffusion and no_texture
        assert tuple(bbox_params_t.shape) == (1, 5, (nc - 1) + 3 + 3 + 1) and len(classes) == nc + 1     # 21 types + 'start' + 'end'
        assert query_objfeats is None or tuple(query_objfeats.shape) == (1, 5, 32)
        (calls["retrieval"] if query_objfeats is not None else calls["onlysize"]).append((bbox_params_t.copy(), query_objfeats))
        return [], [], []
    monkeypatch.setattr(generate_diffusion, "get_textured_objects_based_on_objfeats", rec_retrieve)
    monkeypatch.setattr(generate_diffusion, "get_textured_objects", rec_retrieve)

    out = tmp_path / "generated"
    generate_diffusion.main([str(cfg_path), str(out), str(tmp_path / "futures.pkl"), "--weight_file", str(weights), "--n_sequences", "2",
                             "--clip_denoised", "--retrive_objfeats", "--no_texture", "--fix_order", "--without_screen"])
    assert len(calls["sample"]) == 2 and len(calls["retrieval"]) == 2 and len(calls["onlysize"]) == 2
    # the numbers the script hands to retrieval are the reference's post_process of OUR post-filter's dict: classes as they are,
    # translations / sizes descaled by the dataset bounds, the angle back from (cos, sin), the latent codes descaled
    import numpy as np
    stats = json.load(open(os.path.join(root, "dataset_stats.txt")))
    for x, (boxes, feats) in zip(calls["sample"], calls["retrieval"]):
        kept = x[0, :5].numpy()
        assert np.array_equal(boxes[0, :, :nc - 1], kept[:, 8:8 + nc - 1])
        lo, hi = np.array(stats["bounds_translations"][:3], np.float32), np.array(stats["bounds_translations"][3:], np.float32)
        assert np.allclose(boxes[0, :, nc - 1:nc + 2], (kept[:, 0:3] + 1) / 2 * (hi - lo) + lo, rtol=1e-6, atol=1e-6)
        assert np.allclose(boxes[0, :, -1], np.arctan2(kept[:, 7], kept[:, 6]), rtol=1e-6, atol=1e-6)
        assert feats.shape == (1, 5, 32)


@pytest.mark.parametrize("mode", ["complete", "arrange"])
def test_real_completion_rearrange_script_drives_the_package(mode, tmp_path, golden_dir, monkeypatch, script_imports):
    """/root/reference/scripts/completion_rearrange.py, unchanged, in both of its modes: the reference's test-split dataset -> ``build_network``
    with a checkpoint our model wrote -> (re-arrangement) ``network.delete_empty_boxes`` on the noisy input dict -> ``network.complete_scene(
    room_mask=, num_points=, point_dim=, partial_boxes=, device=, clip_denoised=, batch_seeds=)`` / ``network.arrange_scene(..., input_boxes=, ...)``
    -- our wrapper's real methods over a recorder ``sample`` that checks the [translation | size | angle | class | objfeat] rows the script
    assembled from the dataset sample -> the reference's ``post_process`` of our dict -> the renderer's inputs.  Rendering is stubbed."""
    import numpy as np
    import diffuscene_amd.networks as ours
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import DiffusionSceneLayout_DDPM
    root = str(tmp_path / "cached")
    ids = DR.write_synth_cached_dataset(root, 20, seed=2, max_length=12)
    with open(tmp_path / "splits.csv", "w") as f:
        for i, sid in enumerate(ids):
            f.write("%s,%s\n" % (sid, "test" if i % 4 == 3 else "train"))
    cfgs = json.load(open(os.path.join(golden_dir, "reference_configs.json")))
    name = {"complete": "uncond/diffusion_bedrooms_instancond_lat32_v.yaml",
            "arrange": "rearrange/diffusion_bedrooms_instancond_lat32_v_rearrange.yaml"}[mode]
    config = copy.deepcopy(cfgs[name])
    config["data"].update(dataset_directory=root, annotation_file=str(tmp_path / "splits.csv"), filter_fn="no_filtering")
    config["network"]["diffusion_kwargs"]["train_stats_file"] = os.path.join(root, "dataset_stats.txt")
    config["validation"]["gen_gt"] = False
    nc = DR.N_OBJECT_TYPES + 1
    N, C = config["network"]["sample_num_points"], config["network"]["point_dim"]
    assert N == 12 and bool(config["network"].get("room_arrange_condition", False)) == (mode == "arrange")
    cfg_path = tmp_path / "config.yaml"
    cfg_path.write_text(yaml.safe_dump(config))
    torch.manual_seed(6)
    trained, _, _ = ours.build_network(None, nc, config, None, device="cpu")
    weights = tmp_path / "model_00020"
    torch.save(trained.state_dict(), str(weights))

    import completion_rearrange as script
    futures = ["the pickled 3D-FUTURE models"]
    monkeypatch.setattr(script.ThreedFutureDataset, "from_pickled_dataset", staticmethod(lambda path: futures))
    monkeypatch.setattr(script, "floor_plan_from_scene",
                        lambda scene, textures, no_texture=False: (["floor"], ["floor mesh"], torch.zeros(1, 1, 64, 64)))
    seen = {"sample": [], "folders": [], "final": []}
    P = 3

    def rec_sample(self, room_mask, num_points, point_dim, batch_size=1, text=None, partial_boxes=None, input_boxes=None, ret_traj=False,
                   ddim=False, clip_denoised=False, freq=40, batch_seeds=None):
        assert not self.training and (num_points, point_dim, batch_size) == (N, C, 1) and text is None and clip_denoised is True
        given = partial_boxes if mode == "complete" else input_boxes
        assert (partial_boxes is None) != (input_boxes is None) and given.dtype == torch.float32
        assert tuple(given.shape) == ((1, P, 8 + nc + 32) if mode == "complete" else (1, N, 8 + nc + 32))
        # the column layout the reverse loops expect: class scores (-1 / +1 from the dataset encoding) at 8 .. 8+nc, (cos, sin) at 6, 7
        assert float(given[0, :, 8:8 + nc].abs().max()) == 1.0 and float(given[0, :, 8:8 + nc].abs().min()) == 1.0
        if mode == "complete":
            assert torch.allclose(given[0, :, 6] ** 2 + given[0, :, 7] ** 2, torch.ones(P), atol=1e-5)
        seen["sample"].append(given.clone())
        width = C if mode == "complete" else 8 + nc + 32            # arrange_samples re-assembles the full rows at t == 0
        g = torch.Generator().manual_seed(7)
        x = torch.rand(1, N, width, generator=g) * 2 - 1
        x[0, :, 8 + nc - 1] = torch.tensor([-1.0] * 4 + [1.0] * (N - 4))
        return x
    monkeypatch.setattr(DiffusionSceneLayout_DDPM, "sample", rec_sample)

    def rec_folder(args, folder, tag, dataset, objects_dataset, tr_floor, floor_plan, scene, scene_top2down, bbox_params, add_start_end=False,
                   diffusion=False):
        assert objects_dataset is futures and diffusion and not add_start_end
        seen["folders"].append((folder, {k: tuple(v.shape) for k, v in bbox_params.items()}))
    monkeypatch.setattr(script, "render_to_folder", rec_folder)

    def rec_final(args, bbox_params, dataset, objects_dataset, classes, floor_plan, tr_floor, scene, scene_top2down, path_to_image, path_to_objs,
                  filename, network_objae=None, device=None, diffusion=False):
        seen["final"].append({k: tuple(v.shape) for k, v in bbox_params.items()})
    monkeypatch.setattr(script, "render_scene_from_bbox_params", rec_final)
    monkeypatch.setattr(script, "get_textured_objects", lambda *a, **k: ([], [], []))
    monkeypatch.setattr(script, "get_textured_objects_based_on_objfeats", lambda *a, **k: ([], [], []))

    np.random.seed(0)
    argv = [str(cfg_path), str(tmp_path / "out"), str(tmp_path / "futures.pkl"), "--weight_file", str(weights), "--n_sequences", "2",
            "--clip_denoised", "--no_texture", "--without_screen", "--num_partial", str(P)]
    script.main(argv + (["--arrange_objects"] if mode == "arrange" else []))
    assert len(seen["sample"]) == 2 and len(seen["final"]) == 2
    # the network's dict: 4 kept objects, the reference's keys (class scores without the 'empty' column)
    assert seen["final"][0] == {"class_labels": (1, 4, nc - 1), "translations": (1, 4, 3), "sizes": (1, 4, 3), "angles": (1, 4, 2),
                                "objfeats": (1, 4, 32)}
    if mode == "complete":
        assert [f for f, _ in seen["folders"]] == ["partial", "partial"]
        assert seen["folders"][0][1]["class_labels"] == (1, P, nc) and seen["folders"][0][1]["objfeats"] == (1, P, 32)
    else:
        # delete_empty_boxes ran on the noisy input: only the scene's real objects reach the renderer, classes without the 'empty' column
        assert [f for f, _ in seen["folders"]] == ["noisy", "noisy"]
        shapes = seen["folders"][0][1]
        k = shapes["class_labels"][1]
        assert 1 <= k <= N and shapes == {"class_labels": (1, k, nc - 1), "translations": (1, k, 3), "sizes": (1, k, 3), "angles": (1, k, 2),
                                          "objfeats": (1, k, 32)}
