"""GPU: the static training plan (train_plan.py / train_step.py) -- the default train_on_batch path -- against the autograd
path over the same kernels, graph replay against eager launches, the data-parallel reducer on RCCL (world_size 1), the
derived-weight cache after raw-pointer optimizer updates, and the reference's training-script call sequence driven with the
reference's own YAML configs (tests/golden/reference_configs.json)."""
import contextlib
import copy
import io
import json
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import weights as W  # noqa: E402


def dev():
    return torch.device("cuda:0")


def _cfg(kind, N, stats):
    nc = 25 if kind == "arrange" else 22
    kw = dict({"uncond": W.UNCOND_BEDROOM, "text": W.TEXT_BEDROOM, "arrange": W.REARRANGE_LIVING}[kind])
    cfg = {"type": "diffusion_scene_layout_ddpm", "net_type": "unet1d", "point_dim": 8 + nc + 32, "latent_dim": 0,
           "room_mask_condition": False, "sample_num_points": N, "objectness_dim": 0, "objfeat_dim": 32,
           "class_dim": nc, "angle_dim": 2, "learnable_embedding": True, "instance_condition": True,
           "instance_emb_dim": 128,
           "diffusion_kwargs": dict(schedule_type="linear", beta_start=1e-4, beta_end=0.02, time_num=1000,
                                    loss_type="mse", model_mean_type="v", model_var_type="fixedsmall",
                                    loss_separate=True, loss_iou=True, train_stats_file=stats),
           "net_kwargs": kw}
    if kind == "text":
        cfg.update(text_condition=True, text_embed_dim=512, text_bert_cached=True)
    if kind == "arrange":
        cfg.update(room_arrange_condition=True, arrange_emb_dim=384)
    return cfg, nc


def _model(kind, N, tmp_path, seed=0):
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import DiffusionSceneLayout_DDPM
    stats = os.path.join(str(tmp_path), "dataset_stats.txt")
    with open(stats, "w") as f:
        json.dump(W.DATASET_STATS, f)
    cfg, nc = _cfg(kind, N, stats)
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        m = DiffusionSceneLayout_DDPM(nc + 1, None, cfg)
    return m.to(dev()), nc


def _sample(kind, B, N, nc, seed=1):
    x = W.synth_scene_batch(B, N, nc, 32, seed=seed).to(dev())
    s = {"translations": x[:, :, 0:3].contiguous(), "sizes": x[:, :, 3:6].contiguous(), "angles": x[:, :, 6:8].contiguous(),
         "class_labels": x[:, :, 8:8 + nc].contiguous(), "objfeats_32": x[:, :, 8 + nc:].contiguous(),
         "room_layout": torch.zeros(B, 1, 64, 64, device=dev())}
    if kind == "text":
        s["desc_bert"] = torch.randn(B, 7, 768, generator=torch.Generator().manual_seed(seed)).to(dev())
        s["description"] = ["x"] * B
    return s


def _relnorm(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("kind,B,N", [("uncond", 4, 12), ("uncond", 3, 80), ("text", 2, 12), ("arrange", 2, 21)])
def test_plan_step_matches_autograd_path(kind, B, N, tmp_path, monkeypatch):
    """Same kernels, two drivers: every parameter gradient (denoiser AND wrapper-level: positional embedding, fc_text_f,
    fc_arrange_condition), the loss and the logged terms of the plan step equal the torch.autograd step on identical draws."""
    from diffuscene_amd.train_step import loss_step, plan_supported
    monkeypatch.setenv("DSC_TRAIN_GRAPH", "0")
    m, nc = _model(kind, N, tmp_path)
    assert plan_supported(m)
    s = _sample(kind, B, N, nc)
    torch.manual_seed(7)
    loss_p, dict_p, ent = loss_step(m, s, backward=True)
    fl = m._dsc_flat
    g_plan = {n: fl.grad_view(p).clone() for n, p in m.named_parameters() if p.requires_grad}
    for p in m.parameters():
        p.grad = None
    torch.manual_seed(7)
    loss_a, dict_a = m.get_loss(s)
    loss_a.backward()
    assert abs(float(loss_p) - float(loss_a)) <= 1e-6 * max(1.0, abs(float(loss_a)))
    for k in dict_a:
        assert abs(float(dict_p[k]) - float(dict_a[k])) <= 2e-6 * max(1.0, abs(float(dict_a[k]))), k
    bad = []
    for n, p in m.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None, n
        e = _relnorm(g_plan[n], p.grad)
        if e > 2e-5:
            bad.append((n, e))
    assert not bad, bad[:8]
    print("%s: plan launches fwd %d bwd %d, explicit gradient adds %d" % (kind, len(ent["plan"].fwd), len(ent["plan"].bwd),
                                                                         ent["plan"].n_adds))


def test_graph_replay_equals_eager_plan(tmp_path, monkeypatch):
    from diffuscene_amd.train_step import loss_step
    m, nc = _model("uncond", 21, tmp_path)
    s = _sample("uncond", 6, 21, nc)
    monkeypatch.setenv("DSC_TRAIN_GRAPH", "0")
    torch.manual_seed(3)
    loss_e, _, _ = loss_step(m, s, backward=True)
    g_eager = m._dsc_flat.G.clone()
    monkeypatch.setenv("DSC_TRAIN_GRAPH", "1")
    for i in range(3):                         # warm-up launch, capture + replay, replay
        for p in m._dsc_flat.params:           # (the alignment gaps of G are never written and stay 0)
            m._dsc_flat.grad_view(p).fill_(123.0)
        torch.manual_seed(3)
        loss_g, _, ent = loss_step(m, s, backward=True)
        assert float(loss_g) == float(loss_e)
        assert torch.equal(m._dsc_flat.G, g_eager), i
    assert ent["graph"] is not None


def test_train_on_batch_plan_equals_autograd_training(tmp_path, monkeypatch):
    """Three full train_on_batch steps (plan + graph + FusedAdam) vs the same steps on the autograd path: parameters stay
    together; then the ADVICE r1 scenario: no_grad forwards between training steps must see the UPDATED weights."""
    from diffuscene_amd.networks import optimizer_factory
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import train_on_batch, validate_on_batch
    tcfg = {"training": {"max_grad_norm": 10}}
    ma, nc = _model("uncond", 12, tmp_path, seed=0)
    mb, _ = _model("uncond", 12, tmp_path, seed=0)
    mb.load_state_dict(ma.state_dict())
    oa = optimizer_factory({"optimizer": "Adam", "lr": 2e-4}, filter(lambda p: p.requires_grad, ma.parameters()))
    ob = optimizer_factory({"optimizer": "Adam", "lr": 2e-4}, filter(lambda p: p.requires_grad, mb.parameters()))
    s = _sample("uncond", 8, 12, nc)
    x = torch.cat([s["translations"], s["sizes"], s["angles"], s["class_labels"], s["objfeats_32"]], -1)
    t = torch.tensor([5, 100, 300, 500, 700, 900, 950, 999], device=dev())

    def fresh_forward(model):
        net = Unet1D(**dict(W.UNCOND_BEDROOM))
        net.load_state_dict({k[len("diffusion.model."):]: v for k, v in model.state_dict().items()
                             if k.startswith("diffusion.model.")})
        net.to(dev())
        with torch.no_grad():
            return net(x, t, model.positional_embedding.detach()[None].expand(8, -1, -1), None)

    for rnd_ in range(2):
        for i in range(2):
            torch.manual_seed(11 + i)
            la = train_on_batch(ma, oa, s, tcfg)
            monkeypatch.setenv("DSC_TRAIN_PLAN", "0")
            torch.manual_seed(11 + i)
            lb = train_on_batch(mb, ob, s, tcfg)
            monkeypatch.delenv("DSC_TRAIN_PLAN")
            assert abs(la - lb) <= 1e-5 * max(1.0, abs(lb)), (rnd_, i, la, lb)
        # the two paths sum some products in different fp32 orders (the plan's split-K time-MLP GEMMs, grouped reductions), and
        # Adam's g / sqrt(v) normalisation passes relative gradient differences straight into the parameters.  Conv biases in
        # front of a GroupNorm have a true gradient of ZERO (the norm removes the mean): their computed gradient is rounding noise
        # that Adam turns into +-lr steps, so 1-D tensors are only held to the size of such a random walk.
        for (name, p), q in zip(ma.named_parameters(), mb.parameters()):
            tol = 1e-4 if p.dim() >= 2 else 2e-3
            assert _relnorm(p, q) < tol, (name, _relnorm(p, q))
        with torch.no_grad():                         # engine path with derived weights cached across optimizer steps
            out = ma.diffusion.model(x, t, ma._instance_condition(8, dev()), None)
        assert _relnorm(out, fresh_forward(ma)) < 1e-6, "stale derived weights after FusedAdam steps"
        torch.manual_seed(5)
        va = validate_on_batch(ma, s, tcfg)
        monkeypatch.setenv("DSC_TRAIN_PLAN", "0")
        torch.manual_seed(5)
        vb = validate_on_batch(mb, s, tcfg)
        monkeypatch.delenv("DSC_TRAIN_PLAN")
        assert abs(va - vb) <= 1e-5 * max(1.0, abs(vb))


@pytest.mark.parametrize("flush", ["thirds", "block", "single", "end"])
def test_ddp_reducer_on_rccl_world1_matches_single_process(tmp_path, monkeypatch, flush):
    """The data-parallel path on the real backend: torch.distributed 'nccl' (= RCCL) with world_size 1 and the reducer forced
    on.  Buckets must be launched from the backward's progress callback, and gradients / updated parameters must equal the
    single-process graph path (sum over one rank, 1/world = 1) up to the summation order of the weight-gradient GEMMs,
    which the data-parallel plan flushes in several groups instead of one."""
    import torch.distributed as dist
    from diffuscene_amd.networks import optimizer_factory
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import train_on_batch
    tcfg = {"training": {"max_grad_norm": 10}}
    ma, nc = _model("uncond", 12, tmp_path, seed=0)
    mb, _ = _model("uncond", 12, tmp_path, seed=0)
    mb.load_state_dict(ma.state_dict())
    s = _sample("uncond", 8, 12, nc)
    oa = optimizer_factory({"optimizer": "Adam", "lr": 2e-4}, filter(lambda p: p.requires_grad, ma.parameters()))
    ob = optimizer_factory({"optimizer": "Adam", "lr": 2e-4}, filter(lambda p: p.requires_grad, mb.parameters()))
    for i in range(2):
        torch.manual_seed(21 + i)
        train_on_batch(ma, oa, s, tcfg)
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev())
    try:
        monkeypatch.setenv("DSC_DDP_FORCE", "1")
        monkeypatch.setenv("DSC_DDP_FLUSH", flush)
        for i in range(2):                      # step 1 eager (progress callback per launch), step 2 = replay of the hipGraph segments
            torch.manual_seed(21 + i)
            train_on_batch(mb, ob, s, tcfg)
        ent = next(iter(mb._dsc_plan_runner.plans.values()))
        red = ent["reducer"]
        assert red is not None and len(red.buckets) >= 8
        assert red.launched_during_backward >= (5 if flush in ("thirds", "block") else 1)
        sg = ent["graph"]
        assert sg is not None and len(sg.segments) == len(sg.graphs) >= (5 if flush == "block" else 2), "the DDP step must be captured"
        assert _relnorm(ma._dsc_flat.G, mb._dsc_flat.G) < 2e-6
        assert _relnorm(ma._dsc_flat.P, mb._dsc_flat.P) < 2e-6
    finally:
        dist.destroy_process_group()


def test_reference_training_script_sequence_with_reference_yaml(golden_dir, tmp_path):
    """scripts/train_diffusion.py:179-252 call for call -- build_network(config from the reference's YAML), optimizer_factory,
    schedule_factory, adjust_learning_rate, train_on_batch per batch, validate_on_batch, checkpoint round trip -- on batches
    from the device input pipeline over a synthetic cached dataset."""
    from oracle import dataset_ref as DR
    from diffuscene_amd.datasets import CachedThreedFront, dataset_encoding_factory
    from diffuscene_amd.networks import adjust_learning_rate, build_network, optimizer_factory, schedule_factory
    from diffuscene_amd.stats_logger import StatsLogger
    cfgs = json.load(open(os.path.join(golden_dir, "reference_configs.json")))
    config = copy.deepcopy(cfgs["uncond/diffusion_bedrooms_instancond_lat32_v.yaml"])
    root = str(tmp_path / "cached")
    os.makedirs(root)
    ids = DR.write_synth_cached_dataset(root, 40, seed=0, max_length=12)
    # the two site-specific paths of the YAML (a cluster directory) are the only values a user edits
    config["data"]["dataset_directory"] = root
    config["network"]["diffusion_kwargs"]["train_stats_file"] = os.path.join(root, "dataset_stats.txt")
    dcfg = {"train_stats": "dataset_stats.txt", "room_layout_size": "64,64", "max_length": 12}
    raw = CachedThreedFront(root, config=dcfg, scene_ids=set(ids))
    ds = dataset_encoding_factory(config["data"]["encoding_type"], raw, config["data"].get("augmentations"), None)
    assert config["network"]["sample_num_points"] == ds.max_length == 12
    # the synthetic store has its own number of object types: class_dim = object types + 'end' (the YAML's 22 = 21 + 1)
    nc = ds.n_classes - 1
    config["network"]["class_dim"] = nc
    config["network"]["point_dim"] = 8 + nc + 32
    config["network"]["net_kwargs"]["class_dim"] = nc
    config["network"]["net_kwargs"]["channels"] = 8 + nc + 32
    with contextlib.redirect_stdout(io.StringIO()):
        network, train_on_batch, validate_on_batch = build_network(ds.feature_size, ds.n_classes, config, None, device=dev())
        optimizer = optimizer_factory(config["training"], filter(lambda p: p.requires_grad, network.parameters()))
        lr_scheduler = schedule_factory(config["training"])
    torch.manual_seed(0)
    np.random.seed(0)
    losses = []
    for epoch in range(2):
        adjust_learning_rate(lr_scheduler, optimizer, epoch)
        network.train()
        for b, sample in enumerate(ds.loader(16, shuffle=True, device=dev())):
            for k, v in sample.items():
                if not isinstance(v, list):
                    sample[k] = v.to(dev())
            losses.append(train_on_batch(network, optimizer, sample, config))
            StatsLogger.instance().print_progress(epoch + 1, b + 1, losses[-1])
        StatsLogger.instance().clear()
    assert len(losses) == 6 and all(np.isfinite(losses))
    assert optimizer.param_groups[0]["lr"] == config["training"]["lr"]
    network.eval()
    for b, sample in enumerate(ds.loader(8, shuffle=False, device=dev())):
        v = validate_on_batch(network, sample, config)
        assert np.isfinite(v)
        break
    # checkpoint round trip in the reference's format (utils.save_checkpoints: torch.save of the two state_dicts)
    torch.save(network.state_dict(), str(tmp_path / "model_00000"))
    torch.save(optimizer.state_dict(), str(tmp_path / "opt_00000"))
    with contextlib.redirect_stdout(io.StringIO()):
        net2, _, _ = build_network(ds.feature_size, ds.n_classes, config, str(tmp_path / "model_00000"), device=dev())
    for (k1, v1), (k2, v2) in zip(network.state_dict().items(), net2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_alternating_text_lengths_keep_their_plans_and_clean_gradients(tmp_path, monkeypatch):
    """Text batches change the token count L from batch to batch: the plans of recently used signatures must be kept (LRU, not
    first-in-first-out), their graphs replayed, and switching signature must not leave the previous signature's gradients behind
    for Adam (stale gradients in G)."""
    import copy as _copy
    from diffuscene_amd.networks import optimizer_factory
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import train_on_batch
    monkeypatch.setenv("DSC_PLAN_CACHE_MAX", "2")
    m, nc = _model("text", 12, tmp_path)
    opt = optimizer_factory({"optimizer": "Adam", "lr": 1e-4}, filter(lambda p: p.requires_grad, m.parameters()))
    tcfg = {"training": {"max_grad_norm": 10}}
    batches = {}
    for L in (5, 7, 9):
        s = _sample("text", 4, 12, nc, seed=L)
        s["desc_bert"] = torch.randn(4, L, 768, generator=torch.Generator().manual_seed(L)).to(dev())
        batches[L] = s
    order = [5, 7, 5, 7, 5, 9, 5]
    for i, L in enumerate(order):
        torch.manual_seed(100 + i)
        assert np.isfinite(train_on_batch(m, opt, batches[L], tcfg))
    r = m._dsc_plan_runner
    keys = list(r.plans)
    assert len(keys) == 2 and [k[4] for k in keys] == [9, 5], keys          # L=7 (least recently used) went, 9 and 5 stay, 5 is newest
    assert r.plans[keys[1]]["graph"] is not None                             # the hot signature runs from its captured graph
    # a signature switch must not leave the previous signature's gradients behind in G
    G = m._dsc_flat.G
    assert float(G.abs().max()) > 0
    r.flat.G.fill_(1.0)                                                     # poison, then force a signature switch
    r.last_key = ("other",)
    torch.manual_seed(1)
    train_on_batch(m, opt, batches[5], tcfg)
    covered = torch.zeros_like(G, dtype=torch.bool)
    for p in m._dsc_flat.params:
        o, n = m._dsc_flat.grad_range(p)
        covered[o:o + n] = True
    assert float(G[~covered].abs().max()) == 0.0                            # alignment gaps were re-zeroed with the switch
    assert bool(torch.isfinite(G).all()) and float((G[covered] == 1.0).float().mean()) < 1e-3   # every gradient was rewritten
    assert _copy.deepcopy(m) is not None                                     # a copied model (EMA) drops the plan caches and still copies
