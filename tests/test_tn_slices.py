"""Host logic of the grouped weight-gradient launch: how many token slices (train_plan.tn_token_slices).  The rule was found wrong on the
text configuration by a per-launch profile (profiles/r03_plan_profile_train_text.txt: splits = 32 on every launch, 5.0 of 12.2 ms) -- the
longest group of a launch (k / v projections over B x L text tokens) decided for the 850 bulk tiles over B x N object tokens."""
from diffuscene_amd.train_plan import tn_token_slices


def _old_rule(groups, tile_n, target):
    m_max = max(m for m, _, _ in groups)
    long_tiles = sum(((n + tile_n - 1) // tile_n) * ((k + 127) // 128) for m, n, k in groups if 2 * m >= m_max)
    return max(1, min(32, -(-target // max(long_tiles, 1))))


def test_uniform_length_launches_choose_what_they_chose_before():
    # the metric configuration and the B = 128 ones: every layer over the same B x N tokens, plus the per-scene time-MLP gradient
    for m in (20480, 10240, 5376):
        for layers in (3, 7, 44, 57):
            groups = [(m, 512, 512)] * layers + [(m, 1024, 512)] * (layers // 4) + [(m // 80, 19456, 2048)]
            for tile_n, target in ((256, 768), (128, 4096)):
                got, m_ref = tn_token_slices(groups, tile_n, target)
                assert m_ref == m
                assert got == min(_old_rule(groups, tile_n, target), m // 256), (m, layers, tile_n)
    assert tn_token_slices([(20480, 512, 512)] * 50, 256, 768) == (2, 20480)


def test_text_launch_is_sliced_by_its_bulk_not_by_its_longest_group():
    bulk = [(1536, 512, 512)] * 100 + [(1536, 1024, 512)] * 30            # B x N = 128 x 12 object tokens
    text_side = [(4096, 256, 512)] * 9                                     # k / v projections over B x L = 128 x 32 text tokens
    per_scene = [(128, 19456, 2048)]                                       # packed time-MLP gradient over B rows
    groups = text_side + bulk + per_scene
    assert _old_rule(groups, 256, 768) == 22                               # 70-token slices (the real plan: 32 slices of 48)
    assert tn_token_slices(groups, 256, 768) == (1, 1536)                  # 1316 bulk tiles fill the chip 5 times un-sliced
    assert tn_token_slices(groups, 128, 4096) == (2, 1536)


def test_a_slice_is_at_least_256_tokens_of_the_bulk():
    assert tn_token_slices([(1536, 512, 512)] * 5, 256, 768) == (6, 1536)   # 40 tiles would ask for 20 slices of 77 tokens
    assert tn_token_slices([(160, 512, 512)] * 5, 256, 768) == (1, 160)     # tiny batches are never sliced
    assert tn_token_slices([(20480, 512, 512)] * 2, 256, 768)[0] == 32      # the cap of the slab layout


def test_text_training_plan_builds_on_the_host_with_short_slice_counts(tmp_path, monkeypatch):
    """The whole training plan of BASELINE configs[3] (B=128, N=12, L=32) is BUILT on the host -- argument structs, device tables, the
    launch list; nothing is launched, only the HIP-device check of the table-building code is bypassed.  Round 3: its three grouped
    weight-gradient launches were cut into 2 / 6 / 3 token slices (32 each before the slice-rule fix).  Round 4: nothing forces an
    early flush of the pending weight-gradient GEMMs any more (out-of-place LayerNorm accumulation, column-exact hazard check), so
    the single-GPU plan has ONE grouped launch of all 135 layers -- 1042 bulk tiles, un-sliced, no slabs, no reduction launch."""
    import contextlib
    import io
    import json
    import torch
    from oracle import weights as W
    from diffuscene_amd import ops, train_plan
    from diffuscene_amd._lib import SS_PER_SLOT
    from diffuscene_amd.flat import FlatStorage
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    monkeypatch.setattr(ops, "_dev", lambda t, name=None: t)
    chosen = []
    real = train_plan.tn_token_slices

    def spy(groups, tile_n, target):
        r = real(groups, tile_n, target)
        chosen.append((len(groups), r))
        return r
    monkeypatch.setattr(train_plan, "tn_token_slices", spy)
    stats = tmp_path / "dataset_stats.txt"
    stats.write_text(json.dumps(W.DATASET_STATS))
    kw = dict(W.TEXT_BEDROOM)
    with contextlib.redirect_stdout(io.StringIO()):
        net = Unet1D(**kw)
        d = DiffusionPoint(net, dict(objectness_dim=0, class_dim=22, angle_dim=2, objfeat_dim=32), time_num=1000, model_mean_type="v",
                           loss_separate=True, loss_iou=True, train_stats_file=str(stats)).diffusion
    flat = FlatStorage(net)
    tb = {n: getattr(d, n).float() for n in d._TABLE_NAMES}
    plan = train_plan.TrainPlan(net, flat, d, 128, 12, SS_PER_SLOT, 128, 32, 512, train_plan.HipBackend(torch.device("cpu")), tables=tb)
    assert len(plan.fwd) > 150 and len(plan.bwd) > 250
    assert chosen == [(135, (1, 1536))], chosen
    assert plan.n_adds == 8              # only the 8 skip-connection adds are left (round 3: 8 + one per LayerNorm input)
