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

    def spy(groups, tile_n, target, tile_k=128):
        r = real(groups, tile_n, target, tile_k)
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
    assert chosen and all(c == (135, (1, 1536)) for c in chosen), chosen      # (the block map asks the same rule for the reference length)
    assert plan.n_adds == 8              # only the 8 skip-connection adds are left (round 3: 8 + one per LayerNorm input)
    # text (M = 1536) multiplies on the exact-f32 kernel: every transposed weight copy is read as f32, none may be skipped
    assert plan.n_transposes_skipped == 0


def test_block_map_puts_a_layers_tiles_on_one_xcd_and_covers_every_tile_once():
    """train_plan.tn_block_map: physical block b runs on XCD b % 8 (consecutive workgroup ids are dealt round-robin over the XCDs); the
    tiles of one long layer share its operand strips, so they must share one L2; short groups fill in behind, evenly."""
    from diffuscene_amd.train_plan import tn_block_map
    # the headline plan's launch: 59 layers of 8 tiles, 30 of 16, 3 of 4 over 20480 tokens; the packed time MLP (1216 tiles, 256 tokens)
    groups = [(20480, 512, 512)] * 59 + [(20480, 512, 1024)] * 30 + [(20480, 512, 256)] * 3 + [(256, 19456, 2048)] + [(80, 9216, 128)]
    _check_block_map(groups, 128)
    _check_block_map(groups, 256)             # round 6: 256 x 256 tiles


def _check_block_map(groups, tile_k):
    from diffuscene_amd.train_plan import tn_block_map
    bm = tn_block_map(groups, tile_k=tile_k)
    assert len(bm) % 8 == 0

    def tiles(n, k):
        return ((n + 255) // 256) * ((k + tile_k - 1) // tile_k)
    seen = {}
    for b, (g, t) in enumerate(bm):
        if g < 0:
            assert t < 0
            continue
        assert (g, t) not in seen and 0 <= t < tiles(groups[g][1], groups[g][2])
        seen[(g, t)] = b
    assert len(seen) == sum(tiles(n, k) for _, n, k in groups)
    for g, (m, n, k) in enumerate(groups):
        xcds = {seen[(g, t)] % 8 for t in range(tiles(n, k))}
        if m == 20480:
            assert len(xcds) == 1, "a long layer's tiles must sit on one XCD"
            pos = sorted(seen[(g, t)] // 8 for t in range(tiles(n, k)))
            assert pos == list(range(pos[0], pos[0] + len(pos))), "... back to back in that XCD's list"
        elif tiles(n, k) >= 64:
            assert len(xcds) >= 6                     # dealt to the emptiest XCDs, wherever the long layers left room
    # every XCD carries the same token work to within one long layer, and the long layers come first on each of them
    work = [0] * 8
    first_short = [None] * 8
    last_long = [0] * 8
    for b, (g, t) in enumerate(bm):
        if g < 0:
            continue
        x = b % 8
        work[x] += groups[g][0]
        if groups[g][0] == 20480:
            last_long[x] = b // 8
        elif first_short[x] is None:
            first_short[x] = b // 8
    assert max(work) - min(work) <= 16 * 20480
    assert all(fs is None or fs > ll for fs, ll in zip(first_short, last_long))
    idle = sum(1 for g, _ in bm if g < 0)
    assert idle < len(bm) // 8                          # padding (idle blocks return at once): XCDs with fewer long tiles hold more short ones


def test_block_map_small_launches():
    from diffuscene_amd.train_plan import tn_block_map
    assert tn_block_map([(1290, 512, 512)]) == [(0, t) if x == 0 else (-1, -1) for t in range(8) for x in range(8)]
    bm = tn_block_map([(64, 1024, 512), (1290, 32, 512)])            # 16 short tiles + 4 long: long first on XCD 0, short dealt around
    assert sorted(gt for gt in bm if gt[0] >= 0) == sorted([(0, t) for t in range(16)] + [(1, t) for t in range(4)])
    assert [bm[8 * i] for i in range(4)] == [(1, 0), (1, 1), (1, 2), (1, 3)]


def test_long_k_split_choice_and_hazard_check():
    """Two small host rules of the training plan: the slab count of the single split-K launch that replaced the chunked long-K input
    gradients, and the column-exact hazard check that keeps one half of a skip-connection pair buffer from flushing GEMMs that read the
    other half."""
    import torch
    from diffuscene_amd.train_plan import TrainPlan, long_k_splits
    assert long_k_splits(19456) == 38 and 19456 // 38 == 512            # packed time-MLP outputs: 38 slabs of 512 terms
    assert long_k_splits(9216) == 36 and 9216 // 36 == 256               # context-MLP outputs
    assert long_k_splits(4128) == 3                                      # 129 * 32: only 3 and 43 divide it; 43 slabs would be 96 terms
    assert long_k_splits(8224) == 0                                      # 257 * 32: prime factor, no admissible slab count -> chunked form
    buf = torch.zeros(12, 1024)
    left, right = buf[:, :512], buf[:, 512:]
    assert not TrainPlan._may_overlap(left, right) and not TrainPlan._may_overlap(right, left)
    assert TrainPlan._may_overlap(left, buf[:, 256:768]) and TrainPlan._may_overlap(left, left)
    assert TrainPlan._may_overlap(left, buf.view(-1)[:512].view(1, 512)[:, :])          # different row stride: assume the worst
