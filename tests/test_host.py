"""CPU: host-side logic of the product package -- no kernel is ever executed here.
* the C-ABI library loads and exports every function include/diffuscene_hip.h declares;
* state_dict layout == reference layout (golden key list), checkpoints round-trip;
* schedule tables are bit-identical to the reference's (golden);
* the product refuses to compute on CPU tensors (no fallback);
* RNG draw order / noise_fn protocol of the loops, checked with a recording fake kernel backend."""
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import weights as W
from oracle.make_golden import CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from diffuscene_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "diffuscene_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|int64_t)\s+(dsc_\w+)\s*\(", hdr, flags=re.M))
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), "%s declared in the header but not exported" % name
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert lib.dsc_version() >= 100


@pytest.mark.parametrize("name", list(CASES))
def test_state_dict_layout_is_the_reference_layout(golden_dir, name):
    from diffuscene_amd.networks.denoise_net import Unet1D
    keys = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))[name]
    net = Unet1D(**CASES[name][0])
    assert {k: list(v.shape) for k, v in net.state_dict().items()} == {k: s for k, s in keys}
    sd = W.synth_state_dict(CASES[name][0])
    net.load_state_dict(sd, strict=True)                      # a reference checkpoint loads unchanged
    for k, v in net.state_dict().items():
        assert torch.equal(v, sd[k])


def test_schedule_tables_bit_identical_to_reference(golden_dir):
    from diffuscene_amd.networks.diffusion_ddpm import GaussianDiffusion, get_betas
    g = np.load(os.path.join(golden_dir, "schedule_v_T1000.npz"))
    d = GaussianDiffusion({}, get_betas("linear", 1e-4, 0.02, 1000), "mse", "v", "fixedsmall", True, False, None)
    for k in g.files:
        assert np.array_equal(getattr(d, k).numpy(), g[k]), k
    with pytest.raises(NotImplementedError):
        get_betas("cosine", 1e-4, 0.02, 10)


@pytest.mark.parametrize("sched", ["warm0.1", "warm0.2", "warm0.5"])
def test_warmup_schedules_bit_identical_to_reference(golden_dir, sched):
    """The warm-up beta schedules of get_betas (reference diffusion_ddpm.py:62-79) and every table GaussianDiffusion derives from
    them, against the REAL reference's (tests/golden/meantypes.npz, oracle/make_golden_meantypes.py): bit for bit."""
    from diffuscene_amd.networks.diffusion_ddpm import GaussianDiffusion, get_betas
    g = np.load(os.path.join(golden_dir, "meantypes.npz"))
    d = GaussianDiffusion({}, get_betas(sched, 1e-4, 0.02, 1000), "mse", "eps", "fixedsmall", True, False, None)
    keys = [k[len(sched) + 1:] for k in g.files if k.startswith(sched + ".")]
    assert len(keys) == 9
    for k in keys:
        assert np.array_equal(getattr(d, k).numpy(), g[sched + "." + k]), (sched, k)


def test_no_cpu_fallback():
    from diffuscene_amd import ops
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    net = Unet1D(**W.UNCOND_BEDROOM)
    x = torch.zeros(1, 12, 62)
    with pytest.raises(RuntimeError, match="HIP device"):
        net(x, torch.zeros(1, dtype=torch.int64), None, None)
    diff = DiffusionPoint(net, dict(objectness_dim=0, class_dim=22, angle_dim=2, objfeat_dim=32), model_mean_type="v")
    with pytest.raises(RuntimeError, match="HIP device"):
        diff.diffusion.q_sample(x, torch.zeros(1, dtype=torch.int64), torch.zeros_like(x))
    with pytest.raises(RuntimeError, match="HIP device"):
        ops.gemm(torch.zeros(4, 32), torch.zeros(8, 32))


def test_unsupported_layouts_fail_loudly():
    from diffuscene_amd.networks.denoise_net import Unet1D
    with pytest.raises(NotImplementedError):
        Unet1D(dim=256, dim_mults=(1, 2, 4, 8))


class _FakeOps:
    """Recording stand-in for diffuscene_amd.ops (TEST DOUBLE, lives in tests/ only)."""

    def __init__(self):
        self.calls = []

    def p_sample(self, x_t, model_out, noise, t, *a, **k):
        self.calls.append(("p_sample", int(t[0]), float(noise.flatten()[0])))
        return x_t

    def complete_overwrite(self, x, partial, noise, t, *a):
        self.calls.append(("overwrite", int(t[0]), float(noise.flatten()[0])))
        return x


def test_reverse_loop_draw_order(monkeypatch):
    """x_T first, then per step: (completion: partial noise BEFORE the model call) model call, one p_sample draw --
    also at t == 0 (diffusion_ddpm.py:345,364,461)."""
    from diffuscene_amd.networks import diffusion_ddpm as dd
    fake = _FakeOps()
    monkeypatch.setattr(dd, "ops", fake)
    d = dd.GaussianDiffusion({}, dd.get_betas("linear", 1e-4, 0.02, 3), "mse", "v", "fixedsmall", True, False, None)
    monkeypatch.setattr(d, "tables", lambda device: {n: getattr(d, n) for n in d._TABLE_NAMES})
    events = []
    counter = [0]

    def noise_fn(size=None, dtype=None, device=None):
        counter[0] += 1
        events.append(("draw", counter[0], tuple(size)))
        return torch.full(size, float(counter[0]))

    def denoise(x, t, c, cc):
        events.append(("model", int(t[0])))
        return x

    d.p_sample_loop(denoise, (2, 4, 5), "cpu", None, None, noise_fn=noise_fn)
    assert [e[0] for e in events] == ["draw", "model", "draw", "model", "draw", "model", "draw"]
    assert [c for c in fake.calls] == [("p_sample", 2, 2.0), ("p_sample", 1, 3.0), ("p_sample", 0, 4.0)]
    events.clear(); fake.calls.clear(); counter[0] = 0
    d.p_sample_loop_complete(denoise, (2, 4, 5), "cpu", None, None, noise_fn=noise_fn,
                             partial_boxes=torch.zeros(2, 1, 5))
    assert [e[0] for e in events] == ["draw"] + ["draw", "model", "draw"] * 3
    assert [c[0] for c in fake.calls] == ["overwrite", "p_sample"] * 3
    assert events[1][2] == (2, 1, 5) and events[3][2] == (2, 4, 5)


def test_scene_layout_wrapper_state_dict_and_postfilter():
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import DiffusionSceneLayout_DDPM
    cfg = {"type": "diffusion_scene_layout_ddpm", "net_type": "unet1d", "point_dim": 62, "latent_dim": 0,
           "room_mask_condition": False, "sample_num_points": 12, "objectness_dim": 0, "objfeat_dim": 32,
           "class_dim": 22, "angle_dim": 2, "learnable_embedding": True, "instance_condition": True,
           "instance_emb_dim": 128,
           "diffusion_kwargs": dict(schedule_type="linear", beta_start=1e-4, beta_end=0.02, time_num=1000,
                                    loss_type="mse", model_mean_type="v", model_var_type="fixedsmall",
                                    loss_separate=True, loss_iou=False, train_stats_file=None),
           "net_kwargs": dict(W.UNCOND_BEDROOM)}
    m = DiffusionSceneLayout_DDPM(23, None, cfg)
    keys = set(m.state_dict().keys())
    assert "positional_embedding" in keys
    assert {"diffusion.model." + k for k in W.unet1d_param_spec(**W.UNCOND_BEDROOM)} == keys - {"positional_embedding"}
    cond = m._instance_condition(5, torch.device("cpu"))
    assert cond.shape == (5, 12, 128) and cond.stride(0) == 0       # broadcast view, no (B,N,128) copy
    s = torch.zeros(2, 12, 62)
    s[:, :, 8 + 21] = 1.0          # all slots "empty" ...
    s[0, [1, 4], 8 + 21] = -1.0    # ... except slots 1 and 4 of batch row 0
    s[:, :, 0] = torch.arange(12.)[None]
    out = m.delete_empty_from_network_samples(s)
    assert out["translations"].shape == (2, 2, 3) and out["translations"][1, :, 0].tolist() == [1.0, 4.0]
    assert out["class_labels"].shape == (2, 2, 21) and out["objfeats"].shape == (2, 2, 32)
    assert m.delete_empty_from_network_samples(s, keep_empty=True)["sizes"].shape == (2, 12, 3)


def test_install_as_scene_synthesis():
    import sys
    import diffuscene_amd
    saved = {k: v for k, v in sys.modules.items() if k.startswith("scene_synthesis")}
    try:
        diffuscene_amd.install_as_scene_synthesis()
        from scene_synthesis.networks import build_network, optimizer_factory, schedule_factory, adjust_learning_rate  # noqa
        from scene_synthesis.stats_logger import StatsLogger, WandB  # noqa
        from scene_synthesis.networks.diffusion_ddpm import GaussianDiffusion  # noqa
        sched = schedule_factory({"schedule": "step", "lr": 2e-4, "lr_step": 10, "lr_decay": 0.5})
        assert sched.get_learning_rate(25) == 2e-4 * 0.25
    finally:
        for k in [k for k in sys.modules if k.startswith("scene_synthesis")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_fused_adam_is_a_torch_adam_with_the_same_checkpoint_format():
    """optimizer_factory returns FusedAdam: same param_groups keys / state_dict layout as torch.optim.Adam (opt_XXXXX
    checkpoints interchangeable); stepping CPU parameters is refused (no CPU fallback)."""
    import pytest
    from diffuscene_amd.networks import optimizer_factory
    from diffuscene_amd.optim import FusedAdam
    ps = [torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(5))]
    opt = optimizer_factory({"optimizer": "Adam", "lr": 2e-4}, ps)
    assert isinstance(opt, FusedAdam) and isinstance(opt, torch.optim.Adam)
    ref = torch.optim.Adam([torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(5))], lr=2e-4,
                           weight_decay=0.0)
    assert set(opt.param_groups[0].keys()) == set(ref.param_groups[0].keys())
    assert opt.param_groups[0]["lr"] == 2e-4 and opt.param_groups[0]["weight_decay"] == 0.0
    sd, rsd = opt.state_dict(), ref.state_dict()
    assert sd["state"] == {} and sd["param_groups"][0]["params"] == rsd["param_groups"][0]["params"]
    ref.load_state_dict(sd)
    opt.load_state_dict(rsd)
    ps[0].grad = torch.ones(3, 4)
    with pytest.raises(RuntimeError):
        opt.step()
    with pytest.raises(NotImplementedError):
        FusedAdam(ps, amsgrad=True)
    assert isinstance(optimizer_factory({"optimizer": "SGD"}, ps), torch.optim.SGD)


# ---------------------------------------------------------------------------------------------------------------------
# drop-in surface against the reference's own YAML configs (tests/golden/reference_configs.json, made by
# tests/golden/make_config_fixture.py from /root/reference/config/**) and the flat parameter storage
# ---------------------------------------------------------------------------------------------------------------------
def _ref_configs(golden_dir):
    import json
    import os
    return json.load(open(os.path.join(golden_dir, "reference_configs.json")))


def test_build_network_accepts_every_reference_yaml(golden_dir, tmp_path):
    import contextlib
    import copy
    import io
    import json
    from oracle import weights as W
    from diffuscene_amd.networks import build_network, optimizer_factory, schedule_factory, adjust_learning_rate
    stats = tmp_path / "dataset_stats.txt"
    stats.write_text(json.dumps(W.DATASET_STATS))
    cfgs = _ref_configs(golden_dir)
    assert len(cfgs) == 12
    for name, cfg in cfgs.items():
        config = copy.deepcopy(cfg)
        config["network"]["diffusion_kwargs"]["train_stats_file"] = str(stats)
        if config["network"].get("text_condition"):
            config["network"]["text_bert_cached"] = True          # no BERT weights offline: cached-feature input instead
        nc = config["network"]["class_dim"]
        with contextlib.redirect_stdout(io.StringIO()):
            net, train_fn, val_fn = build_network(8 + nc, nc + 1, config, None, device="cpu")
            opt = optimizer_factory(config["training"], filter(lambda p: p.requires_grad, net.parameters()))
            sched = schedule_factory(config["training"])
        adjust_learning_rate(sched, opt, 0)
        assert opt.param_groups[0]["lr"] == config["training"]["lr"], name
        n_params = sum(p.numel() for p in net.parameters())
        assert 74e6 < n_params < 81e6, (name, n_params)
        assert callable(train_fn) and callable(val_fn)
        keys = set(net.state_dict().keys())
        assert "positional_embedding" in keys and any(k.startswith("diffusion.model.downs.0.0.block1.proj") for k in keys)
        if "rearrange" in name:
            assert any(k.startswith("fc_arrange_condition.") for k in keys) and net.diffusion.model.channels == 5
        if "text" in name:
            assert "fc_text_f.weight" in keys and net.diffusion.model.text_condition


def test_unet1d_constructor_refusals():
    """what Unet1D accepts of the reference's keyword list (denoise_net.py:336-416) and what it refuses --
    loudly, at construction, never by approximating."""
    import contextlib
    import io
    from diffuscene_amd.networks.denoise_net import Unet1D
    ok = dict(dim=512, dim_mults=(1, 1, 1, 1), channels=62, seperate_all=True, objectness_dim=0, class_dim=22, angle_dim=2, objfeat_dim=32,
              context_dim=0, instanclass_dim=128)
    with contextlib.redirect_stdout(io.StringIO()):
        Unet1D(**dict(ok, dim_mults=(1, 1)))                                    # any number of levels
        Unet1D(**dict(ok, init_dim=512, out_dim=62))
        Unet1D(**dict(ok, seperate_all=False, channels=5, class_dim=0, objfeat_dim=0, angle_dim=2, size_dim=0))
        Unet1D(**dict(ok, text_condition=True, text_dim=256))
        Unet1D(**dict(ok, instanclass_dim=0))                                   # un-conditioned
        refused = [dict(dim=256), dict(dim_mults=(1, 2, 4, 8)), dict(dim_mults=(1, 1, 2)), dict(init_dim=256), dict(resnet_block_groups=4),
                   dict(learned_variance=True), dict(learned_sinusoidal_cond=True), dict(random_fourier_features=True),
                   dict(seperate_all=False, channels=65)]
        for kw in refused:
            with pytest.raises(NotImplementedError):
                Unet1D(**dict(ok, **kw))
        with pytest.raises(ValueError):
            Unet1D(**dict(ok, channels=63))                                     # channels must equal the attribute widths when separated
        with pytest.raises(NotImplementedError):
            Unet1D()                                                            # the reference's own defaults: dim=256, dim_mults=(1,2,4,8)


def test_flat_storage_keeps_module_semantics():
    import contextlib
    import io
    import torch
    from oracle import weights as W
    from diffuscene_amd.flat import FlatStorage, ensure_flat
    from diffuscene_amd.networks.denoise_net import Unet1D
    with contextlib.redirect_stdout(io.StringIO()):
        net = Unet1D(**dict(W.UNCOND_BEDROOM))
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    fs = FlatStorage(net)
    assert fs.valid() and ensure_flat(net) is fs
    for k, v in net.state_dict().items():
        assert torch.equal(v, sd0[k]), k                      # values and names unchanged
    n = sum(p.numel() for p in net.parameters())
    assert n <= fs.numel < n + 64 * len(fs.params)
    # packed conditioning weights are views: 19 time-MLP blocks x (1024, 2048), in block order
    tw, tgw = fs.packed["t_w"]
    t_blocks = [rb for rb, kind in net.resblocks_in_order() if kind == "t"]
    assert tw.shape == (19 * 1024, 2048) and tgw.shape == tw.shape
    for i, rb in enumerate(t_blocks):
        assert rb.mlp[1].weight.data_ptr() == tw[i * 1024].data_ptr()
        assert rb.mlp[1].weight.grad.data_ptr() == tgw[i * 1024].data_ptr()
    # in-place updates (optimizers, load_state_dict) keep the views; zero_grad(set_to_none) is undone by attach_grads
    net.load_state_dict(W.synth_state_dict(dict(W.UNCOND_BEDROOM)))
    assert fs.valid()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    opt.zero_grad()
    assert net.init_conv.weight.grad is None
    fs.attach_grads()
    fs.G.fill_(1.0)
    opt.step()
    assert fs.valid() and float(net.init_conv.bias.grad.sum()) == 512.0
    # buckets: contiguous cover of G, cut at parameter boundaries
    b = fs.buckets(8)
    assert b[0][0] == 0 and b[-1][1] == fs.numel and all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1)) and len(b) >= 8
    starts = {fs.offset[id(p)] for p in fs.params}
    assert all(s in starts for s, _ in b)
    # moving the module invalidates the storage and ensure_flat rebuilds it
    net.init_conv.weight.data = net.init_conv.weight.data.clone()
    assert not fs.valid() and ensure_flat(net) is not fs


def test_plans_own_the_storage_their_pointers_refer_to():
    """engine._own (round 5): a launch plan keeps a detached ALIAS of every nn.Parameter it takes a raw pointer of, so that `p.data = ...`
    (flat.FlatStorage re-homes every parameter on the first training step) cannot free the memory under a plan or a captured hipGraph."""
    import torch.nn as nn
    from diffuscene_amd.engine import StalePlanError, _own
    p = nn.Parameter(torch.arange(8, dtype=torch.float32))
    t = torch.ones(3)
    kept = _own((p, t, None, [p, (p, 3)], "x"))
    assert kept[1] is t and kept[2] is None and kept[4] == "x"
    alias = kept[0]
    assert not isinstance(alias, nn.Parameter) and alias.data_ptr() == p.data_ptr() and not alias.requires_grad
    assert kept[3][0].data_ptr() == p.data_ptr() and kept[3][1][0].data_ptr() == p.data_ptr() and kept[3][1][1] == 3
    old_ptr = p.data_ptr()
    p.data = torch.zeros(8)                                   # what FlatStorage / module.to() do
    assert p.data_ptr() != old_ptr and alias.data_ptr() == old_ptr
    assert torch.equal(alias, torch.arange(8, dtype=torch.float32))      # the old storage is alive and untouched
    assert issubclass(StalePlanError, RuntimeError)


def test_ddp_flush_schedule_names(monkeypatch):
    """DSC_DDP_FLUSH: 'single' is the default, 'end' its alias (the name keeps its old meaning), anything unknown is refused."""
    from diffuscene_amd.train_step import DDP_FLUSH_SCHEDULES, ddp_flush_schedule
    monkeypatch.delenv("DSC_DDP_FLUSH", raising=False)
    assert ddp_flush_schedule() == "single" and DDP_FLUSH_SCHEDULES == ("single", "block", "thirds")
    for name, want in (("end", "single"), ("single", "single"), ("block", "block"), ("thirds", "thirds")):
        monkeypatch.setenv("DSC_DDP_FLUSH", name)
        assert ddp_flush_schedule() == want
    monkeypatch.setenv("DSC_DDP_FLUSH", "sometimes")
    with pytest.raises(ValueError):
        ddp_flush_schedule()


def test_guard_allocator_is_built_and_exports_the_pluggable_allocator_entry_points():
    """tools/guard_alloc.cpp (test infrastructure of tests/test_gpu_guard.py) is compiled by __graft_entry__.build() and exports the two
    functions torch.cuda.memory.CUDAPluggableAllocator binds, plus the check / statistics calls of tools/guard_run.py."""
    import ctypes
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "_build", "libdsc_guard_alloc.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(so)
    for name in ("guard_malloc", "guard_free", "guard_check", "guard_stats", "guard_mode"):
        assert getattr(lib, name) is not None


def test_gemm_args_ctypes_mirror_matches_the_header(tmp_path):
    """The ctypes mirror of dsc_gemm_args (diffuscene_amd/_lib.GemmArgs) has the size and the field offsets the C compiler gives the
    header's struct (a field added on one side only would shift every pointer behind it)."""
    import ctypes
    import subprocess
    from diffuscene_amd import _lib
    names = [f[0] for f in _lib.GemmArgs._fields_]
    src = tmp_path / "layout.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "diffuscene_hip.h"', 'int main(void) {',
             '  printf("%zu\\n", sizeof(dsc_gemm_args));']
    lines += ['  printf("%%zu\\n", offsetof(dsc_gemm_args, %s));' % n for n in names]
    lines += ['  return 0; }']
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out[0] == ctypes.sizeof(_lib.GemmArgs), (out[0], ctypes.sizeof(_lib.GemmArgs))
    for n, off in zip(names, out[1:]):
        assert getattr(_lib.GemmArgs, n).offset == off, (n, off, getattr(_lib.GemmArgs, n).offset)
