"""CPU: pin the oracle restatement (oracle/ref_torch.py) against golden vectors produced by the
REAL reference modules (oracle/make_golden.py).  Tolerances: fp32 round-off only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ref_torch as R
from oracle import weights as W
from oracle.make_golden import CASES, case_inputs, noise_list

RTOL = 2e-5   # oracle vs real reference: same ATen ops, differences are summation-order noise


def _rel(a, b):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def test_state_dict_layout_matches_reference(golden_dir):
    keys = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    for name, (kw, *_r) in CASES.items():
        spec = W.unet1d_param_spec(**kw)
        assert {k: list(s) for k, s in spec.items()} == {k: s for k, s in keys[name]}, name   # order-free


def test_param_count():
    n = sum(int(np.prod(s)) for s in W.unet1d_param_spec(**W.UNCOND_BEDROOM).values())
    assert n == 77676094           # SURVEY.md 3.4 (probe of the real module)
    n = sum(int(np.prod(s)) for s in W.unet1d_param_spec(**W.UNCOND_LIVING).values())
    assert n == 77679169


def test_schedule_tables_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, "schedule_v_T1000.npz"))
    tb = R.schedule_tables(1e-4, 0.02, 1000, "v")
    for k in g.files:
        assert np.array_equal(tb[k].numpy(), g[k]), k


@pytest.mark.parametrize("name", list(CASES))
def test_unet_forward_matches_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "unet_forward.npz"))
    kw, x, t, cond, cross = case_inputs(name)
    sd = W.synth_state_dict(kw)
    with torch.no_grad():
        out = R.unet1d_forward(sd, kw, x, t, cond, cross)
    assert out.shape == g[name].shape
    assert _rel(out, g[name]) < RTOL


@pytest.mark.parametrize("name", ["uncond_bedroom", "uncond_living"])
def test_p_losses_matches_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "p_losses.npz"))
    kw, x, t, cond, cross = case_inputs(name)
    sd = W.synth_state_dict(kw)
    tb = R.schedule_tables(1e-4, 0.02, 1000, "v")
    noise = W.synth_noise(tuple(x.shape), 0, "train_noise")
    with torch.no_grad():
        lw, scal, _ = R.p_losses(tb, lambda xt, tt: R.unet1d_forward(sd, kw, xt, tt, cond, cross), x, t, noise,
                                 R.dims_from_kwargs(kw), loss_separate=True, loss_iou=True, stats=W.DATASET_STATS)
    assert _rel(lw, g[name + ".losses"]) < RTOL
    for k, v in scal.items():
        ref = float(g[name + "." + k])
        assert abs(float(v) - ref) <= RTOL * max(1.0, abs(ref)), k


def test_reverse_chain_T50_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "chains.npz"))
    kw, x, t, cond, _ = case_inputs("uncond_bedroom")
    B, N, C = x.shape
    sd = W.synth_state_dict(kw)
    tb = R.schedule_tables(1e-4, 0.02, 50, "v")
    den = lambda xt, tt: R.unet1d_forward(sd, kw, xt, tt, cond, None)
    with torch.no_grad():
        s = R.p_sample_loop(tb, den, (B, N, C), noise_list([(B, N, C)] * 51, 1, "chain50_"), 50, True)
        assert _rel(s, g["uncond_T50"]) < 1e-4
        s = R.p_sample_loop(tb, den, (B, N, C), noise_list([(B, N, C)] * 51, 3, "noclip_"), 50, False)
        assert _rel(s, g["uncond_noclip_T50"]) < 1e-4
        shapes = [(B, N, C)]
        for _ in range(50):
            shapes += [(B, 3, C), (B, N, C)]
        s = R.p_sample_loop_complete(tb, den, (B, N, C), noise_list(shapes, 2, "complete_"), 50,
                                     x[:, :3, :].contiguous(), True)
        assert _rel(s, g["complete_T50"]) < 1e-4


@pytest.mark.parametrize("mean_type", ["eps", "x0"])
def test_other_prediction_types_match_reference(golden_dir, mean_type):
    """'eps' (config/uncond/*_eps.yaml) and 'x0': the restatement's p_losses and T = 50 chains vs the REAL reference
    (tests/golden/meantypes.npz, oracle/make_golden_meantypes.py); every other golden uses 'v'."""
    g = np.load(os.path.join(golden_dir, "meantypes.npz"))
    kw, x, t, cond, _ = case_inputs("uncond_bedroom")
    B, N, C = x.shape
    sd = W.synth_state_dict(kw)
    den = lambda xt, tt: R.unet1d_forward(sd, kw, xt, tt, cond, None)
    noise = W.synth_noise(tuple(x.shape), 0, "train_noise")
    with torch.no_grad():
        lw, scal, _ = R.p_losses(R.schedule_tables(1e-4, 0.02, 1000, mean_type), den, x, t, noise, R.dims_from_kwargs(kw), loss_separate=True,
                                 loss_iou=True, stats=W.DATASET_STATS, mean_type=mean_type)
        assert _rel(lw, g[mean_type + ".losses"]) < RTOL
        for k, v in scal.items():
            ref = float(g[mean_type + "." + k])
            assert abs(float(v) - ref) <= RTOL * max(1.0, abs(ref)), k
        tb = R.schedule_tables(1e-4, 0.02, 50, mean_type)
        for clip, seed, tag in ((True, 11, "clip"), (False, 12, "noclip")):
            s = R.p_sample_loop(tb, den, (B, N, C), noise_list([(B, N, C)] * 51, seed, "mt_%s_" % tag), 50, clip, mean_type=mean_type)
            assert _rel(s, g["%s.T50.%s" % (mean_type, tag)]) < 1e-4, (mean_type, tag)


def test_text_and_arrange_chains_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "chains.npz"))
    kw, x, t, cond, cross = case_inputs("text_bedroom")
    sd = W.synth_state_dict(kw)
    tb = R.schedule_tables(1e-4, 0.02, 20, "v")
    with torch.no_grad():
        s = R.p_sample_loop(tb, lambda xt, tt: R.unet1d_forward(sd, kw, xt, tt, cond, cross), tuple(x.shape),
                            noise_list([tuple(x.shape)] * 21, 4, "text_"), 20, True)
    assert _rel(s, g["text_T20"]) < 1e-4
    kw, x, t, cond, _ = case_inputs("rearrange_living")
    sd = W.synth_state_dict(kw)
    tb = R.schedule_tables(1e-4, 0.02, 50, "v")
    B, N = x.shape[:2]
    full = W.synth_scene_batch(B, N, 25, 32, 5)
    with torch.no_grad():
        s = R.p_sample_loop_arrange(tb, lambda xt, tt: R.unet1d_forward(sd, kw, xt, tt, cond, None), (B, N, 65),
                                    noise_list([(B, N, 5)] * 51, 5, "arrange_"), 50, full,
                                    dict(translation_dim=3, size_dim=3, bbox_dim=8), True)
    assert s.shape == g["arrange_T50"].shape
    assert _rel(s, g["arrange_T50"]) < 1e-4


def test_reverse_chain_T1000_matches_reference(golden_dir):
    """Full-length chain (B=1, N=12): the restatement must stay within 1e-4 relative of the real
    reference over 1000 dependent steps (SURVEY.md 8c noise floor: 1.3e-6)."""
    g = np.load(os.path.join(golden_dir, "chains.npz"))
    kw, x, t, cond, _ = case_inputs("uncond_bedroom")
    _, N, C = x.shape
    sd = W.synth_state_dict(kw)
    tb = R.schedule_tables(1e-4, 0.02, 1000, "v")
    with torch.no_grad():
        s = R.p_sample_loop(tb, lambda xt, tt: R.unet1d_forward(sd, kw, xt, tt, cond[:1], None), (1, N, C),
                            noise_list([(1, N, C)] * 1001, 1, "chain1000_"), 1000, True)
    assert _rel(s, g["uncond_T1000"]) < 1e-4


def test_variational_bound_terms_match_reference(golden_dir):
    """loss_type 'kl' / prior_kl / all_kl restatements vs the real reference (oracle/make_golden_bpd.py)."""
    from oracle.make_golden_bpd import T_LOOP
    g = np.load(os.path.join(golden_dir, "bpd.npz"))
    kw, x, t, cond, _ = case_inputs("uncond_bedroom")
    sd = W.synth_state_dict(kw)
    tb = R.schedule_tables(1e-4, 0.02, 1000, "v")
    noise = W.synth_noise(tuple(x.shape), 11, "bpd_q")
    den = lambda xt, tt: R.unet1d_forward(sd, kw, xt, tt, cond, None)
    with torch.no_grad():
        x_t = R.q_sample(tb, x, t, noise)
        out = den(x_t, t)
        for clip in (True, False):
            kl, xr = R.vb_terms_bpd(tb, x, x_t, t, out, clip)
            assert _rel(kl, g["vb_kl_clip%d" % clip]) < 5e-5
            assert _rel(xr, g["vb_xstart_clip%d" % clip]) < RTOL
        assert _rel(kl, g["p_losses_kl"]) < 5e-5
        assert _rel(R.prior_bpd(tb, x), g["prior_bpd"]) < RTOL
        tb20 = R.schedule_tables(1e-4, 0.02, T_LOOP, "v")
        seq = noise_list([tuple(x.shape)] * T_LOOP, 12, "bpd_loop")
        r = R.calc_bpd_loop(tb20, den, x, seq, True)
    for a, b in zip(r, g["all_kl"]):
        assert abs(float(a) - b) <= 1e-4 * abs(b)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json shapes (tests/golden/wide.npz, real reference outputs from oracle/make_golden_wide.py): N=80 forward and
# training loss, completion with P=20 given objects, 5-channel re-arrangement at N=80, L=32 text tokens, trajectory
# ---------------------------------------------------------------------------------------------------------------------
def test_oracle_at_baseline_shapes_matches_reference(golden_dir):
    from oracle.make_golden_wide import wide_inputs
    g = np.load(os.path.join(golden_dir, "wide.npz"))
    kw, x, t, cond, _ = wide_inputs("living80")
    sd = W.synth_state_dict(kw)
    B, N, C = x.shape
    with torch.no_grad():
        assert _rel(R.unet1d_forward(sd, kw, x, t, cond, None), g["living80.forward"]) < RTOL
        tb = R.schedule_tables(1e-4, 0.02, 1000, "v")
        noise = W.synth_noise(tuple(x.shape), 40, "train_noise")
        lw, scal, _ = R.p_losses(tb, lambda xt, tt: R.unet1d_forward(sd, kw, xt, tt, cond, None), x, t, noise,
                                 R.dims_from_kwargs(kw), loss_separate=True, loss_iou=True, stats=W.DATASET_STATS)
        assert _rel(lw, g["living80.losses"]) < RTOL
        for k, v in scal.items():
            ref = float(g["living80." + k])
            assert abs(float(v) - ref) <= RTOL * max(1.0, abs(ref)), k
        # completion N=80, P=20, T=50
        tb50 = R.schedule_tables(1e-4, 0.02, 50, "v")
        shapes = [(B, N, C)]
        for _ in range(50):
            shapes += [(B, 20, C), (B, N, C)]
        s = R.p_sample_loop_complete(tb50, lambda xt, tt: R.unet1d_forward(sd, kw, xt, tt, cond, None), (B, N, C),
                                     noise_list(shapes, 41, "complete80_"), 50, x[:, :20, :].contiguous(), True)
        assert _rel(s, g["complete80.T50"]) < 1e-4
        # text, L=32
        kwt, xt_, tt_, condt, crosst = wide_inputs("text32")
        sdt = W.synth_state_dict(kwt)
        assert _rel(R.unet1d_forward(sdt, kwt, xt_, tt_, condt, crosst), g["text32.forward"]) < RTOL
        # trajectory: x_T, then the states after t = 49 (first step), 40, 30, 20, 10, 0
        kwb = W.UNCOND_BEDROOM
        sdb = W.synth_state_dict(kwb)
        condb = W.synth_condition(2, 12, 128, 0).contiguous()
        seq = noise_list([(2, 12, 62)] * 51, 44, "traj_")
        img, imgs = seq[0], [seq[0]]
        for i, tt in enumerate(reversed(range(50))):
            t_ = torch.full((2,), tt, dtype=torch.int64)
            img = R.p_sample_step(tb50, img, t_, R.unet1d_forward(sdb, kwb, img, t_, condb, None), seq[i + 1], True, "v")
            if tt % 10 == 0 or tt == 49:
                imgs.append(img)
        assert _rel(torch.stack(imgs), g["traj.T50"]) < 1e-4
