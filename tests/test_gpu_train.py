"""GPU: hand-written backward kernels vs torch autograd of the same op (fp64 on CPU), and the full training-loss
gradient of the HIP denoiser vs the REAL reference's gradients (tests/golden/p_losses.npz)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import ref_torch as R  # noqa: E402
from oracle import weights as W  # noqa: E402
from oracle.make_golden import CASES, case_inputs  # noqa: E402


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + 77 * len(shape) + sum(shape))
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def G(t):
    return t.to(dev()).requires_grad_(True)


@pytest.mark.parametrize("m,n,k,k2", [(300, 512, 512, 0), (1000, 512, 512, 512), (64, 2048, 512, 0), (333, 25, 512, 0),
                                      (20480, 512, 512, 0)])
def test_linear_fn_backward(m, n, k, k2):
    from diffuscene_amd.autograd_ops import LinearFn
    a, w, b, r = rnd(m, k, seed=1), rnd(n, k + k2, seed=2, scale=0.1), rnd(n, seed=3), rnd(m, n, seed=4)
    a2 = rnd(m, k2, seed=5) if k2 else None
    dy = rnd(m, n, seed=6)
    ga, gw, gb, gr = G(a), G(w), G(b), G(r)
    ga2 = G(a2) if k2 else None
    y = LinearFn.apply(ga, gw, gb, ga2, gr)
    y.backward(dy.to(dev()))
    A = torch.cat([a, a2], 1).double() if k2 else a.double()
    A.requires_grad_(True)
    wd, bd, rd = w.double().requires_grad_(True), b.double().requires_grad_(True), r.double().requires_grad_(True)
    (A @ wd.T + bd + rd).backward(dy.double())
    tol = 3e-5 if m > 5000 else 5e-6
    assert rel(ga.grad, A.grad[:, :k]) < tol
    if k2:
        assert rel(ga2.grad, A.grad[:, k:]) < tol
    assert rel(gw.grad, wd.grad) < tol and rel(gb.grad, bd.grad) < tol and rel(gr.grad, rd.grad) < 1e-7


def test_smallk_and_act_backward():
    from diffuscene_amd.autograd_ops import ActFn, SmallKLinearFn
    from diffuscene_amd._lib import ACT_GELU, ACT_SILU
    M = 500
    x = rnd(M, 65, seed=7)
    for c0, k in ((0, 8), (8, 25), (33, 32), (0, 5)):
        w, b, dy = rnd(512, k, 1, seed=8 + k), rnd(512, seed=9 + k), rnd(M, 512, seed=10 + k)
        gw, gb = G(w), G(b)
        y = ActFn.apply(SmallKLinearFn.apply(x.to(dev())[:, c0:c0 + k], gw, gb), ACT_GELU)
        y.backward(dy.to(dev()))
        wd, bd = w.double()[:, :, 0].requires_grad_(True), b.double().requires_grad_(True)
        F.gelu(x[:, c0:c0 + k].double() @ wd.T + bd).backward(dy.double())
        assert rel(gw.grad[:, :, 0], wd.grad) < 5e-6 and rel(gb.grad, bd.grad) < 5e-6
    xs = rnd(1000, 64, seed=11) * 3
    gx = G(xs)
    ActFn.apply(gx, ACT_SILU).backward(torch.ones(1000, 64, device=dev()))
    xd = xs.double().requires_grad_(True)
    F.silu(xd).sum().backward()
    assert rel(gx.grad, xd.grad) < 2e-6


@pytest.mark.parametrize("B,N,mode,two", [(3, 80, 2, False), (5, 21, 1, False), (4, 12, 3, True), (2, 33, 0, False),
                                          (2, 100, 2, True), (3, 160, 0, False)])      # N > 80: the 10-row register form
def test_conv_gn_silu_backward(B, N, mode, two):
    from diffuscene_amd.autograd_ops import ConvGnSiluFn
    M, D = B * N, 512
    K = 1024 if two else 512
    a, w, b = rnd(M, 512, seed=12), rnd(D, K, seed=13, scale=0.08), rnd(D, seed=14)
    a2 = rnd(M, 512, seed=15) if two else None
    gamma, beta, res, dy = 1 + 0.1 * rnd(D, seed=16), 0.1 * rnd(D, seed=17), rnd(M, D, seed=18), rnd(M, D, seed=19)
    rows = {0: 0, 1: M, 2: B, 3: N}[mode]
    ss = rnd(rows, 2 * D, seed=20) if rows else None
    ga, gw, gb, gg, gbe, gres = G(a), G(w), G(b), G(gamma), G(beta), G(res)
    ga2 = G(a2) if two else None
    gss = G(ss) if ss is not None else None
    y = ConvGnSiluFn.apply(ga, gw, gb, gg, gbe, ga2, gss, gres, N, mode)
    y.backward(dy.to(dev()))
    A = (torch.cat([a, a2], 1) if two else a).double().requires_grad_(True)
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    gd, bed, rd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True), res.double().requires_grad_(True)
    ssd = ss.double().requires_grad_(True) if ss is not None else None
    z = (A @ wd.T + bd).reshape(B, N, D).permute(0, 2, 1)
    g = F.group_norm(z, 8, gd, bed, eps=1e-5)
    if mode:
        e = {1: lambda: ssd.reshape(B, N, 2 * D), 2: lambda: ssd[:, None, :].expand(B, N, 2 * D),
             3: lambda: ssd[None].expand(B, N, 2 * D)}[mode]().permute(0, 2, 1)
        g = g * (e[:, :D] + 1) + e[:, D:]
    ref = F.silu(g).permute(0, 2, 1).reshape(M, D) + rd
    ref.backward(dy.double())
    assert rel(y, ref) < 5e-6
    assert rel(ga.grad, A.grad[:, :512]) < 2e-5
    if two:
        assert rel(ga2.grad, A.grad[:, 512:]) < 2e-5
    for name, mine, theirs in (("w", gw.grad, wd.grad), ("bias", gb.grad, bd.grad), ("gamma", gg.grad, gd.grad),
                               ("beta", gbe.grad, bed.grad), ("res", gres.grad, rd.grad)):
        assert rel(mine, theirs) < 2e-5, (name, rel(mine, theirs))
    if ss is not None:
        assert rel(gss.grad, ssd.grad) < 2e-5


def test_weight_standardize_backward():
    from diffuscene_amd.autograd_ops import WeightStandardizeAllFn
    ws = [rnd(512, 512, 1, seed=21), rnd(512, 1024, 1, seed=22) + 0.2]
    dys = [rnd(512, 512, seed=23), rnd(512, 1024, seed=24)]
    gs = [G(w) for w in ws]
    outs = WeightStandardizeAllFn.apply(*gs)
    (outs[0] * dys[0].to(dev())).sum().add((outs[1] * dys[1].to(dev())).sum()).backward()
    for w, dy, g in zip(ws, dys, gs):
        wd = w.double().requires_grad_(True)
        mean = wd.mean(dim=(1, 2), keepdim=True)
        var = wd.var(dim=(1, 2), unbiased=False, keepdim=True)
        (((wd - mean) * (var + 1e-5).rsqrt())[:, :, 0] * dy.double()).sum().backward()
        assert rel(g.grad, wd.grad) < 1e-5


def test_layernorm_backward():
    from diffuscene_amd.autograd_ops import LayerNormFn
    x, g, r, dy = rnd(777, 512, seed=25) * 2 + 0.3, 1 + 0.1 * rnd(512, seed=26), rnd(777, 512, seed=27), rnd(777, 512, seed=28)
    gx, gg, gr = G(x), G(g), G(r)
    LayerNormFn.apply(gx, gg, gr).backward(dy.to(dev()))
    xd, gd, rd = x.double().requires_grad_(True), g.double().requires_grad_(True), r.double().requires_grad_(True)
    ((xd - xd.mean(1, keepdim=True)) * (xd.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt() * gd + rd).backward(dy.double())
    assert rel(gx.grad, xd.grad) < 5e-6 and rel(gg.grad, gd.grad) < 5e-6 and rel(gr.grad, rd.grad) < 1e-7


@pytest.mark.parametrize("B,N", [(3, 80), (2, 12), (1, 160), (2, 70), (2, 96)])   # cached softmax backward: 320 / 256 threads, (uncached kernel), 320 ragged, 384
def test_attention_cores_backward(B, N):
    from diffuscene_amd.autograd_ops import AttentionFn, LinearAttentionCrossFn, LinearAttentionSelfFn
    qkv, dy = rnd(B * N, 384, seed=29) * 2, rnd(B * N, 128, seed=30)
    sc = 32 ** -0.5

    def heads(t, n):
        return t.reshape(B, n, 4, 32).permute(0, 2, 3, 1)

    g = G(qkv)
    LinearAttentionSelfFn.apply(g, B, N, sc).backward(dy.to(dev()))
    qd = qkv.double().requires_grad_(True)
    ref = R._linear_attention_core(heads(qd[:, :128], N), heads(qd[:, 128:256], N), heads(qd[:, 256:], N))
    ref.permute(0, 2, 1).reshape(B * N, 128).backward(dy.double())
    assert rel(g.grad, qd.grad) < 1e-5, ("linear", rel(g.grad, qd.grad))

    g = G(qkv)
    AttentionFn.apply(g, B, N, sc).backward(dy.to(dev()))
    qd = qkv.double().requires_grad_(True)
    q, k, v = heads(qd[:, :128], N) * sc, heads(qd[:, 128:256], N), heads(qd[:, 256:], N)
    attn = torch.einsum("bhdi,bhdj->bhij", q, k).softmax(-1)
    torch.einsum("bhij,bhdj->bhid", attn, v).permute(0, 2, 1, 3).reshape(B * N, 128).backward(dy.double())
    assert rel(g.grad, qd.grad) < 1e-5, ("softmax", rel(g.grad, qd.grad))

    L = 7
    q0, kv = rnd(B * N, 128, seed=31) * 2, rnd(B * L, 256, seed=32) * 2
    gq, gkv = G(q0), G(kv)
    LinearAttentionCrossFn.apply(gq, gkv, B, N, L, sc).backward(dy.to(dev()))
    qd, kvd = q0.double().requires_grad_(True), kv.double().requires_grad_(True)
    ref = R._linear_attention_core(heads(qd, N), heads(kvd[:, :128], L), heads(kvd[:, 128:], L))
    ref.permute(0, 2, 1).reshape(B * N, 128).backward(dy.double())
    assert rel(gq.grad, qd.grad) < 1e-5 and rel(gkv.grad, kvd.grad) < 1e-5


def _oracle_grad_norms_fp64(kw, x, t, cond, noise, names, cross=None, iou=True):
    """Gradient norms of the p_losses objective evaluated in float64 by the oracle restatement (eps stays 1e-5)."""
    sd = {k: v.double().requires_grad_(True) for k, v in W.synth_state_dict(kw).items()}
    tb = {k: v.double() for k, v in R.schedule_tables(1e-4, 0.02, 1000, "v").items()}
    emb = R.sinusoidal_embedding
    R.sinusoidal_embedding = lambda tt, dim: emb(tt, dim).double()
    cd = cross.double() if cross is not None else None
    try:
        lw, _, _ = R.p_losses(tb, lambda xt, tt: R.unet1d_forward(sd, kw, xt, tt, cond.double(), cd), x.double(), t,
                              noise.double(), R.dims_from_kwargs(kw), True, iou, W.DATASET_STATS)
        lw.mean().backward()
    finally:
        R.sinusoidal_embedding = emb
    return np.array([float(sd[k].grad.norm()) for k in names]), lw.detach()


@pytest.mark.parametrize("mean_type,iou,B,N", [("v", True, 5, 21), ("eps", True, 3, 12), ("x0", True, 2, 80), ("v", False, 4, 12)])
def test_fused_loss_kernel_matches_torch_definition(tmp_path, mean_type, iou, B, N):
    """dsc_ddpm_loss_f32 (all loss terms + d loss / d denoise_out in one kernel) vs the torch-op definition of the same
    loss under autograd (train_loss.diffusion_losses(fused=False)), incl. overlapping boxes and empty slots."""
    from diffuscene_amd import train_loss as tg
    from diffuscene_amd.networks.diffusion_ddpm import GaussianDiffusion, get_betas
    stats = tmp_path / "dataset_stats.txt"
    stats.write_text(json.dumps(W.DATASET_STATS))
    cfg = dict(objectness_dim=0, class_dim=25, angle_dim=2, objfeat_dim=32)
    d = GaussianDiffusion(cfg, get_betas("linear", 1e-4, 0.02, 1000), "mse", mean_type, "fixedsmall", True, iou,
                          str(stats) if iou else None)
    tb = d.tables(dev())
    x0 = W.synth_scene_batch(B, N, 25, 32, seed=21)
    x0[:, :, 0:3] *= 0.3                       # crowd the boxes so that many pairs overlap
    t = torch.tensor([(17 + 233 * i) % 1000 for i in range(B)], dtype=torch.int64)
    noise = W.synth_noise((B, N, 65), 3, "lossn")
    tbc = R.schedule_tables(1e-4, 0.02, 1000, mean_type)
    xt = R.q_sample(tbc, x0, t, noise)
    target = {"v": R.predict_v(tbc, x0, t, noise), "eps": noise, "x0": x0}[mean_type]
    out = target + 0.3 * rnd(B, N, 65, seed=22)
    res = []
    for fused in (True, False):
        o = out.to(dev()).requires_grad_(True)
        lw, scal = tg.diffusion_losses(d, tb, x0.to(dev()), xt.to(dev()), target.to(dev()), o, t.to(dev()), fused=fused)
        (lw * torch.arange(1, B + 1, device=dev())).sum().backward()
        res.append((lw.detach(), {k: float(v) for k, v in scal.items()}, o.grad))
    (lw_f, sc_f, g_f), (lw_t, sc_t, g_t) = res
    assert rel(lw_f, lw_t) < 2e-5
    for k in sc_t:
        assert abs(sc_f[k] - sc_t[k]) <= 2e-5 * max(1.0, abs(sc_t[k])), (k, sc_f[k], sc_t[k])
    if iou:
        assert sc_t["loss.bbox_iou"] > 1e-3                      # the IoU term is really exercised
    assert rel(g_f, g_t) < 5e-5, rel(g_f, g_t)


def test_text_conditioned_training_gradients_vs_fp64():
    """config/text (cross-attention on 512-d text tokens): loss and every parameter gradient of the
    HIP path against the fp64 oracle; also gradients w.r.t. the conditioning inputs."""
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    kw, x, t, cond, cross = case_inputs("text_bedroom")
    net = Unet1D(**kw)
    net.load_state_dict(W.synth_state_dict(kw))
    net.to(dev())
    cfg = dict(objectness_dim=0, class_dim=22, angle_dim=2, objfeat_dim=32)
    diff = DiffusionPoint(net, cfg, time_num=1000, model_mean_type="v", loss_separate=True, loss_iou=False)
    noise = W.synth_noise(tuple(x.shape), 0, "train_noise")
    gc, gx = cond.to(dev()).requires_grad_(True), cross.to(dev()).requires_grad_(True)
    losses, _ = diff.diffusion.p_losses(diff._denoise, x.to(dev()), t.to(dev()), noise=noise.to(dev()), condition=gc,
                                        condition_cross=gx)
    losses.mean().backward()
    names = [k for k, _ in net.named_parameters()]
    truth, lw = _oracle_grad_norms_fp64(kw, x, t, cond, noise, names, cross=cross, iou=False)
    assert rel(losses, lw) < 1e-5
    gn = np.array([float(p.grad.norm()) for _, p in net.named_parameters()])
    err = np.abs(gn - truth) / np.maximum(truth, 1e-3 * truth.max())
    print("text model: grad-norm rel err vs fp64 max %.3g at %s" % (err.max(), names[int(err.argmax())]))
    assert err.max() < 1e-4
    assert gc.grad is not None and gx.grad is not None and float(gx.grad.abs().sum()) > 0


def test_rearrange_training_loss_path():
    """config/rearrange: 5 diffused channels, non-separate init/final conv, 512-d per-token conditioning, arrange loss."""
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    kw, x, t, cond, _ = case_inputs("rearrange_living")
    net = Unet1D(**kw)
    net.load_state_dict(W.synth_state_dict(kw))
    net.to(dev())
    cfg = dict(objectness_dim=0, class_dim=25, angle_dim=2, objfeat_dim=32, room_arrange_condition=True)
    diff = DiffusionPoint(net, cfg, time_num=1000, model_mean_type="v", loss_separate=True, loss_iou=False)
    noise = W.synth_noise(tuple(x.shape), 0, "train_noise")
    losses, scal = diff.diffusion.p_losses(diff._denoise, x.to(dev()), t.to(dev()), noise=noise.to(dev()),
                                           condition=cond.to(dev()), condition_cross=None)
    losses.mean().backward()
    assert set(scal) == {"loss.trans", "loss.angle"}
    sd = W.synth_state_dict(kw)
    tb = R.schedule_tables(1e-4, 0.02, 1000, "v")
    with torch.no_grad():
        xt = R.q_sample(tb, x, t, noise)
        out = R.unet1d_forward(sd, kw, xt, t, cond, None)
        tgt = R.predict_v(tb, x, t, noise)
        ref = (((tgt[:, :, :3] - out[:, :, :3]) ** 2).mean((1, 2)) + ((tgt[:, :, 3:] - out[:, :, 3:]) ** 2).mean((1, 2))) \
            * tb["loss_weight"][t]
    assert rel(losses, ref) < 1e-4
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


@pytest.mark.parametrize("name", ["uncond_bedroom", "uncond_living"])
def test_training_gradients_match_reference(golden_dir, tmp_path, name):
    """p_losses (+IoU term) and the gradient of EVERY parameter.  Losses and committed gradient slices vs the real
    reference (golden) within 1e-4.  Gradient norms of all 442 tensors: the reference's own fp32 CPU result is up to
    2.6e-4 away from an fp64 evaluation (time-MLP weights: long, cancelling reductions), so the HIP path is held to 1e-4
    against the fp64 truth and to 1e-3 against the fp32 reference."""
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    g = np.load(os.path.join(golden_dir, "p_losses.npz"))
    names = json.load(open(os.path.join(golden_dir, "grad_names_%s.json" % name)))
    kw, x, t, cond, cross = case_inputs(name)
    stats = tmp_path / "dataset_stats.txt"
    stats.write_text(json.dumps(W.DATASET_STATS))
    net = Unet1D(**kw)
    net.load_state_dict(W.synth_state_dict(kw))
    net.to(dev())
    cfg = dict(objectness_dim=0, class_dim=kw["class_dim"], angle_dim=2, objfeat_dim=32)
    diff = DiffusionPoint(net, cfg, time_num=1000, model_mean_type="v", loss_separate=True, loss_iou=True,
                          train_stats_file=str(stats))
    noise = W.synth_noise(tuple(x.shape), 0, "train_noise")
    losses, scal = diff.diffusion.p_losses(diff._denoise, x.to(dev()), t.to(dev()), noise=noise.to(dev()),
                                           condition=cond.to(dev()), condition_cross=None)
    losses.mean().backward()
    assert rel(losses, g[name + ".losses"]) < 1e-4
    for k, v in scal.items():
        assert abs(float(v.detach()) - float(g[name + "." + k])) <= 1e-4 * max(1.0, abs(float(g[name + "." + k]))), k
    params = dict(net.named_parameters())
    gn = np.array([float(params[k].grad.norm()) for k in names])
    ref = g[name + ".grad_norms"]

    def relerr(a, b):
        return np.abs(a - b) / np.maximum(b, 1e-3 * b.max())

    # fp64 evaluation of the same loss by the oracle (same eps rule): the arbiter between two fp32 implementations
    truth, _ = _oracle_grad_norms_fp64(kw, x, t, cond, noise, names)
    e_hip, e_ref = relerr(gn, truth), relerr(ref, truth)
    print("grad-norm rel err vs fp64: HIP max %.3g (%s) | reference fp32 max %.3g (%s)" % (
        e_hip.max(), names[int(e_hip.argmax())], e_ref.max(), names[int(e_ref.argmax())]))
    assert e_hip.max() < 1e-4                      # the HIP path is held to 1e-4 against the fp64 truth
    assert relerr(gn, ref).max() < 1e-3            # and stays within the reference's own fp32 noise (max 2.6e-4) of it
    assert rel(net.init_conv.bias.grad, g[name + ".grad.init_conv.bias"]) < 1e-4
    assert rel(net.mid_attn.fn.fn.to_qkv.weight.grad[:8, :16, 0], g[name + ".grad.mid_attn.to_qkv"]) < 1e-4
    assert rel(net.downs[0][0].block1.proj.weight.grad[:8, :16, 0], g[name + ".grad.downs0.block1.proj"]) < 1e-4


def test_train_on_batch_runs_and_learns(tmp_path):
    """train_on_batch end to end (Adam, clip, logging): the loss on a fixed batch must go down."""
    from diffuscene_amd.networks import optimizer_factory
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import DiffusionSceneLayout_DDPM, train_on_batch
    from diffuscene_amd.stats_logger import StatsLogger
    stats = tmp_path / "dataset_stats.txt"
    stats.write_text(json.dumps(W.DATASET_STATS))
    cfg = {"type": "diffusion_scene_layout_ddpm", "net_type": "unet1d", "point_dim": 62, "latent_dim": 0,
           "room_mask_condition": False, "sample_num_points": 12, "objectness_dim": 0, "objfeat_dim": 32,
           "class_dim": 22, "angle_dim": 2, "learnable_embedding": True, "instance_condition": True,
           "instance_emb_dim": 128,
           "diffusion_kwargs": dict(schedule_type="linear", beta_start=1e-4, beta_end=0.02, time_num=1000,
                                    loss_type="mse", model_mean_type="v", model_var_type="fixedsmall",
                                    loss_separate=True, loss_iou=True, train_stats_file=str(stats)),
           "net_kwargs": dict(W.UNCOND_BEDROOM)}
    torch.manual_seed(0)
    m = DiffusionSceneLayout_DDPM(23, None, cfg).to(dev())
    opt = optimizer_factory({"optimizer": "Adam", "lr": 2e-4}, m.parameters())
    x = W.synth_scene_batch(8, 12, 22, 32, seed=1).to(dev())
    sample = {"translations": x[:, :, 0:3].contiguous(), "sizes": x[:, :, 3:6].contiguous(),
              "angles": x[:, :, 6:8].contiguous(), "class_labels": x[:, :, 8:30].contiguous(),
              "objfeats_32": x[:, :, 30:62].contiguous(), "room_layout": torch.zeros(8, 1, 64, 64, device=dev())}
    ls = []
    for i in range(12):
        torch.manual_seed(123)                       # same t / noise every step: a fixed objective
        ls.append(train_on_batch(m, opt, sample, {"training": {"max_grad_norm": 10}}))
    print("losses", ["%.4f" % v for v in ls])
    assert all(np.isfinite(ls)) and ls[-1] < 0.8 * ls[0]
    assert StatsLogger.instance()["gradnorm"].value > 0
    assert m.positional_embedding.grad is not None and float(m.positional_embedding.grad.abs().sum()) > 0


def test_fused_adam_matches_torch_adam_with_clipping():
    """FusedAdam (csrc/optim.hip) vs torch.nn.utils.clip_grad_norm_ + torch.optim.Adam over several steps, incl. odd
    sizes, a tensor spanning several chunks and a state_dict round trip into a plain torch.optim.Adam."""
    from diffuscene_amd.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(512, 512), (70001,), (3, 5, 7), (1,), (2048, 64), (13,)]
    ref_p = [torch.nn.Parameter(torch.randn(s, device=dev()) * 0.3) for s in shapes]
    my_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    ref = torch.optim.Adam(ref_p, lr=2e-3, weight_decay=0.0)
    mine = FusedAdam(my_p, lr=2e-3, weight_decay=0.0)
    for it in range(4):
        gs = [torch.randn(s, device=dev()) * (5.0 if it % 2 == 0 else 0.01) for s in shapes]     # clipped / unclipped steps
        for p, q, g in zip(ref_p, my_p, gs):
            p.grad = g.clone()
            q.grad = g.clone()
        n_ref = torch.nn.utils.clip_grad_norm_(ref_p, 10.0)
        n_my = mine.clip_grad_norm_(10.0)
        assert abs(float(n_ref) - float(n_my)) <= 1e-5 * float(n_ref)
        ref.step()
        mine.step()
        for p, q in zip(ref_p, my_p):
            assert torch.allclose(p, q, rtol=2e-6, atol=1e-7), (it, p.shape, float((p - q).abs().max()))
    sd = mine.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 4.0
    import copy
    ref.load_state_dict(copy.deepcopy(sd))                            # interchangeable checkpoint format (deep copies:
    mine2 = FusedAdam(my_p, lr=2e-3)                                  #  load_state_dict aliases same-device tensors)
    mine2.load_state_dict(copy.deepcopy(ref.state_dict()))
    for p, q in zip(ref_p, my_p):
        g = torch.randn_like(p)
        p.grad, q.grad = g.clone(), g.clone()
    ref.step()
    mine2.step()                                                      # no clip call: coefficient must not be applied
    for p, q in zip(ref_p, my_p):
        assert torch.allclose(p, q, rtol=2e-6, atol=1e-7)
    cpu = FusedAdam([torch.nn.Parameter(torch.zeros(4))], lr=1e-3)
    cpu.param_groups[0]["params"][0].grad = torch.ones(4)
    with pytest.raises(RuntimeError):
        cpu.step()


@pytest.mark.parametrize("gemm_arith", ["split", "f32"], indirect=True)
@pytest.mark.parametrize("scale", [1, 40])
def test_grouped_weight_gradient_gemm(scale, gemm_arith):
    """dsc_gemm_tn_grouped_[split_]f32: many weight gradients in one launch (mixed shapes: two K segments, zero-padded small K with
    kvalid, bias column sums, tiny token counts, a ragged token tail) vs fp64; `scale` makes the group large enough for the unsplit
    path; both arithmetics: the split-bf16 form (default) and the exact-f32 MFMA kernel (set_gemm_arithmetic("f32"))."""
    from diffuscene_amd.train_plan import HipBackend
    arith = gemm_arith
    be = HipBackend(dev())
    assert be.split == (arith == "split")
    M = 1290
    shapes = [(M, 512, 512, 0, None, True), (M, 512, 512, 512, None, False), (M, 512, 32, 0, 25, True),
              (64, 1024, 512, 0, None, True), (M, 384, 512, 0, None, False), (M, 32, 512, 0, None, True)] * scale
    items, refs = [], []
    for i, (m, n, k1, k2, kv, bias) in enumerate(shapes):
        a, dy = rnd(m, k1, seed=100 + i).to(dev()), rnd(m, n, seed=200 + i).to(dev())
        a2 = rnd(m, k2, seed=300 + i).to(dev()) if k2 else None
        K = k1 + k2
        out = torch.full((n, kv or K), float("nan"), device=dev())
        db = torch.full((n,), float("nan"), device=dev()) if bias else None
        items.append(dict(a=a, dy=dy, out=out, a2=a2, kvalid=kv, dbias=db))
        A = torch.cat([a, a2], 1) if k2 else a
        refs.append(((dy.double().T @ A.double())[:, :kv or K], dy.double().sum(0)))
    step = be.gemm_tn_grouped(items)
    be.finalize()
    assert step["step"][2] == ("dsc_gemm_tn_grouped_split_f32" if arith == "split" else "dsc_gemm_tn_grouped_f32")
    be.run([step], torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for it, (rw, rb) in zip(items, refs):
        assert rel(it["out"], rw) < 5e-6
        if it["dbias"] is not None:
            assert rel(it["dbias"], rb) < 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize("splits_scale", [1, 40])
def test_weight_gradient_block_bodies_are_bit_identical(splits_scale):
    """The producer / consumer form of the split-bf16 weight-gradient launch (round 5: four waves stage two steps ahead, four multiply)
    against the round-4 body (every wave stages and multiplies): same LDS image, same six products in the same order per accumulator,
    same bias partial sums -> torch.equal on every output: strided operands, two K segments, kvalid, a ragged token tail and token slices with the slab reduction."""
    from diffuscene_amd import _lib
    from diffuscene_amd.train_plan import HipBackend
    if not _lib.split_enabled():
        pytest.skip("exact-f32 arithmetic selected")
    M = 1290
    shapes = [(M, 512, 512, 0, None, True), (M, 512, 512, 512, None, False), (M, 512, 32, 0, 25, True),
              (64, 1024, 512, 0, None, True), (M, 384, 512, 0, None, False), (M, 32, 512, 0, None, True)] * splits_scale
    outs = {}
    for form in (0, 1, 2):                      # round 4 body, producer / consumer waves (round 5), 256 x 256 tiles (round 6)
        prev = _lib.fn("dsc_set_tn_split_form")(form)
        try:
            be = HipBackend(dev())
            items = []
            for i, (m, n, k1, k2, kv, bias) in enumerate(shapes):
                a, dy = rnd(m, k1, seed=100 + i).to(dev()), rnd(m, n, seed=200 + i).to(dev())
                if i % 6 == 4:                      # a column slice of a wider buffer (leading dimension != width)
                    wide = torch.zeros(m, n + 64, device=dev())
                    wide[:, :n] = dy
                    dy = wide[:, :n]
                a2 = rnd(m, k2, seed=300 + i).to(dev()) if k2 else None
                out = torch.full((n, kv or (k1 + k2)), float("nan"), device=dev())
                db = torch.full((n,), float("nan"), device=dev()) if bias else None
                items.append(dict(a=a, dy=dy, out=out, a2=a2, kvalid=kv, dbias=db))
            step = be.gemm_tn_grouped(items)
            be.finalize()
            be.run([step], torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            outs[form] = [(it["out"].clone(), None if it["dbias"] is None else it["dbias"].clone()) for it in items]
        finally:
            _lib.fn("dsc_set_tn_split_form")(prev)
    for other in (1, 2):
        for i, ((w0, b0), (w1, b1)) in enumerate(zip(outs[0], outs[other])):
            assert bool(torch.isfinite(w1).all()) and torch.equal(w0, w1), "form %d, group %d: dW differs from the round-4 body" % (other, i)
            if b0 is not None:
                assert torch.equal(b0, b1), "form %d, group %d: bias gradient differs" % (other, i)


@pytest.mark.gpu
def test_condition_mlps_run_on_the_hip_gemm_and_match_torch():
    """The wrapper's condition layers (reference diffusion_scene_layout_ddpm.py:94-125, :47-51): Linear -> LeakyReLU(0.1) ->
    Linear without biases on an un-aligned input width, and the biased text projection -- forward and all gradients vs the
    same modules evaluated by torch in fp64.  Tolerance 1e-5 relative (fp32 MFMA vs fp64)."""
    import torch.nn as nn
    from diffuscene_amd.networks.diffusion_scene_layout_ddpm import _HipLinear, _TwoLayer
    d = torch.device("cuda:0")
    torch.manual_seed(3)
    for n_in, n_out, lead in ((60, 384, (3, 21)), (12, 128, (12,)), (65, 64, (2, 12))):
        m = _TwoLayer(n_in, n_out).to(d)
        assert sorted(m.state_dict()) == ["0.weight", "2.weight"]              # the reference's nn.Sequential keys
        x = torch.randn(*lead, n_in, device=d, requires_grad=True)
        y = m(x)
        g = torch.randn_like(y)
        y.backward(g)
        ref = nn.Sequential(nn.Linear(n_in, n_out, bias=False), nn.LeakyReLU(0.1), nn.Linear(n_out, n_out, bias=False)).double()
        ref.load_state_dict({k: v.double().cpu() for k, v in m.state_dict().items()})
        xr = x.detach().double().cpu().requires_grad_(True)
        yr = ref(xr)
        yr.backward(g.double().cpu())
        for a, b in ((y, yr), (x.grad, xr.grad), (m[0].weight.grad, ref[0].weight.grad), (m[2].weight.grad, ref[2].weight.grad)):
            assert float((a.detach().double().cpu() - b).abs().max()) <= 1e-5 * float(b.abs().max()), (n_in, n_out)
    lin = _HipLinear(768, 512).to(d)
    x = torch.randn(4, 7, 768, device=d, requires_grad=True)
    y = lin(x)
    g = torch.randn_like(y)
    y.backward(g)
    yr = torch.nn.functional.linear(x.detach().double().cpu(), lin.weight.detach().double().cpu(), lin.bias.detach().double().cpu())
    assert float((y.detach().double().cpu() - yr).abs().max()) <= 1e-5 * float(yr.abs().max())
    dwr = g.double().cpu().reshape(-1, 512).t() @ x.detach().double().cpu().reshape(-1, 768)
    assert float((lin.weight.grad.double().cpu() - dwr).abs().max()) <= 1e-5 * float(dwr.abs().max())
    assert float((lin.bias.grad.double().cpu() - g.double().cpu().sum(dim=(0, 1))).abs().max()) <= 1e-5 * float(g.abs().sum())
