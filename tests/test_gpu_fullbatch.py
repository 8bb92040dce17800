"""GPU: parity of the OTHER BASELINE.json configurations at their full per-GPU batch against outputs of the REAL reference
(tests/golden/fullbatch.npz, produced by oracle/make_golden_fullbatch.py with the reference's own modules) -- what
`bench.py --config bedroom21|text|arrange|complete` times:
  bedroom21  B=256, N=21, C=62          training step (p_losses + IoU, scalars, 16 gradient norms) and one reverse step
  text       B=128, N=12, L=32 tokens   the same with cross-attention (+ the gradient that flows back to the text features)
  arrange    B=128, N=80, 5 channels    the arrange branch of p_losses, per-token 512-d condition, one reverse step
  complete   B=128, N=80, 20 given      a 10-step completion loop, eagerly and from its hipGraph
tests/test_gpu_b256.py does the same for the metric configuration (B=256, N=80).  Tolerances as there: 1e-4 norm-relative and
element-wise on outputs and losses, 1e-3 on gradient norms."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import weights as W  # noqa: E402
from oracle.make_golden_fullbatch import COMPLETE_P, COMPLETE_T, complete_noise, fullbatch_inputs  # noqa: E402

from test_gpu_wide import check, dev  # noqa: E402

_PART_KEYS = ('loss.bbox', 'loss.trans', 'loss.size', 'loss.angle', 'loss.class', 'loss.object', 'loss.objfeat', 'loss.liou',
              'loss.bbox_iou')


def _model(name, tmp_path, time_num=1000):
    from diffuscene_amd.networks.denoise_net import Unet1D
    from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
    kw = fullbatch_inputs(name)[0]
    stats = tmp_path / "dataset_stats.txt"
    stats.write_text(json.dumps(W.DATASET_STATS))
    net = Unet1D(**kw)
    net.load_state_dict(W.synth_state_dict(kw))
    net.to(dev())
    arrange = name == "arrange"
    cfg = dict(objectness_dim=0, class_dim=kw["class_dim"], angle_dim=2, objfeat_dim=32)
    if arrange:
        cfg["room_arrange_condition"] = True
    diff = DiffusionPoint(net, cfg, time_num=time_num, model_mean_type="v", loss_separate=True, loss_iou=not arrange,
                          train_stats_file=str(stats))
    return net, diff


@pytest.mark.parametrize("gemm_arith", ["split", "f32"], indirect=True)
@pytest.mark.parametrize("name", ["bedroom21", "text", "arrange"])
def test_training_step_at_full_batch(name, golden_dir, tmp_path, gemm_arith):
    from diffuscene_amd._lib import SS_PER_SLOT, SS_PER_TOKEN
    from diffuscene_amd.flat import FlatStorage
    from diffuscene_amd.train_plan import HipBackend, TrainPlan
    from diffuscene_amd.train_step import _capture
    g = np.load(os.path.join(golden_dir, "fullbatch.npz"))
    names = json.load(open(os.path.join(golden_dir, "grad_names_fullbatch.json")))[name]
    kw, x, t, cond, cross, noise, _ = fullbatch_inputs(name)
    net, diff = _model(name, tmp_path)
    flat = FlatStorage(net)
    B, N, C = x.shape
    L = cross.shape[1] if cross is not None else 0
    per_token = cond.shape[-1] == 512
    plan = TrainPlan(net, flat, diff.diffusion, B, N, SS_PER_TOKEN if per_token else SS_PER_SLOT, cond.shape[-1], L,
                     512 if L else 0, HipBackend(dev()))
    plan.x0.copy_(x.to(dev())); plan.noise.copy_(noise.to(dev())); plan.t.copy_(t.to(dev()))
    plan.ctx_in.t.copy_((cond.reshape(B * N, -1) if per_token else cond[0]).to(dev()))
    if L:
        plan.cross_in.t.copy_(cross.reshape(B * L, 512).to(dev()))
    params = dict(net.named_parameters())
    ref = g[name + ".grad_norms"]

    def verify(what):
        check(plan.losses, g[name + ".losses"], "%s losses (%s)" % (name, what))
        means = plan.parts.mean(dim=0).cpu()
        seen = 0
        for i, k in enumerate(_PART_KEYS):
            if name + "." + k in g.files:
                want = float(g[name + "." + k])
                assert abs(float(means[i]) - want) <= 1e-4 * max(1.0, abs(want)), (what, k, float(means[i]), want)
                seen += 1
        assert seen == (2 if name == "arrange" else 9)
        gn = np.array([float(flat.grad_view(params[k]).norm()) for k in names])
        e = np.abs(gn - ref) / np.maximum(ref, 1e-3 * ref.max())
        print("%s, %s: grad-norm rel err vs the reference's fp32 CPU gradients: max %.3g at %s" % (name, what, e.max(), names[int(e.argmax())]))
        assert e.max() < 1e-3, (what, names[int(e.argmax())], e.max())
        if L:
            want = float(g[name + ".d_cross_norm"])
            assert abs(float(plan.d_cross.norm()) - want) <= 1e-3 * want, (float(plan.d_cross.norm()), want)
        return flat.G.clone()

    flat.G.fill_(float("nan"))                      # every gradient the plan owns must be WRITTEN (alignment gaps stay NaN)
    flat.zero_head()
    plan.run_forward()
    plan.run_backward()
    g_eager = verify("plan, eager")
    flat.G.fill_(float("nan"))
    flat.zero_head()
    sg = _capture(plan, None, dev())
    assert sg is not None and len(sg.graphs) == 1
    sg.replay()
    torch.cuda.synchronize()
    g_graph = verify("plan, hipGraph replay")
    body = slice(flat.head_floats, None)
    same = (g_eager[body] == g_graph[body]) | (g_eager[body].isnan() & g_graph[body].isnan())
    assert bool(same.all()), "graph replay must reproduce the eager launches bit for bit"


@pytest.mark.parametrize("name", ["bedroom21", "text", "arrange"])
def test_reverse_step_at_full_batch(name, golden_dir, tmp_path):
    from diffuscene_amd.sampler import NoiseReplay
    g = np.load(os.path.join(golden_dir, "fullbatch.npz"))
    kw, x, t, cond, cross, noise, step_noise = fullbatch_inputs(name)
    net, diff = _model(name, tmp_path)
    d = dev()
    with torch.no_grad():
        x_t = diff.diffusion.q_sample(x.to(d), t.to(d), noise=noise.to(d))
        y = diff.diffusion.p_sample(diff._denoise, x_t, t.to(d), cond.to(d), cross.to(d) if cross is not None else None,
                                    noise_fn=NoiseReplay(step_noise[None].to(d)), clip_denoised=True)
    check(y[::16], g[name + ".p_sample.scenes16"], "%s reverse step, every 16th scene" % name)
    s, a = float(y.double().sum()), float(y.double().abs().sum())
    want_s, want_a = float(g[name + ".p_sample.sum"]), float(g[name + ".p_sample.abs_sum"])
    assert abs(a - want_a) <= 1e-5 * want_a, (a, want_a)
    assert abs(s - want_s) <= 1e-5 * want_a, (s, want_s)


@pytest.mark.parametrize("gemm_arith", ["split", "f32"], indirect=True)
def test_completion_loop_at_b128(golden_dir, tmp_path, gemm_arith):
    from diffuscene_amd.sampler import NoiseReplay
    g = np.load(os.path.join(golden_dir, "fullbatch.npz"))
    kw, x, t, cond, _, _, _ = fullbatch_inputs("complete")
    B, N, C = x.shape
    net, diff = _model("complete", tmp_path, time_num=COMPLETE_T)
    seq = complete_noise(B, N, C)
    main = torch.stack([seq[0]] + seq[2::2]).to(dev())
    part = torch.stack(seq[1::2]).to(dev())
    partial = x[:, :COMPLETE_P, :].contiguous().to(dev())
    res = []
    for graph in (False, True):
        with torch.no_grad():
            res.append(diff.complete_samples((B, N, C), dev(), condition=cond.to(dev()), noise_fn=NoiseReplay(main, part),
                                             clip_denoised=True, partial_boxes=partial, graph=graph))
        y = res[-1]
        check(y[::16], g["complete.scenes16"], "completion B=128 N=80 P=20 T=10 (graph=%s)" % graph)
        s, a = float(y.double().sum()), float(y.double().abs().sum())
        assert abs(a - float(g["complete.abs_sum"])) <= 1e-5 * float(g["complete.abs_sum"])
        assert abs(s - float(g["complete.sum"])) <= 1e-5 * float(g["complete.abs_sum"])
    assert torch.equal(res[0], res[1])
    assert torch.equal(res[1][:, :COMPLETE_P], partial)
