"""Golden outputs of the REAL reference WRAPPER (build container only; TEST INFRASTRUCTURE -- imported by tests/ only).

    python -m oracle.make_golden_wrapper   ->  tests/golden/wrapper.npz (+ wrapper_keys.json)        (about 2 minutes of CPU)

SURVEY.md 8a row a21: ``DiffusionSceneLayout_DDPM.get_loss`` / ``.sample`` / ``.generate_layout`` / ``.complete_scene`` /
``.arrange_scene`` and ``train_on_batch`` of scene_synthesis/networks/diffusion_scene_layout_ddpm.py:131-226,228-347,456-473 --
the target / condition assembly (attribute cat, instance embedding broadcast, partial mask, arrange slices, text projection) in
front of the diffusion calls.  The reference's own class is built from the reference's shipped YAML ``network`` sections
(tests/golden/reference_configs.json, parsed from /root/reference/config by tests/golden/make_config_fixture.py):

  uncond    config/uncond/diffusion_bedrooms_instancond_lat32_v.yaml
  arrange   config/rearrange/diffusion_bedrooms_instancond_lat32_v_rearrange.yaml   (room_arrange_condition, 5 diffused channels)
  partial   uncond + room_partial_condition (partial_num_points 3, partial_emb_dim 64, instanclass_dim 192): no shipped YAML sets it,
            the code path exists (:107-117,193-199,262-267)
  text      config/text/diffusion_bedrooms_instancond_lat32_v_bert.yaml with BertTokenizer / BertModel REPLACED by seeded stand-ins
            (bert-base-cased is a download; SURVEY 8c: BERT features are "parity unpinned", the path is pinned from the encoder's
            last_hidden_state onwards -- the stand-in returns fake_bert_features(text))
  glove     the text config with text_glove_embedding (fc_text_f = Linear(50, 512) on ``desc_emb``)
  fixedinst uncond with learnable_embedding false: the instance condition is fc_instance_condition(one-hot slot index) (:97-104,172-175)
            instead of the learned positional_embedding every shipped YAML uses
  noinst    uncond with instance_condition false: NO conditioning at all (condition = None, :178-190); the context ResnetBlocks keep
            a Linear(0, 1024) the forward never applies (denoise_net.py:166-176,190-193) and whose .grad stays None

RNG: the reference draws t (torch.randint) and the noise (torch.randn) from torch's global CPU generator; each call below is made
right after ``torch.manual_seed(seed)``.  The GPU tests patch torch.randint / torch.randn to draw from the same CPU generator and
move the result to the device, so identical draws in identical ORDER are part of what is pinned.
Per case, stored: the target / condition / condition_cross handed to ``get_loss_iter``, loss + logged terms of ``get_loss``,
``train_on_batch`` (loss, gradient norm, parameter deltas of 12 parameters after one Adam step), raw ``sample`` outputs at B=4 with
T=20, the post-filtered dicts of generate_layout / complete_scene / arrange_scene at batch_size 1, and (uncond) the dict of dicts of
generate_layout_progressive.
"""
import contextlib
import copy
import io
import json
import os
import sys
import tempfile
import zlib

import numpy as np
import torch

from . import weights as W
from .make_golden import GOLDEN
from .ref_loader import load_reference_package

B, N, L_TEXT, SAMPLE_T, PARTIAL_P = 4, 12, 7, 20, 3
CASES = ("uncond", "arrange", "partial", "text", "glove", "fixedinst", "noinst")
_YAML = {"uncond": "uncond/diffusion_bedrooms_instancond_lat32_v.yaml",
         "arrange": "rearrange/diffusion_bedrooms_instancond_lat32_v_rearrange.yaml",
         "partial": "uncond/diffusion_bedrooms_instancond_lat32_v.yaml",
         "text": "text/diffusion_bedrooms_instancond_lat32_v_bert.yaml",
         "glove": "text/diffusion_bedrooms_instancond_lat32_v_bert.yaml",
         "fixedinst": "uncond/diffusion_bedrooms_instancond_lat32_v.yaml",
         "noinst": "uncond/diffusion_bedrooms_instancond_lat32_v.yaml"}
SEED_LOSS, SEED_TRAIN, SEED_SAMPLE, SEED_ONE = 1234, 1235, 1236, 1237
DELTA_PARAMS = 12


def network_config(case, stats_file, time_num=1000):
    """The ``network`` section a user of the reference has for ``case`` (deep copy, train_stats_file -> the synthetic stats)."""
    with open(os.path.join(GOLDEN, "reference_configs.json")) as f:
        cfg = copy.deepcopy(json.load(f)[_YAML[case]]["network"])
    cfg["diffusion_kwargs"]["train_stats_file"] = stats_file
    cfg["diffusion_kwargs"]["time_num"] = time_num
    if case == "partial":
        cfg.update(room_partial_condition=True, partial_num_points=PARTIAL_P, partial_emb_dim=64)
        cfg["net_kwargs"]["instanclass_dim"] = 128 + 64
    if case == "glove":
        cfg["text_glove_embedding"] = True
    if case == "fixedinst":
        cfg["learnable_embedding"] = False
    if case == "noinst":
        cfg["instance_condition"] = False
        cfg["net_kwargs"]["instanclass_dim"] = 0
    return cfg


def texts():
    return ["The room has a bed, two nightstands and a wardrobe .", "A desk is next to the bed .",
            "There is a double bed with a pendant lamp above it .", "The room has a wardrobe and a chair ."][:B]


def fake_bert_features(text_list):
    """Stand-in for BertModel(**tokenizer(texts)).last_hidden_state: (B, L_TEXT, 768), a function of each string alone."""
    return torch.stack([W.synth_noise((L_TEXT, 768), zlib.crc32(s.encode()) & 0xFFFF, "fake_bert") * 0.5 for s in text_list])


def glove_features():
    return W.synth_noise((B, L_TEXT, 50), 90, "glove") * 0.5


def wrapper_batch(case):
    """sample_params of the training loader (threed_front_dataset.py:888-935 keys) for B scenes of N objects."""
    nc = 22
    x = W.synth_scene_batch(B, N, nc, 32, seed=90)
    s = {"translations": x[:, :, 0:3].contiguous(), "sizes": x[:, :, 3:6].contiguous(), "angles": x[:, :, 6:8].contiguous(),
         "class_labels": x[:, :, 8:8 + nc].contiguous(), "objfeats_32": x[:, :, 8 + nc:].contiguous(),
         "room_layout": torch.zeros(B, 1, 64, 64)}
    if case == "text":
        s["description"] = texts()
    if case == "glove":
        s["desc_emb"] = glove_features()
    return s, x


def wrapper_state_dict(module):
    """Seeded values for every entry of ``module.state_dict()`` (the denoiser from oracle/weights.synth_state_dict, the wrapper-level
    parameters -- positional_embedding, fc_text_f, fc_partial_condition, fc_arrange_condition -- by name)."""
    sd = {}
    for k, v in module.state_dict().items():
        if v.numel() == 0:                          # the Linear(0, 1024) of an un-conditioned context block
            sd[k] = torch.zeros(tuple(v.shape))
        elif k == "positional_embedding":
            sd[k] = W.synth_noise(tuple(v.shape), 90, "wrapper_positional_embedding")
        else:
            sd[k] = W.synth_tensor(k[len("diffusion.model."):] if k.startswith("diffusion.model.") else k, tuple(v.shape), 0)
    return sd


def delta_param_names(all_names):
    idx = np.linspace(0, len(all_names) - 1, DELTA_PARAMS).round().astype(int)
    return [all_names[i] for i in idx]


class _FakeTokens(dict):
    def to(self, device):
        return self


class _FakeTokenizer:
    @classmethod
    def from_pretrained(cls, name):
        return cls()

    def __call__(self, text, return_tensors=None, padding=None):
        return _FakeTokens(texts=list(text))


class _FakeBert(torch.nn.Module):
    @classmethod
    def from_pretrained(cls, name):
        return cls()

    def forward(self, texts=None):
        import types
        return types.SimpleNamespace(last_hidden_state=fake_bert_features(texts))


def build_reference_wrapper(case, stats_file, time_num=1000):
    mod = load_reference_package()["diffusion_scene_layout_ddpm"]
    mod.BertTokenizer, mod.BertModel = _FakeTokenizer, _FakeBert
    cfg = network_config(case, stats_file, time_num)
    with contextlib.redirect_stdout(io.StringIO()):
        m = mod.DiffusionSceneLayout_DDPM(cfg["class_dim"] + 1, None, cfg)
    m.load_state_dict(wrapper_state_dict(m), strict=True)
    return mod, m, cfg


def sample_text_arg(case):
    return texts() if case == "text" else glove_features() if case == "glove" else None


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    stats_file = os.path.join(tempfile.mkdtemp(), "dataset_stats.txt")
    with open(stats_file, "w") as f:
        json.dump(W.DATASET_STATS, f)
    out, keys = {}, {}
    quiet = contextlib.redirect_stdout(io.StringIO())
    for case in CASES:
        mod, m, cfg = build_reference_wrapper(case, stats_file)
        keys[case] = [[k, list(v.shape)] for k, v in m.state_dict().items()]
        s, x = wrapper_batch(case)
        # ---- get_loss: inputs of the diffusion call + loss terms -------------------------------------------------------------------
        seen = {}
        inner = m.diffusion.get_loss_iter

        def spy(data, noises=None, condition=None, condition_cross=None):
            seen.update(target=data.detach().clone(), condition=None if condition is None else condition.detach().clone(),
                        cross=None if condition_cross is None else condition_cross.detach().clone())
            return inner(data, noises=noises, condition=condition, condition_cross=condition_cross)
        m.diffusion.get_loss_iter = spy
        torch.manual_seed(SEED_LOSS)
        loss, parts = m.get_loss(s)
        m.diffusion.get_loss_iter = inner
        out[case + ".target"] = seen["target"].numpy()
        if seen["condition"] is not None:
            out[case + ".condition"] = seen["condition"].numpy()
        if seen["cross"] is not None:
            out[case + ".cross"] = seen["cross"].numpy()
        out[case + ".loss"] = np.float32(loss.item())
        for k, v in parts.items():
            out[case + ".part." + k] = np.float32(v.item())
        # ---- train_on_batch: one Adam step with the reference's own function -------------------------------------------------------
        names = delta_param_names([k for k, p in m.named_parameters() if p.requires_grad])
        keys[case + ".delta_params"] = names
        before = {k: p.detach().clone() for k, p in m.named_parameters() if k in names}
        opt = torch.optim.Adam(m.parameters(), lr=0.0002, weight_decay=0.0)       # networks/__init__.py:29-30 for optimizer "Adam"
        torch.manual_seed(SEED_TRAIN)
        ret = mod.train_on_batch(m, opt, s, {"training": {"max_grad_norm": 10}})
        out[case + ".train.loss"] = np.float32(ret)
        logger = sys.modules[mod.__name__.rsplit(".", 2)[0] + ".stats_logger"].StatsLogger.instance()
        out[case + ".train.gradnorm"] = np.float32(logger["gradnorm"]._value / logger["gradnorm"]._count)
        logger.clear()
        params = dict(m.named_parameters())
        out[case + ".train.delta_norms"] = np.array([float((params[k].detach() - before[k]).norm()) for k in names], dtype=np.float32)
        out[case + ".train.grad_norms"] = np.array([0.0 if params[k].grad is None else float(params[k].grad.norm()) for k in names], dtype=np.float32)
        print("%-8s loss %.6f  train %.6f  gradnorm %.4f  target %s condition %s" % (
            case, out[case + ".loss"], ret, out[case + ".train.gradnorm"], tuple(seen["target"].shape), None if seen["condition"] is None else tuple(seen["condition"].shape)))

        # ---- sampling (T = 20): raw samples at B = 4, post-filtered dicts at batch_size 1 -------------------------------------------
        mod, m, cfg = build_reference_wrapper(case, stats_file, time_num=SAMPLE_T)
        C = cfg["point_dim"]
        room = torch.zeros(B, 1, 64, 64)
        kw = {}
        if case == "arrange":
            kw["input_boxes"] = x
        if case == "partial":
            kw["partial_boxes"] = x[:, :PARTIAL_P].contiguous()
        text = sample_text_arg(case)
        torch.manual_seed(SEED_SAMPLE)
        with torch.no_grad(), quiet:
            y = m.sample(room, N, C, batch_size=B, text=text, clip_denoised=True, **kw)
        out[case + ".sample"] = y.numpy()
        one_text = None if text is None else text[:1]
        torch.manual_seed(SEED_ONE)
        with quiet:
            if case == "arrange":
                d = m.arrange_scene(room[:1], N, C, x[:1], batch_size=1, clip_denoised=True)
            elif case == "partial":
                d = m.complete_scene(room[:1], N, C, x[:1, :PARTIAL_P].contiguous(), batch_size=1, clip_denoised=True)
            else:
                d = m.generate_layout(room[:1], N, C, batch_size=1, text=one_text, clip_denoised=True)
        for k, v in d.items():
            out[case + ".layout." + k] = v.numpy()
        if case == "uncond":
            # completion WITHOUT a partial-condition MLP (what completion_rearrange.py does with the uncond checkpoints), B = 4
            torch.manual_seed(SEED_SAMPLE + 10)
            with torch.no_grad(), quiet:
                y = m.sample(room, N, C, batch_size=B, partial_boxes=x[:, :PARTIAL_P].contiguous(), clip_denoised=True)
            out["uncond.complete"] = y.numpy()
            torch.manual_seed(SEED_ONE + 10)
            with quiet:
                d = m.generate_layout(room[:1], N, C, batch_size=1, clip_denoised=False, keep_empty=True)
            for k, v in d.items():
                out["uncond.layout_noclip_keep." + k] = v.numpy()
            # the progressive entry point (:320-333): the trajectory of the reverse loop, every 5th step post-filtered on its own
            torch.manual_seed(SEED_ONE + 20)
            with quiet:
                traj = m.generate_layout_progressive(room[:1], N, C, batch_size=1, ret_traj=True, clip_denoised=True, num_step=5)
            keys["uncond.progressive_steps"] = sorted(int(k) for k in traj)
            for kt, dd in traj.items():
                for k, v in dd.items():
                    out["uncond.progressive.%d.%s" % (kt, k)] = v.numpy()
        print("%-8s sample |mean| %.5f, layout boxes kept %d of %d" % (case, float(y.abs().mean()), d["translations"].shape[1], N))
    with open(os.path.join(GOLDEN, "wrapper_keys.json"), "w") as f:
        json.dump(keys, f)
    np.savez_compressed(os.path.join(GOLDEN, "wrapper.npz"), **out)
    print("written", os.path.join(GOLDEN, "wrapper.npz"), len(out), "arrays")


if __name__ == "__main__":
    sys.exit(main())
