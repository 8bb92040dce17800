"""Deterministic synthetic weights and inputs (TEST INFRASTRUCTURE, see oracle/__init__.py).

The reference ships no checkpoints or fixtures for this path and
77.7 M parameters cannot be committed, so oracle, golden generator, tests and
bench all rebuild the *same* tensors from a seed.  Every tensor gets its own
generator seeded from crc32(name) so the values do not depend on enumeration
order or on which subset of tensors is requested.

``unet1d_param_spec`` restates the state_dict layout of the reference Unet1D
(scene_synthesis/networks/denoise_net.py:335-504; key names listed in SURVEY.md
section 3.4).  tests/test_oracle.py pins it against the key/shape list captured from
the real reference module (tests/golden/state_dict_keys.json).
"""
import math
import zlib
from collections import OrderedDict

import numpy as np
import torch

HIDDEN = 128  # heads(4) * dim_head(32), denoise_net.py:209-214


def _resblock(spec, p, d_in, d_out, emb):
    spec[p + "mlp.1.weight"] = (2 * d_out, emb)
    spec[p + "mlp.1.bias"] = (2 * d_out,)
    spec[p + "block1.proj.weight"] = (d_out, d_in, 1)
    spec[p + "block1.proj.bias"] = (d_out,)
    spec[p + "block1.norm.weight"] = (d_out,)
    spec[p + "block1.norm.bias"] = (d_out,)
    spec[p + "block2.proj.weight"] = (d_out, d_out, 1)
    spec[p + "block2.proj.bias"] = (d_out,)
    spec[p + "block2.norm.weight"] = (d_out,)
    spec[p + "block2.norm.bias"] = (d_out,)
    if d_in != d_out:
        spec[p + "res_conv.weight"] = (d_out, d_in, 1)
        spec[p + "res_conv.bias"] = (d_out,)


def _linattn(spec, p, d):
    spec[p + "fn.norm.g"] = (1, d, 1)
    spec[p + "fn.fn.to_qkv.weight"] = (3 * HIDDEN, d, 1)
    spec[p + "fn.fn.to_out.0.weight"] = (d, HIDDEN, 1)
    spec[p + "fn.fn.to_out.0.bias"] = (d,)
    spec[p + "fn.fn.to_out.1.g"] = (1, d, 1)


def _crossattn(spec, p, d, text_dim):
    spec[p + "fn.norm.g"] = (1, d, 1)
    spec[p + "fn.fn.to_q.weight"] = (HIDDEN, d, 1)
    spec[p + "fn.fn.to_kv.weight"] = (2 * HIDDEN, text_dim, 1)
    spec[p + "fn.fn.to_out.0.weight"] = (d, HIDDEN, 1)
    spec[p + "fn.fn.to_out.0.bias"] = (d,)
    spec[p + "fn.fn.to_out.1.g"] = (1, d, 1)


def _mlp3(spec, p, sizes):
    for idx, (i, o) in zip((0, 2, 4), sizes):
        spec[p + "%d.weight" % idx] = (o, i, 1)
        spec[p + "%d.bias" % idx] = (o,)


def unet1d_param_spec(dim=512, dim_mults=(1, 1, 1, 1), channels=62, seperate_all=False,
                      objectness_dim=1, class_dim=21, translation_dim=3, size_dim=3,
                      angle_dim=1, objfeat_dim=0, context_dim=256, instanclass_dim=0,
                      text_condition=False, text_dim=256, **_ignored):
    """name -> shape for every tensor of the reference Unet1D.state_dict()."""
    assert all(m == 1 for m in dim_mults), "only the shipped dim_mults=[1,1,1,1] layout is restated"
    d = dim
    bbox = translation_dim + size_dim + angle_dim
    time_dim = 4 * d
    ctx = context_dim + instanclass_dim
    spec = OrderedDict()
    if seperate_all:
        if objectness_dim > 0:
            _mlp3(spec, "objectness_embedf.", [(objectness_dim, d), (d, 2 * d), (2 * d, d)])
        if objfeat_dim > 0:
            _mlp3(spec, "objfeat_embedf.", [(objfeat_dim, d), (d, 2 * d), (2 * d, d)])
        _mlp3(spec, "class_embedf.", [(class_dim, d), (d, 2 * d), (2 * d, d)])
        _mlp3(spec, "bbox_embedf.", [(bbox, d), (d, 2 * d), (2 * d, d)])
        in_ch = d
    else:
        in_ch = channels
    spec["init_conv.weight"] = (d, in_ch, 1)
    spec["init_conv.bias"] = (d,)
    spec["time_mlp.1.weight"] = (time_dim, d)
    spec["time_mlp.1.bias"] = (time_dim,)
    spec["time_mlp.3.weight"] = (time_dim, time_dim)
    spec["time_mlp.3.bias"] = (time_dim,)
    n_res = len(dim_mults)
    for i in range(n_res):
        p = "downs.%d." % i
        _resblock(spec, p + "0.", d, d, ctx)
        _resblock(spec, p + "1.", d, d, time_dim)
        if text_condition:
            _crossattn(spec, p + "2.", d, text_dim)
        _resblock(spec, p + "3.", d, d, time_dim)
        _linattn(spec, p + "4.", d)
        if i == n_res - 1:
            spec[p + "5.weight"] = (d, d, 1)
            spec[p + "5.bias"] = (d,)
    _resblock(spec, "mid_block0.", d, d, ctx)
    _resblock(spec, "mid_block1.", d, d, time_dim)
    if text_condition:
        _crossattn(spec, "mid_attn_cross.", d, text_dim)
    spec["mid_attn.fn.norm.g"] = (1, d, 1)
    spec["mid_attn.fn.fn.to_qkv.weight"] = (3 * HIDDEN, d, 1)
    spec["mid_attn.fn.fn.to_out.weight"] = (d, HIDDEN, 1)
    spec["mid_attn.fn.fn.to_out.bias"] = (d,)
    _resblock(spec, "mid_block2.", d, d, time_dim)
    for i in range(n_res):
        p = "ups.%d." % i
        _resblock(spec, p + "0.", d, d, ctx)
        _resblock(spec, p + "1.", 2 * d, d, time_dim)
        if text_condition:
            _crossattn(spec, p + "2.", d, text_dim)
        _resblock(spec, p + "3.", 2 * d, d, time_dim)
        _linattn(spec, p + "4.", d)
        if i == n_res - 1:
            spec[p + "5.weight"] = (d, d, 1)
            spec[p + "5.bias"] = (d,)
    _resblock(spec, "final_res_block.", 2 * d, d, time_dim)
    if seperate_all:
        if objectness_dim > 0:
            _mlp3(spec, "objectness_hidden2output.", [(d, 2 * d), (2 * d, d), (d, objectness_dim)])
        if objfeat_dim > 0:
            _mlp3(spec, "objfeat_hidden2output.", [(d, 2 * d), (2 * d, d), (d, objfeat_dim)])
        _mlp3(spec, "class_hidden2output.", [(d, 2 * d), (2 * d, d), (d, class_dim)])
        _mlp3(spec, "bbox_hidden2output.", [(d, 2 * d), (2 * d, d), (d, bbox)])
    else:
        spec["final_conv.weight"] = (channels, d, 1)
        spec["final_conv.bias"] = (channels,)
    return spec


def _gen(name, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    return g


def synth_tensor(name, shape, seed=0):
    """One deterministic fp32 tensor.  Matrices: U(-b, b), b = 1/sqrt(fan_in) (the
    scale of PyTorch's default init); affine gains 1+U(-.1,.1); biases U(-.05,.05)."""
    g = _gen(name, seed)
    u = torch.rand(shape, generator=g, dtype=torch.float32) * 2.0 - 1.0
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "g" or (leaf == "weight" and len(shape) == 1):
        return 1.0 + 0.1 * u
    if leaf == "bias":
        return 0.05 * u
    fan_in = int(np.prod(shape[1:]))
    return u * (1.0 / math.sqrt(fan_in))


def synth_state_dict(net_kwargs, seed=0, prefix=""):
    spec = unet1d_param_spec(**net_kwargs)
    return OrderedDict((prefix + k, synth_tensor(k, s, seed)) for k, s in spec.items())


# ----------------------------------------------------------------------------------------
# synthetic scene batches with the value distribution of the real encoders (SURVEY.md 8d;
# reference datasets/threed_front_dataset.py:377-382,500-507,906-921)
# ----------------------------------------------------------------------------------------

from diffuscene_amd.workloads import (DATASET_STATS, REARRANGE_LIVING, TEXT_BEDROOM, UNCOND_BEDROOM, UNCOND_LIVING,  # noqa: E402,F401
                                      synth_scene_batch)


def synth_condition(B, N, dim=128, seed=0, shared=True):
    """Instance condition: randn(N, dim) broadcast over B (diffusion_scene_layout_ddpm.py:88-93,172-175)."""
    g = _gen("positional_embedding", seed)
    e = torch.randn(N, dim, generator=g)
    if shared:
        return e[None].expand(B, N, dim)
    g2 = _gen("condition_unshared", seed)
    return torch.randn(B, N, dim, generator=g2)


def synth_text_condition(B, L=32, dim=512, seed=0):
    g = _gen("condition_cross", seed)
    return torch.randn(B, L, dim, generator=g) * 0.5


def synth_noise(shape, seed, tag="noise"):
    g = _gen(tag, seed)
    return torch.randn(*shape, generator=g)


def synth_module_state(module, seed=0):
    """Deterministic, well-conditioned values for every parameter / buffer of ``module`` (used for the FoldingNet auto-encoder,
    whose reference and HIP implementations share parameter names): conv / linear weights ~ N(0, 1/fan_in), biases ~ 0.05 N,
    BatchNorm gain 1 + 0.1 N, shift 0.1 N, running_var in [1, 1.1); keyed by the parameter NAME, not by creation order."""
    sd = {}
    for name, t in module.state_dict().items():
        g = _gen("ae." + name, seed)
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros_like(t)
        elif name.endswith("running_var"):
            sd[name] = 1.0 + 0.1 * torch.rand(t.shape, generator=g)
        elif name.endswith("running_mean"):
            sd[name] = 0.05 * torch.randn(t.shape, generator=g)
        elif t.dim() >= 2:
            fan_in = t.shape[1]
            sd[name] = torch.randn(t.shape, generator=g) / (fan_in ** 0.5)
        elif ".bn" in name or name.split(".")[-2].startswith("bn") or "layers.1." in name or "layers.4." in name:
            sd[name] = (1.0 + 0.1 * torch.randn(t.shape, generator=g)) if name.endswith("weight") else 0.1 * torch.randn(t.shape, generator=g)
        else:
            sd[name] = 0.05 * torch.randn(t.shape, generator=g)
    return sd


def synth_point_clouds(B, N, seed=0):
    """(B, N, 3) clouds: points on noisy boxes of different extents (shape-like, not isotropic noise)."""
    g = _gen("ae.points", seed)
    ext = 0.15 + 0.3 * torch.rand(B, 1, 3, generator=g)
    p = (torch.rand(B, N, 3, generator=g) - 0.5) * 2.0
    face = torch.randint(0, 3, (B, N), generator=g)
    sign = torch.sign(torch.rand(B, N, generator=g) - 0.5)
    p.scatter_(2, face[..., None], sign[..., None])
    return (p * ext + 0.01 * torch.randn(B, N, 3, generator=g)).contiguous()
