"""Scene-layout wrapper around the DDPM -- drop-in for
scene_synthesis/networks/diffusion_scene_layout_ddpm.py (``DiffusionSceneLayout_DDPM``, ``train_on_batch``,
``validate_on_batch``): same constructor, config keys, sample_params keys, method names, state_dict keys
(``diffusion.model.*``, ``positional_embedding``, ``fc_text_f.*``, ``fc_arrange_condition.*`` ...).

What changed underneath:
* the instance embedding is broadcast over the batch as a stride-0 view instead of a gathered (B,N,128)
  copy (:174-175), which lets the denoiser push N instead of B*N rows through its 9 context MLPs;
* ``train_on_batch`` reads all logged scalars with ONE device->host copy instead of 11 ``.item()`` syncs
  (:461-473), and all-reduces gradients over RCCL when torch.distributed is initialised (data parallel);
* post-filtering of empty boxes (:351-406) is vectorised; it still looks only at batch row 0, as the reference.
Optional third-party encoders (BERT / CLIP / floor-plan ResNet) are imported only when the config asks for them.
"""
import torch
import torch.nn as nn
from torch.nn import Module

from ..stats_logger import StatsLogger
from .denoise_net import Unet1D
from .diffusion_ddpm import DiffusionPoint


class _TwoLayer(nn.Sequential):
    """Linear(bias=False) -> LeakyReLU(0.1) -> Linear(bias=False): the instance / partial / arrange condition MLPs of the
    reference (:94-125).  Same parameters and state_dict keys (``0.weight``, ``2.weight``) as the reference's nn.Sequential;
    forward and backward run on the fp32 MFMA GEMM and the elementwise HIP kernels (autograd_ops), not on ATen / rocBLAS."""

    def __init__(self, n_in, n_out):
        super().__init__(nn.Linear(n_in, n_out, bias=False), nn.LeakyReLU(0.1, inplace=True),
                         nn.Linear(n_out, n_out, bias=False))

    def forward(self, x):
        from .._lib import ACT_LEAKY01
        from ..autograd_ops import ActFn, linear_any
        h = linear_any(x, self[0].weight)
        h = ActFn.apply(h.reshape(-1, h.shape[-1]).contiguous(), ACT_LEAKY01).reshape(h.shape)
        return linear_any(h, self[2].weight)


class _HipLinear(nn.Linear):
    """nn.Linear whose forward / backward run on the HIP GEMM (fc_text_f, fc_room_f)."""

    def forward(self, x):
        from ..autograd_ops import linear_any
        return linear_any(x, self.weight, self.bias)


def _two_layer(n_in, n_out):
    return _TwoLayer(n_in, n_out)


class DiffusionSceneLayout_DDPM(Module):

    def __init__(self, n_classes, feature_extractor, config):
        super().__init__()
        self.room_mask_condition = config.get("room_mask_condition", True)
        self.text_condition = config.get("text_condition", False)
        self.text_glove_embedding = config.get("text_glove_embedding", False)
        self.text_clip_embedding = config.get("text_clip_embedding", False)
        if self.room_mask_condition:
            self.feature_extractor = feature_extractor
            self.fc_room_f = _HipLinear(self.feature_extractor.feature_size, config["latent_dim"])
            print('use room mask as condition')
        elif self.text_condition:
            text_embed_dim = config.get("text_embed_dim", 512)
            if self.text_glove_embedding:
                self.fc_text_f = _HipLinear(50, text_embed_dim)
                print('use text as condition, and pretrained glove embedding')
            elif self.text_clip_embedding:
                import clip
                device = "cuda" if torch.cuda.is_available() else "cpu"
                self.clip_model, self.clip_preprocess = clip.load("ViT-B/32", device=device)
                for p in self.clip_model.parameters():
                    p.requires_grad = False
                print('use text as condition, and pretrained clip embedding')
            else:
                # ``text_bert_cached: true`` (new key, default false = reference behaviour): the frozen BERT encoder is not
                # instantiated; batches carry its cached last_hidden_state as ``desc_bert`` (B, L, 768), produced once
                # per description by diffuscene_amd.text_cache.BertFeatureCache
                self.text_bert_cached = config.get("text_bert_cached", False)
                if not self.text_bert_cached:
                    from transformers import BertModel, BertTokenizer
                    self.tokenizer = BertTokenizer.from_pretrained('bert-base-cased')
                    self.bertmodel = BertModel.from_pretrained("bert-base-cased")
                    for p in self.bertmodel.parameters():
                        p.requires_grad = False
                self.fc_text_f = _HipLinear(768, text_embed_dim)
                print('use text as condition, and pretrained bert model')
        else:
            print('NOT use room and text as condition')

        if config["net_type"] == "unet1d":
            denoise_net = Unet1D(**config["net_kwargs"])
        else:
            raise NotImplementedError()
        self.diffusion = DiffusionPoint(denoise_net=denoise_net, config=config, **config["diffusion_kwargs"])
        self.n_classes = n_classes
        self.config = config

        self.objectness_dim = config.get("objectness_dim", 1)
        self.class_dim = config.get("class_dim", 21)
        self.translation_dim = config.get("translation_dim", 3)
        self.size_dim = config.get("size_dim", 3)
        self.angle_dim = config.get("angle_dim", 1)
        self.bbox_dim = self.translation_dim + self.size_dim + self.angle_dim
        self.objfeat_dim = config.get("objfeat_dim", 0)

        self.learnable_embedding = config.get("learnable_embedding", False)
        self.instance_condition = config.get("instance_condition", False)
        self.sample_num_points = config.get("sample_num_points", 12)
        self.instance_emb_dim = config.get("instance_emb_dim", 64)
        if self.learnable_embedding:
            if self.instance_condition:
                self.register_parameter("positional_embedding",
                                        nn.Parameter(torch.randn(self.sample_num_points, self.instance_emb_dim)))
            else:
                self.instance_emb_dim = 0
        else:
            if self.instance_condition:
                self.fc_instance_condition = _two_layer(self.sample_num_points, self.instance_emb_dim)
            else:
                self.instance_emb_dim = 0

        self.room_partial_condition = config.get("room_partial_condition", False)
        self.partial_num_points = config.get("partial_num_points", 0)
        self.partial_emb_dim = config.get("partial_emb_dim", 64)
        full = self.bbox_dim + self.class_dim + self.objectness_dim + self.objfeat_dim
        if self.room_partial_condition:
            self.fc_partial_condition = _two_layer(full, self.partial_emb_dim)
        else:
            self.partial_emb_dim = 0
        self.room_arrange_condition = config.get("room_arrange_condition", False)
        self.arrange_emb_dim = config.get("arrange_emb_dim", 64)
        if self.room_arrange_condition:
            self.fc_arrange_condition = _two_layer(full - self.translation_dim - self.angle_dim, self.arrange_emb_dim)
        else:
            self.arrange_emb_dim = 0

    # ------------------------------------------------------------------------------------ conditions
    def _instance_condition(self, batch_size, device):
        if not self.instance_condition:
            return None
        if self.learnable_embedding:
            # same values as positional_embedding[arange(N)].repeat(B, 1, 1) (:174-175), without the copy
            return self.positional_embedding[None, :, :].expand(batch_size, -1, -1)
        eye = torch.eye(self.sample_num_points, device=device)
        return self.fc_instance_condition(eye)[None].expand(batch_size, -1, -1)

    def _base_condition(self, room_mask, batch_size, num_points, device):
        room_layout_f = self.fc_room_f(self.feature_extractor(room_mask)) if self.room_mask_condition else None
        inst = self._instance_condition(batch_size, device)
        if room_layout_f is not None and inst is not None:
            return torch.cat([room_layout_f[:, None, :].repeat(1, num_points, 1), inst], dim=-1).contiguous()
        if room_layout_f is not None:
            return room_layout_f[:, None, :].repeat(1, num_points, 1)
        return inst

    def attach_bert_cache(self, cache):
        """Use a text_cache.BertFeatureCache for the descriptions instead of running BERT inside every step."""
        object.__setattr__(self, "_bert_cache", cache)

    def _text_condition(self, text, desc_emb, device, desc_bert=None):
        if not self.text_condition:
            return None
        if self.text_glove_embedding:
            return self.fc_text_f(desc_emb)
        if self.text_clip_embedding:
            import clip
            return self.clip_model.encode_text(clip.tokenize(text).to(device))
        cache = getattr(self, "_bert_cache", None)
        if desc_bert is None and cache is not None:
            desc_bert = cache.batch(text, device)                # frozen encoder: features are a function of the text only
        if desc_bert is not None:
            return self.fc_text_f(desc_bert)
        if getattr(self, "text_bert_cached", False):
            raise KeyError("text_bert_cached: the batch must carry 'desc_bert' (B, L, 768) or a BertFeatureCache must be "
                           "attached (attach_bert_cache)")
        tokenized = self.tokenizer(text, return_tensors='pt', padding=True).to(device)
        return self.fc_text_f(self.bertmodel(**tokenized).last_hidden_state)

    def _arrange_input(self, boxes):
        tr, sz, bb = self.translation_dim, self.size_dim, self.bbox_dim
        return torch.cat([boxes[:, :, tr:tr + sz], boxes[:, :, bb:]], dim=-1).contiguous()

    # ------------------------------------------------------------------------------------ training
    def get_loss(self, sample_params):
        """reference :131-226"""
        target, condition, condition_cross = self._loss_inputs(sample_params)
        return self.diffusion.get_loss_iter(target, condition=condition, condition_cross=condition_cross)

    def _loss_inputs(self, sample_params):
        """The part of get_loss before the diffusion call (:131-221): diffusion target (B, N, C), per-object condition and
        cross-attention condition."""
        class_labels = sample_params["class_labels"]
        translations, sizes, angles = sample_params["translations"], sample_params["sizes"], sample_params["angles"]
        batch_size, num_points, _ = class_labels.shape
        device = class_labels.device
        full = self.bbox_dim + self.class_dim + self.objectness_dim + self.objfeat_dim
        packed = sample_params.get("_packed")      # diffuscene_amd.datasets batches arrive already in channel order
        if (self.config["point_dim"] == full and packed is not None and packed.shape[-1] == full
                and self.objectness_dim == 0 and packed.is_contiguous()):
            target = packed
        elif self.config["point_dim"] == full:
            parts = [translations, sizes, angles, class_labels]
            if self.objectness_dim > 0:
                parts.append(sample_params["objectness"])
            if self.objfeat_dim > 0:
                parts.append(sample_params["objfeats_32"] if self.objfeat_dim == 32 else sample_params["objfeats"])
            target = torch.cat(parts, dim=-1).contiguous()
        elif self.config["point_dim"] == self.bbox_dim:
            target = torch.cat([translations, sizes, angles], dim=-1).contiguous()
        else:
            raise NotImplementedError
        condition = self._base_condition(sample_params["room_layout"] if self.room_mask_condition else None,
                                         batch_size, num_points, device)
        if self.room_partial_condition:
            mask = torch.zeros((batch_size, num_points, 1), device=device)
            mask[:, :self.partial_num_points] = 1.0
            condition = torch.cat([condition, self.fc_partial_condition(target * mask)], dim=-1).contiguous()
        if self.room_arrange_condition:
            condition = torch.cat([condition, self.fc_arrange_condition(self._arrange_input(target))],
                                  dim=-1).contiguous()
            tr, sz, bb = self.translation_dim, self.size_dim, self.bbox_dim
            target = torch.cat([target[:, :, 0:tr], target[:, :, tr + sz:bb]], dim=-1).contiguous()
        condition_cross = self._text_condition(sample_params.get("description"), sample_params.get("desc_emb"), device,
                                               desc_bert=sample_params.get("desc_bert"))
        return target, condition, condition_cross

    # ------------------------------------------------------------------------------------ sampling
    def sample(self, room_mask, num_points, point_dim, batch_size=1, text=None, partial_boxes=None,
               input_boxes=None, ret_traj=False, ddim=False, clip_denoised=False, freq=40, batch_seeds=None):
        """reference :228-310"""
        device = room_mask.device
        noise = torch.randn((batch_size, num_points, point_dim))   # CPU draw kept: it advances the CPU RNG (:232)
        condition = self._base_condition(room_mask, room_mask.size(0), num_points, device)
        if self.room_partial_condition:
            zeros = torch.zeros((batch_size, num_points - partial_boxes.shape[1], partial_boxes.shape[2]),
                                device=device)
            cond_p = self.fc_partial_condition(torch.cat([partial_boxes, zeros], dim=1).contiguous())
            condition = torch.cat([condition, cond_p], dim=-1).contiguous()
        if self.room_arrange_condition:
            condition = torch.cat([condition, self.fc_arrange_condition(self._arrange_input(input_boxes))],
                                  dim=-1).contiguous()
        condition_cross = self._text_condition(text, text, device)
        if self.text_condition and not (self.text_glove_embedding or self.text_clip_embedding):
            print('after bert:', condition_cross.shape)
        if input_boxes is not None:
            print('scene arrangement sampling')
            return self.diffusion.arrange_samples(noise.shape, device, condition=condition,
                                                  condition_cross=condition_cross, clip_denoised=clip_denoised,
                                                  input_boxes=input_boxes)
        if partial_boxes is not None:
            print('scene completion sampling')
            return self.diffusion.complete_samples(noise.shape, device, condition=condition,
                                                   condition_cross=condition_cross, clip_denoised=clip_denoised,
                                                   partial_boxes=partial_boxes)
        print('unconditional / conditional generation sampling')
        if ret_traj:
            return self.diffusion.gen_sample_traj(noise.shape, device, freq=freq, condition=condition,
                                                  condition_cross=condition_cross, clip_denoised=clip_denoised)
        return self.diffusion.gen_samples(noise.shape, device, condition=condition, condition_cross=condition_cross,
                                          clip_denoised=clip_denoised)

    @torch.no_grad()
    def generate_layout(self, room_mask, num_points, point_dim, batch_size=1, text=None, ret_traj=False, ddim=False,
                        clip_denoised=False, batch_seeds=None, device="cpu", keep_empty=False):
        samples = self.sample(room_mask, num_points, point_dim, batch_size, text=text, ret_traj=ret_traj, ddim=ddim,
                              clip_denoised=clip_denoised, batch_seeds=batch_seeds)
        return self.delete_empty_from_network_samples(samples, device=device, keep_empty=keep_empty)

    @torch.no_grad()
    def generate_layout_progressive(self, room_mask, num_points, point_dim, batch_size=1, text=None, ret_traj=False,
                                    ddim=False, clip_denoised=False, batch_seeds=None, device="cpu", keep_empty=False,
                                    num_step=100):
        traj = self.sample(room_mask, num_points, point_dim, batch_size, text=text, ret_traj=ret_traj, ddim=ddim,
                           clip_denoised=clip_denoised, batch_seeds=batch_seeds, freq=num_step)[1:]
        return {num_step * i: self.delete_empty_from_network_samples(s, device=device, keep_empty=keep_empty)
                for i, s in enumerate(traj)}

    @torch.no_grad()
    def complete_scene(self, room_mask, num_points, point_dim, partial_boxes, batch_size=1, ret_traj=False, ddim=False,
                       clip_denoised=False, batch_seeds=None, device="cpu", keep_empty=False):
        samples = self.sample(room_mask, num_points, point_dim, batch_size, partial_boxes=partial_boxes,
                              ret_traj=ret_traj, ddim=ddim, clip_denoised=clip_denoised, batch_seeds=batch_seeds)
        return self.delete_empty_from_network_samples(samples, device=device, keep_empty=keep_empty)

    @torch.no_grad()
    def arrange_scene(self, room_mask, num_points, point_dim, input_boxes, batch_size=1, ret_traj=False, ddim=False,
                      clip_denoised=False, batch_seeds=None, device="cpu", keep_empty=False):
        samples = self.sample(room_mask, num_points, point_dim, batch_size, input_boxes=input_boxes, ret_traj=ret_traj,
                              ddim=ddim, clip_denoised=clip_denoised, batch_seeds=batch_seeds)
        return self.delete_empty_from_network_samples(samples, device=device, keep_empty=keep_empty)

    # ------------------------------------------------------------------------------------ post-filter
    def _keep_rows(self, empty_logit_row0, keep_empty):
        """Rows kept by the reference loop (:377-380, :424-427): drop slot i when the 'empty' logit of BATCH ROW 0
        is >= 0 (network samples) -- the decision is shared by the whole batch, as in the reference."""
        n = empty_logit_row0.shape[0]
        if keep_empty:
            return torch.arange(n)
        return torch.nonzero(~empty_logit_row0, as_tuple=False).flatten()

    def _split_boxes(self, rows):
        """(B, K, C) kept rows -> the reference's output dict (raw class scores, :383-386; CPU tensors, :390-406)."""
        tr, sz, bb, nc = self.translation_dim, self.size_dim, self.bbox_dim, self.class_dim
        out = {
            "class_labels": rows[:, :, bb:bb + nc - 1].contiguous(),
            "translations": rows[:, :, 0:tr].contiguous(),
            "sizes": rows[:, :, tr:tr + sz].contiguous(),
            "angles": rows[:, :, tr + sz:bb].contiguous(),
        }
        if self.objfeat_dim > 0:
            out["objfeats"] = rows[:, :, bb + nc:bb + nc + self.objfeat_dim].contiguous()
        return out

    @torch.no_grad()
    def delete_empty_from_network_samples(self, samples, device="cpu", keep_empty=False):
        """Reference :351-406.  Its loop takes the keep / drop decision of every slot from BATCH ROW 0 (:379) (and only runs
        for batch_size 1: its accumulators are (1, 0, .) tensors); here that decision is applied to the whole batch, so
        B = 1 is exactly the reference and B > 1 returns equally long scenes.  Samples on a HIP device are compacted there
        (dsc_postfilter_compact_f32, one launch) and cross to the host once; per-scene filtering of a batch is
        ``delete_empty_per_scene``."""
        samples = samples.detach()
        bb, nc = self.bbox_dim, self.class_dim
        if samples.is_cuda and samples.dtype == torch.float32 and samples.shape[1] <= 192:
            from .. import ops
            packed, counts = ops.postfilter_compact(samples.contiguous(), bb + nc - 1, per_scene=False, keep_empty=keep_empty)
            k = int(counts[0].item())
            return self._split_boxes(packed[:, :k].to("cpu"))
        samples = samples.to("cpu")
        keep = self._keep_rows(samples[0, :, bb + nc - 1] >= 0, keep_empty)
        return self._split_boxes(samples[:, keep])

    @torch.no_grad()
    def delete_empty_per_scene(self, samples, keep_empty=False):
        """Batched generation: every scene of ``samples`` (B, N, C) is filtered by ITS OWN 'empty' logits, i.e. the reference
        method applied to each scene alone.  One device launch + one device->host copy; returns a list of B dicts (the
        reference's keys, leading dimension 1)."""
        from .. import ops
        samples = samples.detach().contiguous()
        packed, counts = ops.postfilter_compact(samples, self.bbox_dim + self.class_dim - 1, per_scene=True,
                                                keep_empty=keep_empty)
        packed, counts = packed.to("cpu"), counts.to("cpu").tolist()
        return [self._split_boxes(packed[b:b + 1, :counts[b]]) for b in range(samples.shape[0])]

    @torch.no_grad()
    def generate_layout_batched(self, room_mask, num_points, point_dim, batch_size, text=None, clip_denoised=False,
                                batch_seeds=None, keep_empty=False):
        """``generate_layout`` for a whole batch: one reverse loop for ``batch_size`` scenes, each post-filtered on its own."""
        samples = self.sample(room_mask, num_points, point_dim, batch_size, text=text, clip_denoised=clip_denoised,
                              batch_seeds=batch_seeds)
        return self.delete_empty_per_scene(samples, keep_empty=keep_empty)

    @torch.no_grad()
    def delete_empty_boxes(self, samples_dict, device="cpu", keep_empty=False):
        cl = samples_dict["class_labels"].detach().to("cpu")
        keep = self._keep_rows(cl[0, :, -1] > 0, keep_empty)
        out = {"class_labels": cl[:, keep, :self.class_dim - 1].contiguous()}
        for k in ("translations", "sizes", "angles") + (("objfeats",) if self.objfeat_dim > 0 else ()):
            out[k] = samples_dict[k].detach().to("cpu")[:, keep, :].contiguous()
        return out


def train_on_batch(model, optimizer, sample_params, config):
    """reference :456-473: zero_grad, loss, backward, clip_grad_norm_(max_grad_norm), optimizer step.

    Default path: the static training plan (train_step.py / train_plan.py) -- forward, loss and backward are one hipGraph
    replay that leaves every gradient in the flat buffer G; under torch.distributed the buckets of G are all-reduced over
    RCCL while the backward is still running, so every rank clips and steps identically.  Configurations the plan does not
    cover run the same HIP kernels under torch.autograd (autograd_ops.py).  All logged scalars are fetched with one
    device->host copy."""
    from ..ddp import average_gradients, clip_grad_norm_fused, overlapped_reducer
    from ..train_step import loss_step, plan_supported
    if plan_supported(model):
        # optimizer.zero_grad(): every gradient is OVERWRITTEN by the plan, so nothing is zeroed or freed; only a pending
        # deferred clip coefficient is cancelled (FusedAdam)
        if hasattr(optimizer, "cancel_pending_clip"):
            optimizer.cancel_pending_clip()
        loss, loss_dict, _ = loss_step(model, sample_params, backward=True)
    else:
        optimizer.zero_grad()
        reducer = overlapped_reducer(model)      # None on one GPU; hooks launch bucket all-reduces during backward
        loss, loss_dict = model.get_loss(sample_params)
        loss.backward()
        if reducer is not None:
            reducer.finish()
        else:
            average_gradients(model)             # no-op on one GPU; DSC_DDP_OVERLAP=0 selects this post-backward form
    if hasattr(optimizer, "clip_grad_norm_"):         # FusedAdam: norm + coefficient on the device, applied in step()
        grad_norm = optimizer.clip_grad_norm_(config["training"]["max_grad_norm"])
    else:
        grad_norm = clip_grad_norm_fused(model.parameters(), config["training"]["max_grad_norm"])
    keys = list(loss_dict.keys())
    packed_t = torch.stack([loss.detach(), grad_norm.detach()] + [loss_dict[k].detach() for k in keys])
    lr = optimizer.param_groups[0]['lr']
    if packed_t.is_cuda:
        # The logged scalars cross to the host ASYNCHRONOUSLY (pinned buffer + event) and the optimizer step is enqueued before the
        # host waits for them: the device runs the Adam sweep while the copy lands and the host gets on with the next batch.  (The
        # reference reads 11 .item() values, then steps: the values are the same -- everything logged is computed before the step --
        # but a blocking read in front of step() left the GPU idle for ~0.25 ms per step, profiles/r03_train_kernel_trace.txt.)
        host = getattr(model, "_dsc_scalars_host", None)
        if host is None or host.numel() != packed_t.numel():
            host = torch.empty(packed_t.numel(), dtype=torch.float32).pin_memory()
            object.__setattr__(model, "_dsc_scalars_host", host)
        host.copy_(packed_t, non_blocking=True)
        landed = torch.cuda.Event()
        landed.record(torch.cuda.current_stream(packed_t.device))
        optimizer.step()
        landed.synchronize()
        packed = host.tolist()
    else:
        packed = packed_t.tolist()
        optimizer.step()
    logger = StatsLogger.instance()
    for k, v in zip(keys, packed[2:]):
        logger[k].value = v
    logger["gradnorm"].value = packed[1]
    logger["lr"].value = lr
    from .._lib import check_indices
    check_indices("train_on_batch")            # DSC_CHECK_INDICES=1: a clamped (out-of-range) device timestep is an error, as in the reference
    return packed[0]


@torch.no_grad()
def validate_on_batch(model, sample_params, config):
    from ..train_step import loss_step, plan_supported
    if plan_supported(model):
        loss, loss_dict, _ = loss_step(model, sample_params, backward=False)
    else:
        loss, loss_dict = model.get_loss(sample_params)
    keys = list(loss_dict.keys())
    packed = torch.stack([loss.detach()] + [loss_dict[k].detach() for k in keys]).tolist()
    for k, v in zip(keys, packed[1:]):
        StatsLogger.instance()[k].value = v
    return packed[0]
