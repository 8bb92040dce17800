"""One DDPM training / validation step of ``DiffusionSceneLayout_DDPM`` on the static training plan (train_plan.py).

``train_on_batch`` (reference diffusion_scene_layout_ddpm.py:456-473) keeps its semantics -- zero_grad, loss, backward,
clip_grad_norm_(max_grad_norm), optimizer step, the 11 logged scalars -- but the work between the RNG draws and the optimizer
is ONE hipGraph replay on a single GPU:

    host: build target / conditioning (reference :131-226), draw t and the noise in the reference's RNG order
    graph: q_sample -> Unet1D forward -> p_losses (+ d loss / d out) -> backward of every layer -> gradients in flat G
    host: (only for conditioning computed by torch modules: fc_text_f, fc_arrange_condition ...) autograd through them
    data parallel: buckets of G are all-reduced over RCCL as the backward finishes them (no graph; launches are eager)
    FusedAdam: gradient norm + clip coefficient + Adam sweep on the device; one device->host copy for the logged scalars
"""
import os

import torch

from ._lib import SS_NONE, SS_PER_SLOT, SS_PER_TOKEN
from .flat import ensure_flat

_PART_KEYS = ('loss.bbox', 'loss.trans', 'loss.size', 'loss.angle', 'loss.class', 'loss.object', 'loss.objfeat',
              'loss.liou', 'loss.bbox_iou')


def plan_supported(model):
    """The static plan covers every shipped training configuration; anything else keeps the autograd path (same kernels)."""
    if os.environ.get("DSC_TRAIN_PLAN", "1") == "0":
        return False
    d = model.diffusion.diffusion
    net = model.diffusion.model
    p = next(model.parameters())
    if not p.is_cuda or model.room_mask_condition or model.room_partial_condition:
        return False
    if d.loss_type != 'mse' or d.translation_dim != 3 or model.sample_num_points > 160:
        return False
    full = model.bbox_dim + model.class_dim + model.objectness_dim + model.objfeat_dim
    if model.room_arrange_condition:
        return net.channels == model.translation_dim + model.angle_dim
    return d.size_dim == 3 and model.config["point_dim"] == full and net.channels == full


class PlanRunner:
    """Plans + captured graphs of one model (one per batch signature)."""

    def __init__(self, model):
        self.model = model
        self.flat = ensure_flat(model)
        self.plans = {}
        self.synced = False

    def world(self):
        import torch.distributed as dist
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def distributed(self):
        """Reducer active: more than one rank (DSC_DDP_FORCE=1 also drives it with a single-rank process group, for tests)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size() > 1 or os.environ.get("DSC_DDP_FORCE", "0") == "1"

    def plan_for(self, B, N, ctx_mode, ctx_dim, L, text_dim, ctx_param):
        from .train_plan import HipBackend, TrainPlan
        ws = self.world()
        key = (B, N, ctx_mode, ctx_dim, L, text_dim, ctx_param is not None, ws, self.distributed())
        ent = self.plans.get(key)
        if ent is None:
            model = self.model
            dev = self.flat.device
            plan = TrainPlan(model.diffusion.model, self.flat, model.diffusion.diffusion, B, N, ctx_mode, ctx_dim, L,
                             text_dim, HipBackend(dev), per_block_grads=self.distributed(), ctx_param=ctx_param,
                             grad_scale=1.0 / (B * ws))
            ent = {"plan": plan, "graph": None, "reducer": None, "warm": 0}
            if self.distributed():
                from .ddp import FlatGradientReducer
                ent["reducer"] = FlatGradientReducer(self.flat, plan)
            while len(self.plans) >= 3:             # a plan owns all its activations (~20 GB at B=256, N=80): keep a few
                self.plans.pop(next(iter(self.plans)))
            self.plans[key] = ent
        return ent

    def run(self, ent, backward):
        plan = ent["plan"]
        if not backward:
            plan.run_forward()
            return
        if ent["reducer"] is not None:
            plan.run_forward()
            plan.run_backward(on_progress=ent["reducer"].on_progress)
            return
        if os.environ.get("DSC_TRAIN_GRAPH", "1") == "0":
            plan.run_forward()
            plan.run_backward()
            return
        if ent["graph"] is None:
            if ent["warm"] < 1:
                # first step eagerly: loads every code object and surfaces launch errors with a Python stack
                ent["warm"] += 1
                plan.run_forward()
                plan.run_backward()
                return
            dev = self.flat.device
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                plan.run_forward()
                plan.run_backward()
            ent["graph"] = g
            torch.cuda.current_stream(dev).synchronize()
        ent["graph"].replay()


def _runner(model):
    r = getattr(model, "_dsc_plan_runner", None)
    if r is None or not r.flat.valid() or r.flat is not getattr(model, "_dsc_flat", None):
        r = PlanRunner(model)
        object.__setattr__(model, "_dsc_plan_runner", r)
    r.flat.attach_grads()
    return r


def loss_step(model, sample_params, backward):
    """-> (loss 0-d device tensor, loss_dict of 0-d device tensors, runner entry).  With ``backward`` the gradients of
    every trainable parameter are in the flat buffer G afterwards (averaged over the ranks' batches when distributed)."""
    r = _runner(model)
    fl = r.flat
    ws = r.world()
    if ws > 1 and not r.synced:
        from .ddp import broadcast_parameters
        broadcast_parameters(model)
        r.synced = True
    grad_ctx = torch.enable_grad() if backward else torch.no_grad()
    with grad_ctx:
        target, condition, condition_cross = model._loss_inputs(sample_params)
    B, N, C = target.shape
    device = target.device
    # RNG draws of get_loss_iter / p_losses (diffusion_ddpm.py:758-772, :527): t first, then the noise
    T = model.diffusion.diffusion.num_timesteps
    t = torch.randint(0, T, size=(B,), device=device)
    noise = torch.randn(target.shape, dtype=target.dtype, device=device)
    # ---- conditioning signature
    ctx_mode, ctx_dim, ctx_param, ctx_rows = SS_NONE, 0, None, None
    if condition is not None:
        ctx_dim = condition.shape[-1]
        shared = condition.stride(0) == 0 or B == 1
        pe = getattr(model, "positional_embedding", None)
        if shared and pe is not None and condition.data_ptr() == pe.data_ptr() and tuple(condition.shape[1:]) == tuple(pe.shape):
            ctx_mode, ctx_param = SS_PER_SLOT, pe            # learnable instance embedding read in place, gradient -> G
        elif shared:
            ctx_mode, ctx_rows = SS_PER_SLOT, condition[0]
        else:
            ctx_mode, ctx_rows = SS_PER_TOKEN, condition.reshape(B * N, ctx_dim)
    L = text_dim = 0
    cross_rows = None
    if condition_cross is not None and model.diffusion.model.text_condition:
        L, text_dim = condition_cross.shape[1], condition_cross.shape[2]
        cross_rows = condition_cross.reshape(B * L, text_dim)
    ent = r.plan_for(B, N, ctx_mode, ctx_dim, L, text_dim, ctx_param)
    plan = ent["plan"]
    with torch.no_grad():
        plan.x0.copy_(target)
        plan.noise.copy_(noise)
        plan.t.copy_(t)
        if ctx_rows is not None:
            plan.ctx_in.t.copy_(ctx_rows)
        if cross_rows is not None:
            plan.cross_in.t.copy_(cross_rows)
    tail = [(x, g) for x, g in ((ctx_rows, plan.d_ctx), (cross_rows, plan.d_cross))
            if backward and x is not None and x.requires_grad]
    if backward:
        fl.zero_head()                      # autograd ACCUMULATES into the wrapper-level gradients (tiny region of G)
    r.run(ent, backward)
    if tail:
        torch.autograd.backward([x for x, _ in tail], [g.view_as(x) for x, g in tail])
    if backward and ent["reducer"] is not None:
        ent["reducer"].finish()
    loss = plan.losses.mean()
    means = plan.parts.mean(dim=0)
    if model.room_arrange_condition:
        loss_dict = {'loss.trans': means[1], 'loss.angle': means[3]}
    else:
        loss_dict = {k: means[i] for i, k in enumerate(_PART_KEYS)}
    return loss, loss_dict, ent
