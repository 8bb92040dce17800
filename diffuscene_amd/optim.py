"""Fused clip + Adam for train_on_batch (reference: torch.nn.utils.clip_grad_norm_ at
diffusion_scene_layout_ddpm.py:465 and torch.optim.Adam from networks/__init__.py:29-30).

``FusedAdam`` IS a ``torch.optim.Adam`` (same constructor, ``param_groups``, ``state`` layout and ``state_dict`` format, so
``opt_XXXXX`` checkpoints are interchangeable with the reference's), but ``step()`` runs csrc/optim.hip: one sweep for the
global gradient norm, a device-side clip coefficient, one Adam sweep -- no host synchronisation, ~32 B/parameter of traffic.
"""

import numpy as np
import torch

from . import _lib

CHUNK = 32768

_WEIGHTS_EPOCH = [0]


def weights_epoch():
    """Counter bumped by every raw-pointer parameter update of this module (part of DenoiserEngine's signature)."""
    return _WEIGHTS_EPOCH[0]


def _bump_weights_epoch():
    _WEIGHTS_EPOCH[0] += 1


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self._plans = {}          # group index -> cached chunk plan
        self._steps = {}          # id(param) -> int step count (materialised into state['step'] lazily)
        self._active_cache = {}   # id(group) -> (fingerprint, validated parameter list)

    # ---- state compatibility with torch.optim.Adam -------------------------------------------------------------
    def _sync_step_tensors(self):
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if st is not None and id(p) in self._steps:
                    st["step"] = torch.tensor(float(self._steps[id(p)]), dtype=torch.float32)

    def state_dict(self):
        self._sync_step_tensors()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._steps = {}
        self._plans = {}
        self._active_cache = {}
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if st is not None and "step" in st:
                    self._steps[id(p)] = int(float(st["step"]))

    # ---- work lists -------------------------------------------------------------------------------------------------
    def _active(self, group):
        ps = [p for p in group["params"] if p.grad is not None]
        # validated once per (parameter set, storage): the 442-parameter loop below costs ~0.3 ms of host time per call, and
        # step() runs while the GPU waits for the Adam launch
        fast = (len(ps), ps[0].data_ptr() if ps else 0, ps[-1].data_ptr() if ps else 0, ps[0].grad.data_ptr() if ps else 0)
        cached = self._active_cache.get(id(group))
        if cached is not None and cached[0] == fast and all(a is b for a, b in zip(cached[1], ps)):
            return ps
        for p in ps:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise RuntimeError("FusedAdam: parameters must be contiguous float32 GPU tensors (no CPU fallback)")
            if p.grad.is_sparse:
                raise RuntimeError("FusedAdam does not support sparse gradients")
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                self._steps.setdefault(id(p), 0)
            elif id(p) not in self._steps:
                self._steps[id(p)] = int(float(st["step"]))
        self._active_cache[id(group)] = (fast, list(ps))
        return ps

    def _plan(self, gi, ps):
        key = tuple(id(p) for p in ps)
        plan = self._plans.get(gi)
        if plan is None or plan["key"] != key or any(pp != p.data_ptr() for pp, p in zip(plan["pptr"], ps)):
            numel = np.array([p.numel() for p in ps], dtype=np.int64)
            nch = (numel + CHUNK - 1) // CHUNK
            tix = np.repeat(np.arange(len(ps)), nch)
            first = np.concatenate([[0], np.cumsum(nch)[:-1]])
            off = (np.arange(int(nch.sum())) - np.repeat(first, nch)) * CHUNK
            cnt = np.minimum(CHUNK, numel[tix] - off)
            pptr = np.array([p.data_ptr() for p in ps], dtype=np.int64)
            mptr = np.array([self.state[p]["exp_avg"].data_ptr() for p in ps], dtype=np.int64)
            vptr = np.array([self.state[p]["exp_avg_sq"].data_ptr() for p in ps], dtype=np.int64)
            table = np.zeros((len(tix), 5), dtype=np.int64)
            table[:, 0] = pptr[tix] + off * 4
            table[:, 2] = mptr[tix] + off * 4
            table[:, 3] = vptr[tix] + off * 4
            table[:, 4] = cnt
            dev = ps[0].device
            plan = {"key": key, "pptr": pptr.tolist(), "tix": tix, "off4": off * 4, "table": table,
                    "host": torch.empty((len(tix), 5), dtype=torch.int64).pin_memory(),
                    "dev": torch.empty((len(tix), 5), dtype=torch.int64, device=dev),
                    "partial": torch.empty((len(tix),), dtype=torch.float64, device=dev),
                    "norm": torch.zeros((), dtype=torch.float32, device=dev),
                    "coef": torch.ones((), dtype=torch.float32, device=dev), "gptr": None, "uploaded": None}
            self._plans[gi] = plan
        for p in ps:
            if not p.grad.is_contiguous():
                p.grad = p.grad.contiguous()
        gptr = np.array([p.grad.data_ptr() for p in ps], dtype=np.int64)
        if plan.get("gptr") is None or not np.array_equal(gptr, plan["gptr"]):
            # the pinned staging buffer may still be the source of the previous step's asynchronous upload: wait for that
            # copy before rewriting it (callers other than train_on_batch do not synchronise between steps)
            if plan.get("uploaded") is not None:
                plan["uploaded"].synchronize()
            plan["table"][:, 1] = gptr[plan["tix"]] + plan["off4"]
            plan["host"].copy_(torch.from_numpy(plan["table"]))
            plan["dev"].copy_(plan["host"], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(plan["dev"].device))
            plan["uploaded"] = ev
            plan["gptr"] = gptr
        return plan

    # ---- clip + step ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def clip_grad_norm_(self, max_norm):
        """Global L2 norm of all gradients and the clip coefficient, both left on the device.  DEFERRED clip: unlike
        ``torch.nn.utils.clip_grad_norm_`` the gradients are NOT scaled in place -- the coefficient is applied inside the
        next ``step()`` while the Adam sweep reads them (readers of ``p.grad`` between the two calls see the unclipped
        values; the returned norm is the pre-clip norm, as in torch).  ``zero_grad()`` cancels a pending coefficient.
        With several param groups the gradients are clipped in place (torch semantics) and ``step()`` runs unclipped.
        Returns the total norm (0-d device tensor)."""
        plans = []
        for gi, group in enumerate(self.param_groups):
            ps = self._active(group)
            if ps:
                plans.append(self._plan(gi, ps))
        if not plans:
            return torch.zeros(())
        if len(plans) != 1:
            from .ddp import clip_grad_norm_fused
            for pl in plans:
                pl["clip_ready"] = False
            return clip_grad_norm_fused([p for g in self.param_groups for p in g["params"]], max_norm)
        plan = plans[0]
        dev = plan["dev"].device
        with torch.cuda.device(dev):
            s = torch.cuda.current_stream(dev).cuda_stream
            n = plan["dev"].shape[0]
            _lib.check(_lib.fn("dsc_grad_sumsq_f32")(plan["dev"].data_ptr(), n, plan["partial"].data_ptr(), s),
                       "dsc_grad_sumsq_f32")
            _lib.check(_lib.fn("dsc_clip_coef_f32")(plan["partial"].data_ptr(), n, float(max_norm), plan["norm"].data_ptr(),
                                                    plan["coef"].data_ptr(), s), "dsc_clip_coef_f32")
        plan["clip_ready"] = True
        return plan["norm"]

    def cancel_pending_clip(self):
        for plan in self._plans.values():
            plan["clip_ready"] = False          # a coefficient computed for gradients that no longer exist must not be applied

    def zero_grad(self, set_to_none=True):
        self.cancel_pending_clip()
        return super().zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = self._active(group)
            if not ps:
                continue
            clip = bool(self._plans.get(gi, {}).get("clip_ready"))
            plan = self._plans[gi] if clip else self._plan(gi, ps)     # clip_grad_norm_ already uploaded the work list
            plan["clip_ready"] = False
            b1, b2 = group["betas"]
            steps = sorted({self._steps[id(p)] for p in ps})
            dev = plan["dev"].device
            with torch.cuda.device(dev):
                s = torch.cuda.current_stream(dev).cuda_stream
                if len(steps) == 1:
                    sel = [(steps[0], plan["dev"], plan["dev"].shape[0])]
                else:      # parameters that joined later carry their own bias correction
                    sel = []
                    pstep = np.array([self._steps[id(p)] for p in ps])
                    for sv in steps:
                        rows = torch.from_numpy(np.nonzero(pstep[plan["tix"]] == sv)[0]).to(dev)
                        sub = plan["dev"].index_select(0, rows).contiguous()
                        sel.append((sv, sub, sub.shape[0]))
                for sv, tab, n in sel:
                    t = sv + 1
                    bc1 = 1.0 - b1 ** t
                    bc2_sqrt = (1.0 - b2 ** t) ** 0.5
                    _lib.check(_lib.fn("dsc_adam_step_f32")(tab.data_ptr(), n, group["lr"] / bc1, b1, b2, bc2_sqrt,
                                                            group["eps"], group["weight_decay"],
                                                            plan["coef"].data_ptr() if clip else None, s),
                               "dsc_adam_step_f32")
            for p in ps:
                self._steps[id(p)] += 1
            # the kernel wrote the parameters through raw pointers: tell autograd / the engine's derived-weight cache
            # (DenoiserEngine._signature reads p._version) that they changed
            torch.autograd.graph.increment_version(ps)
            _bump_weights_epoch()
        return loss
