"""Flat parameter / gradient storage of the training step (new work: the reference keeps 442 separate tensors and lets
autograd allocate 442 gradients every step).

``FlatStorage`` re-homes every trainable parameter of a module as a view into ONE contiguous fp32 buffer ``P`` and its
``.grad`` as the matching view into ONE buffer ``G`` (311 MB each for the 77.7 M-parameter denoiser):

* the static training plan (train_plan.py) writes weight gradients straight into ``G`` -- nothing is allocated,
  concatenated or copied per step;
* the data-parallel reducer (ddp.py) all-reduces contiguous slices of ``G`` in place, bucket by bucket, in the order the
  backward pass finishes them;
* the packed conditioning weights of the denoiser -- the 19 time-MLP ``Linear(2048 -> 1024)`` and the 9 context-MLP
  ``Linear(ctx -> 1024)`` of the ResnetBlocks (denoise_net.py:181-184), which the engine runs as ONE GEMM each -- are laid out
  contiguously, so the packed matrices (and their gradients) are plain views of ``P`` / ``G`` instead of per-step
  ``torch.cat`` copies.

``state_dict`` / ``load_state_dict`` / optimizers are unaffected: parameters stay ordinary ``nn.Parameter`` objects with the
reference's names and shapes; only their storage moved.
"""
import torch

ALIGN = 64          # floats: every parameter starts on a 256-byte boundary


def find_unet(module):
    from .networks.denoise_net import Unet1D
    for m in module.modules():
        if isinstance(m, Unet1D):
            return m
    return None


class FlatStorage:
    def __init__(self, module):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("FlatStorage: module has no trainable parameters")
        dev = params[0].device
        for p in params:
            if p.device != dev or p.dtype != torch.float32:
                raise RuntimeError("FlatStorage: all trainable parameters must be fp32 on one device")
        self.module = module
        self.device = dev
        net = find_unet(module)
        packed = {}                                     # name -> list of parameters laid out back to back
        if net is not None:
            t_blocks = [rb for rb, kind in net.resblocks_in_order() if kind == "t"]
            # (an un-conditioned network -- context_dim + instanclass_dim = 0 -- keeps the reference's Linear(0, 1024) in its context blocks:
            # never applied, nothing to pack; its bias stays an ordinary parameter whose gradient is never written, as in the reference)
            c_blocks = [rb for rb, kind in net.resblocks_in_order() if kind == "c" and rb.mlp is not None and rb.mlp[1].weight.shape[1] > 0]
            packed = {"c_w": [rb.mlp[1].weight for rb in c_blocks], "c_b": [rb.mlp[1].bias for rb in c_blocks],
                      "t_w": [rb.mlp[1].weight for rb in t_blocks], "t_b": [rb.mlp[1].bias for rb in t_blocks]}
        in_pack = {id(p) for ps in packed.values() for p in ps}
        in_net = {id(p) for p in net.parameters()} if net is not None else set()
        head = [p for p in params if id(p) not in in_net]                   # wrapper-level parameters (autograd side)
        body = [p for p in params if id(p) in in_net and id(p) not in in_pack]
        order, seen = [], set()
        # order of G = order in which data parallelism can ship it: the backward finishes the body from its END (decoder
        # heads first) and the packed time-MLP rows from the last block backwards; the small packs that are only complete
        # at the very end of the backward (context MLPs, time-MLP biases) sit next to the wrapper-level parameters in the
        # first bucket, which is reduced last
        small = [p for k in ("c_w", "c_b", "t_b") for p in packed.get(k, []) if p.requires_grad]
        big = [p for p in packed.get("t_w", []) if p.requires_grad]
        for p in head + small + body + big:
            if id(p) not in seen:
                seen.add(id(p))
                order.append(p)
        self.params = order
        self.offset = {}
        off = 0
        pack_first = {id(ps[0]): k for k, ps in packed.items() if ps}
        pack_member = {id(p): k for k, ps in packed.items() for p in ps}
        self.pack_range = {}
        for p in order:
            k = pack_member.get(id(p))
            if k is None or id(p) in pack_first:
                off = (off + ALIGN - 1) // ALIGN * ALIGN          # packed members follow each other without padding
            if id(p) in pack_first:
                self.pack_range[k] = [off, 0]
            self.offset[id(p)] = off
            off += p.numel()
            if k is not None:
                self.pack_range[k][1] = off - self.pack_range[k][0]
        self.head_floats = 0
        if head:
            last = head[-1]
            self.head_floats = (self.offset[id(last)] + last.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.numel = (off + ALIGN - 1) // ALIGN * ALIGN
        self.P = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.G = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p in order:
                o, n = self.offset[id(p)], p.numel()
                v = self.P[o:o + n].view(p.shape)
                v.copy_(p.data)
                p.data = v
        self.attach_grads()
        self.packed = {}
        for k, ps in packed.items():
            if not ps or k not in self.pack_range:
                continue
            o, n = self.pack_range[k]
            shape = (n // ps[0].shape[1], ps[0].shape[1]) if ps[0].dim() == 2 else (n,)
            self.packed[k] = (self.P[o:o + n].view(shape), self.G[o:o + n].view(shape))
        if net is not None:
            object.__setattr__(net, "_flat", self)
        object.__setattr__(module, "_dsc_flat", self)

    # copy.deepcopy(model) (an EMA copy) / pickling must not drag the flat buffers, the ctypes argument tables or the captured
    # graphs along: the copy gets no flat storage and builds its own on its first training step (ensure_flat)
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (type(None), ())

    def attach_grads(self):
        """(Re-)point every ``p.grad`` at its slice of G (``optimizer.zero_grad(set_to_none=True)`` drops them)."""
        for p in self.params:
            g = p.grad
            o, n = self.offset[id(p)], p.numel()
            if g is None or g.data_ptr() != self.G.data_ptr() + 4 * o:
                p.grad = self.G[o:o + n].view(p.shape)

    def valid(self):
        """False once something (``module.to()``, a manual ``p.data = ...``) moved a parameter out of P."""
        base = self.P.data_ptr()
        return all(p.data_ptr() == base + 4 * self.offset[id(p)] for p in self.params)

    def grad_view(self, p):
        o, n = self.offset[id(p)], p.numel()
        return self.G[o:o + n].view(p.shape)

    def grad_range(self, p):
        return self.offset[id(p)], p.numel()

    def zero_head(self):
        """Wrapper-level parameters receive their gradients from autograd, which ACCUMULATES into a defined ``.grad``."""
        if self.head_floats:
            self.G[:self.head_floats].zero_()

    def buckets(self, n_buckets):
        """Contiguous [start, end) float ranges of G of roughly equal size, cut at parameter boundaries."""
        n_buckets = max(1, int(n_buckets))
        target = (self.numel + n_buckets - 1) // n_buckets
        cuts, start = [], 0
        starts = sorted(self.offset[id(p)] for p in self.params)
        for s in starts[1:]:
            if s - start >= target:
                cuts.append((start, s))
                start = s
        cuts.append((start, self.numel))
        return cuts


def ensure_flat(module):
    """The module's FlatStorage (created on first use, rebuilt when parameters were moved or added)."""
    fs = getattr(module, "_dsc_flat", None)
    if fs is not None:
        cur = [p for p in module.parameters() if p.requires_grad]
        if len(cur) == len(fs.params) and fs.valid():
            fs.attach_grads()
            return fs
    return FlatStorage(module)
