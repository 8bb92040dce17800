"""Shape retrieval after sampling on the GPU -- the step right behind the DDPM path in generate_diffusion.py:335-348
(SURVEY.md 8f-3).  Mirrors ``ThreedFutureDataset.get_closest_furniture_to_objfeats`` and
``..._to_objfeats_and_size`` (scene_synthesis/datasets/threed_future_dataset.py:49-77) and adds the batched form a
B>1 generation needs: all boxes of all scenes in one launch instead of a Python loop over the object list per box."""

import numpy as np
import torch

from . import _lib, ops


class ShapeCodeIndex:
    """Device-resident copy of the object database: latent codes (n, 32) fp32, class ids, sizes (n, 3) float64."""

    def __init__(self, objects, device):
        assert len(objects) > 0
        self.objects = list(objects)
        self.device = torch.device(device)
        labels = sorted({o.label for o in self.objects})
        self.label_to_id = {l: i for i, l in enumerate(labels)}
        feats = np.stack([np.asarray(o.raw_model_norm_pc_lat32(), dtype=np.float32) for o in self.objects])
        if feats.shape[1] != 32:
            raise NotImplementedError("only the 32-d latent shape codes (objfeats_32) are indexed")
        self.feats = torch.from_numpy(np.ascontiguousarray(feats)).to(self.device)
        self.labels = torch.tensor([self.label_to_id[o.label] for o in self.objects], dtype=torch.int32, device=self.device)
        self.sizes = torch.from_numpy(np.stack([np.asarray(o.size, dtype=np.float64) for o in self.objects])).to(self.device)

    def __len__(self):
        return len(self.objects)

    def closest(self, query_labels, query_objfeats, query_sizes=None):
        """Batched retrieval.  query_labels: list of label strings (or int tensor of ids), query_objfeats (Q, 32) fp32,
        query_sizes optional (Q, 3) -> int32 tensor (Q,) of database indices (-1: no object with that label)."""
        if not torch.is_tensor(query_labels):
            query_labels = torch.tensor([self.label_to_id.get(l, -1) for l in query_labels], dtype=torch.int32)
        ql = query_labels.to(self.device, torch.int32).contiguous()
        qf = torch.as_tensor(query_objfeats, dtype=torch.float32).to(self.device).contiguous()
        Q = qf.shape[0]
        qs = None
        if query_sizes is not None:
            qs = torch.as_tensor(np.asarray(query_sizes, dtype=np.float64) if not torch.is_tensor(query_sizes)
                                 else query_sizes).to(self.device, torch.float64).contiguous()
        out = torch.empty((Q,), dtype=torch.int32, device=self.device)
        dist = torch.empty((Q,), dtype=torch.float32, device=self.device)
        _lib.check(_lib.fn("dsc_retrieve_nearest_f32")(
            qf.data_ptr(), ql.data_ptr(), qs.data_ptr() if qs is not None else None, self.feats.data_ptr(),
            self.labels.data_ptr(), self.sizes.data_ptr() if qs is not None else None, Q, len(self.objects), 32,
            out.data_ptr(), dist.data_ptr(), ops.stream_ptr()), "dsc_retrieve_nearest_f32")
        return out

    # --- reference-named single-query methods ------------------------------------------------------------------
    def _one(self, query_label, query_objfeat, query_size=None):
        idx = int(self.closest([query_label], np.asarray(query_objfeat, dtype=np.float32)[None],
                               None if query_size is None else np.asarray(query_size, dtype=np.float64)[None])[0])
        if idx < 0:
            raise IndexError("list index out of range")       # what sorted_mses[0] raises in the reference
        return self.objects[idx]

    def get_closest_furniture_to_objfeats(self, query_label, query_objfeat):
        return self._one(query_label, query_objfeat)

    def get_closest_furniture_to_objfeats_and_size(self, query_label, query_objfeat, query_size):
        return self._one(query_label, query_objfeat, query_size)
