#!/usr/bin/env python
"""Measurement for the shape-retrieval row: GPU kernel vs the numpy reference port on the host."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import retrieval_ref as RR
from diffuscene_amd.retrieval import ShapeCodeIndex

objs = RR.synth_objects(n=16000, n_labels=30, seed=3)
labels, feats, sizes = RR.synth_queries(objs, q=5120, seed=4)
idx = ShapeCodeIndex(objs, "cuda:0")
ql = torch.tensor([idx.label_to_id[l] for l in labels], dtype=torch.int32, device="cuda:0")
qf = torch.from_numpy(feats).cuda()
idx.closest(ql, qf); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    idx.closest(ql, qf)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
t0 = time.perf_counter()
for k in range(64):
    RR.closest_to_objfeats(objs, labels[k], feats[k])
cpu_q = 64 / (time.perf_counter() - t0)
scanned = 5120 * 16000 * 32 * 4          # bytes touched through L2 if every row were read (label filter skips ~29/30)
print(json.dumps({"metric": "shape retrieval queries/s (16k objects, 5120 queries)", "value": round(5120 / (ms * 1e-3)),
                  "ms_per_batch": round(ms, 3), "cpu_numpy_port_queries_per_s": round(cpu_q, 1),
                  "db_bytes": 16000 * 32 * 4, "note": "database (2 MB) is L2-resident; label test before the row load"}))
