#!/usr/bin/env python
"""FoldingNet KL auto-encoder training step on the HIP kernels: ms per step (forward + Chamfer/KL loss + backward + clip + Adam)
at the reference's training shape (clouds of 2048 points, 2025-point reconstructions).
    python tools/bench_foldingnet.py [batch] [points]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import weights as W  # noqa: E402
from diffuscene_amd.networks.foldingnet_autoencoder import KLAutoEncoder, train_on_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = KLAutoEncoder(latent_dim=64, kl_weight=0.001).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
pc = W.synth_point_clouds(B, N, seed=1).to(dev)
cfg = {"training": {"max_grad_norm": 10}}
for _ in range(3):
    loss = train_on_batch(model, opt, {"points": pc}, cfg)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    loss = train_on_batch(model, opt, {"points": pc}, cfg)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print('{"what": "FoldingNet KLAutoEncoder train_on_batch", "batch": %d, "points": %d, "ms_per_step": %.2f, "clouds_per_s": %.1f, '
      '"loss": %.5f}' % (B, N, dt * 1e3, B / dt, loss))
