"""Chamfer timing on the workload of the reference's own unit test (ChamferDistancePytorch/unit_test.py:39-50:
clouds 32x2000x3 and 32x1000x3, forward + backward of sum(dist1)) plus the FoldingNet AE shape (B=32, 2048 vs 2025)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(B, n, m, iters=100):
    from diffuscene_amd.chamfer import chamfer_3DDist
    cd = chamfer_3DDist()
    p1 = torch.rand(B, n, 3, device="cuda", requires_grad=True)
    p2 = torch.rand(B, m, 3, device="cuda")
    for _ in range(5):
        d1, d2, _, _ = cd(p1, p2)
        d1.sum().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        d1, d2, _, _ = cd(p1, p2)
        d1.sum().backward()
    torch.cuda.synchronize()
    fb = (time.perf_counter() - t0) / iters
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        e0.record()
        for _ in range(iters):
            cd(p1, p2)
        e1.record()
    torch.cuda.synchronize()
    fwd = e0.elapsed_time(e1) / iters * 1e-3
    pairs = 2.0 * B * n * m
    return {"B": B, "n": n, "m": m, "fwd_bwd_ms": round(fb * 1e3, 4), "fwd_ms": round(fwd * 1e3, 4),
            "fwd_gpairs_per_s": round(pairs / fwd / 1e9, 1)}


def cpu_port(B, n, m):
    from oracle import chamfer_ref as CR          # bench baseline leg only
    a, b = CR.synth_clouds(B, n, m, 0)
    t0 = time.perf_counter()
    CR.chamfer_forward(a, b)
    return time.perf_counter() - t0


if __name__ == "__main__":
    res = [run(32, 2000, 1000), run(32, 2048, 2025), run(4, 100, 200)]
    cpu = cpu_port(4, 2000, 1000)
    print(json.dumps({"metric": "chamfer3D forward+backward", "cases": res,
                      "cpu_numpy_port_fwd_s_B4_2000x1000": round(cpu, 3),
                      "cpu_numpy_port_gpairs_per_s": round(2.0 * 4 * 2000 * 1000 / cpu / 1e9, 3)}))
