#!/usr/bin/env python
"""Per-launch table of one denoiser forward plan: every step timed alone with HIP events (median of several repeats), with
shape / FLOPs / TFLOP/s for the GEMM steps ("gemm*": bf16 weight planes supplied, i.e. the split-bf16 path where the shape qualifies).    python tools/plan_profile.py [config] [--batch B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from diffuscene_amd import _lib, ops  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "living80"
spec = dict(bench.CONFIGS[name])
for i, a in enumerate(sys.argv):
    if a == "--batch":
        spec["batch"] = int(sys.argv[i + 1])
dev = torch.device("cuda:0")
model, _ = bench.build_model(spec, dev)
gemm_f, gn_f = _lib.fn("dsc_gemm_f32"), _lib.fn("dsc_gemm_gn_silu_f32")
if "--train" in sys.argv:
    tr = bench.TrainRunner(spec, model, dev, 0)
    tr.run(2)
    ent = next(iter(model._dsc_plan_runner.plans.values()))
    tp = ent["plan"]
    steps = []
    for st in tp.fwd + tp.bwd:
        if isinstance(st, dict):
            st = st["step"]
        steps.append((st[0], st[1]))
    print("# training plan: %d forward(+loss) launches, %d backward launches" % (len(tp.fwd), len(tp.bwd)))
else:
    sr = bench.SampleRunner(spec, model, dev, seed=0)
    steps = list(sr.g.plan.steps)
s = ops.stream_ptr()
REPS = 7
rows = []
for idx, (f, a) in enumerate(steps):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(REPS)]
    f(*a, s)
    for e0, e1 in evs:
        e0.record()
        f(*a, s)
        f(*a, s)
        e1.record()
    torch.cuda.synchronize()
    us = sorted(e0.elapsed_time(e1) * 500.0 for e0, e1 in evs)[REPS // 2]
    desc, fl = f.__name__, 0.0
    if f.__name__ in ("dsc_gemm_tn_grouped_f32", "dsc_gemm_tn_grouped_split_f32"):
        desc = "%s groups=%s %s splits=%s" % (f.__name__, a[1], ("blocks=%s" % a[4]) if "split" in f.__name__ else ("tiles=%s" % a[2]),
                                             a[5] if "split" in f.__name__ else a[3])
    if f is gemm_f or f is gn_f:
        g = a[0]._obj
        fl = 2.0 * g.m * g.n * (g.k1 + g.k2) * max(g.batch, 1)
        tile = _lib.fn("dsc_gemm_split_tile")(g, 1 if f is gn_f else 0)
        desc = "%s%s m=%d n=%d k=%d+%d b=%d act=%d res=%d%s%s" % ("gn_gemm" if f is gn_f else "gemm", ("*t%d" % tile) if g.w_planes else "", g.m, g.n,
                                                                g.k1, g.k2, g.batch, g.act_out, 1 if g.residual else 0,
                                                                " +preact" if (g.preact and f is gemm_f) else "",
                                                                " *act'" if g.actgrad_x else "")
    rows.append((idx, desc, us, fl))
tot = sum(r[2] for r in rows)
print("# %s B=%d N=%d: %d launches, sum of isolated launch times %.1f us" % (name, spec["batch"], spec["objects"], len(rows), tot))
agg = {}
for idx, desc, us, fl in rows:
    print("%3d %-64s %8.1f us %7.1f TF" % (idx, desc, us, fl / us / 1e6 if fl else 0.0))
    k = agg.setdefault(desc, [0, 0.0, 0.0])
    k[0] += 1; k[1] += us; k[2] += fl
print("\n# aggregated by launch signature")
for desc, (n, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%3d x %-64s %9.1f us %5.1f%% %7.1f TF" % (n, desc, us, 100 * us / tot, fl / us / 1e6 if fl else 0.0))
