#!/usr/bin/env python
"""Training-side PMC summary (tools/gpu_round.sh <tag> pmctrain): per kernel of the training step, MFMA-busy fraction per SIMD and
HBM bytes per launch (FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE), from three separate rocprofv3 --pmc passes."""
import glob
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = os.path.join(ROOT, "gpurun_out")


def table(sub):
    dbs = glob.glob(os.path.join(out, "%s_pmct_%s" % (tag, sub), "**", "*.db"), recursive=True)
    if not dbs:
        return {}
    cur = sqlite3.connect(dbs[0]).cursor()
    res = {}
    for k, c, n, v in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                  "group by kernel_name, counter_name"):
        res.setdefault(k, {})[c] = (n, v)
    return res


def durations(sub):
    dbs = glob.glob(os.path.join(out, "%s_pmct_%s" % (tag, sub), "**", "*.db"), recursive=True)
    if not dbs:
        return {}
    try:
        return {k: d / 1e3 for k, d in sqlite3.connect(dbs[0]).cursor().execute("select name, avg(end - start) from kernels group by name")}
    except sqlite3.Error:
        return {}


sq, fe, wr = table("sq"), table("fetch"), table("write")
dur = durations("sq")
rows = []
for k, c in sq.items():
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0.0))[1]
    act = c.get("GRBM_GUI_ACTIVE", (0, 0.0))[1]
    n = c.get("GRBM_GUI_ACTIVE", (0, 0.0))[0]
    frac = (busy / 1024.0) / (act / 8.0) if act else 0.0
    f = fe.get(k, {}).get("FETCH_SIZE", (0, 0.0))[1] * 1024 * 2
    w = wr.get(k, {}).get("WRITE_SIZE", (0, 0.0))[1] * 1024
    d = dur.get(k)
    rows.append((act * n, k, n, act / 8.0, frac, f, w, d, (act / 8.0 / d / 1e3) if d else 0.0))
rows.sort(reverse=True)
print("# training step PMC summary (%s): per-launch averages; busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs;" % tag)
print("# HBM bytes = FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, from separate passes of `bench.py --mode train`")
print("# clock_GHz = effective shader clock of the launch: cycles/XCD over its duration in the same (profiled) pass")
print("%-88s %6s %12s %9s %12s %12s %10s %9s" % ("kernel", "calls", "cycles/XCD", "mfma_busy", "fetch_MB", "write_MB", "avg_us", "clock_GHz"))
for _, k, n, cyc, frac, f, w, d, ghz in rows[:24]:
    print("%-88s %6d %12.0f %9.3f %12.2f %12.2f %10.2f %9.3f" % (k[:88], n, cyc, frac, f / 1e6, w / 1e6, d or 0.0, ghz))
