#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2) rocpd SQLite database into the text tables committed under profiles/:
per-kernel time (the `--kernel-trace --stats` view) and, when present, per-kernel averages of PMC counters.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_name.txt
"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                       "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1.0
    print("# kernel trace summary of %s" % path)
    print("# total kernel time %.1f us over %d dispatches" % (tot, sum(r[1] for r in rows)))
    print("%-96s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for r in rows:
        print("%-96s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (r[0][:96], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
    # timeline: how much of the traced span the GPU spent inside kernels, and where the idle time sits
    tl = cur.execute("select start, end, name from kernels order by start").fetchall()
    if tl:
        span = (max(r[1] for r in tl) - tl[0][0]) / 1e3
        busy_end, gaps = tl[0][1], []
        for st, en, nm in tl[1:]:
            if st > busy_end:
                gaps.append(((st - busy_end) / 1e3, nm))
            busy_end = max(busy_end, en)
        idle = sum(g for g, _ in gaps)
        print("\n# timeline: span %.1f us, idle between kernels %.1f us (%.1f %%) in %d gaps; gaps > 100 us: %d totalling %.1f us"
              % (span, idle, 100 * idle / span, len(gaps), sum(1 for g, _ in gaps if g > 100), sum(g for g, _ in gaps if g > 100)))
        big = {}
        for g, nm in gaps:
            if g > 100:
                k = big.setdefault(nm[:80], [0, 0.0])
                k[0] += 1; k[1] += g
        for nm, (c, t) in sorted(big.items(), key=lambda kv: -kv[1][1])[:8]:
            print("#   idle before %-80s %5d x, %10.1f us" % (nm, c, t))
    # steady-state steps: intervals between consecutive ends of a once-per-step kernel (the Adam sweep of a training step)
    marks = [r[1] for r in tl if "adam_kernel" in r[2]] if tl else []
    if len(marks) >= 3:
        print("\n# per-step accounting (step = interval between consecutive adam_kernel ends; last %d steps)" % min(len(marks) - 1, 4))
        for a, b in list(zip(marks, marks[1:]))[-4:]:
            ks = [(st, en, nm) for st, en, nm in tl if st >= a and en <= b]
            busy = sum(en - st for st, en, _ in ks) / 1e3
            gaps = sorted(((ks[i + 1][0] - ks[i][1]) / 1e3, ks[i + 1][2][:60]) for i in range(len(ks) - 1))
            pos = [g for g, _ in gaps if g > 0]
            print("#   span %.1f us, %d kernels, in-kernel %.1f us, between kernels %.1f us (median gap %.2f us, gaps > 20 us: %d = %.1f us)"
                  % ((b - a) / 1e3, len(ks), busy, sum(pos), pos[len(pos) // 2] if pos else 0.0, sum(1 for g in pos if g > 20),
                     sum(g for g in pos if g > 20)))
            for g_, nm in gaps[-4:][::-1]:
                print("#       %.1f us before %s" % (g_, nm))
    try:
        pmc = cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                          "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    except sqlite3.Error:
        pmc = []
    if pmc:
        print("\n# PMC counters: per-dispatch average (summed over XCDs/SEs as rocprofv3 reports them)")
        print("%-96s %-32s %7s %16s" % ("kernel", "counter", "n", "avg_value"))
        for r in pmc:
            print("%-96s %-32s %7d %16.1f" % (r[0][:96], r[1], r[2], r[3]))


if __name__ == "__main__":
    main(sys.argv[1])
