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
