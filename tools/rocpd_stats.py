#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table (like --stats).

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/r1_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % disp)]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % sym)]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    q = ("select s.%s, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         "from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, disp, sym, name_col))
    rows = list(c.execute(q))
    total = sum(r[2] for r in rows) or 1
    print("%-90s %8s %12s %12s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for name, n, tot, avg, mn, mx in rows:
        print("%-90s %8d %12.1f %12.2f %12.2f %12.2f %6.2f%%" % (name[:90], n, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))


if __name__ == "__main__":
    main(sys.argv[1])
