#!/usr/bin/env python
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd sqlite database. Usage: rocpd_pmc.py results.db [regex]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    cols = [d[1] for d in db.execute("pragma table_info(counters_collection)")]
    # columns of interest: kernel name, counter name, value
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = db.execute(f"select {name_col}, counter_name, avg(value), count(*), sum(value) from counters_collection group by {name_col}, counter_name")
    agg = {}
    for k, c, avg, n, tot in rows:
        if pat and not pat.search(k):
            continue
        agg.setdefault(k, {})[c] = (avg, n)
    for k, d in agg.items():
        print(k[:100])
        for c, (avg, n) in sorted(d.items()):
            print(f"    {c:32s} avg/dispatch = {avg:18.1f}   (n={n})")


if __name__ == "__main__":
    main()
