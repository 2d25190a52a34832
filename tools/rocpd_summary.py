#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel table that `--stats` prints:
calls, total / average / min / max duration. Usage: rocpd_summary.py results.db [out.csv]"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), max(vgpr_count), max(accum_vgpr_count), "
                           "max(scratch_size), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "VGPR", "AGPR", "ScratchBytes", "LDSBytes", "GridX", "WorkgroupX"])
    for r in rows:
        out.writerow([r[0], r[1], int(r[2]), round(r[3], 1), round(100.0 * r[2] / tot, 3), r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11]])


if __name__ == "__main__":
    main()
