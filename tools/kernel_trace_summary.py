#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel count / average / total, and -- for kernels that are launched several
times per training step with different work (the lazy K2's rounds) -- the average per position inside the step.
usage: kernel_trace_summary.py <kernel_trace.csv> [skip_first_n_dispatches]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("ngp::", "").replace("void ", "")
    return name[:70]


def main():
    path = sys.argv[1]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    if len(sys.argv) > 2:
        rows = rows[int(sys.argv[2]):]
    by = defaultdict(list)
    for s, e, n in rows:
        by[n].append((e - s) / 1e3)
    total = sum(sum(v) for v in by.values())
    print(f"{len(rows)} dispatches, {total / 1e3:.2f} ms of kernel time")
    print(f"{'kernel':70s} {'count':>8s} {'avg us':>9s} {'total ms':>9s} {'%':>6s}")
    for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print(f"{n:70s} {len(v):8d} {sum(v) / len(v):9.2f} {sum(v) / 1e3:9.2f} {100 * sum(v) / total:6.2f}")
    # K2 rounds: consecutive runs of the same tiles kernel = one step's rounds
    for key in [k for k in by if "k_inference_tiles" in k]:
        runs, cur = [], []
        for s, e, n in rows:
            if n == key:
                cur.append((e - s) / 1e3)
            elif cur:
                runs.append(cur); cur = []
        if cur:
            runs.append(cur)
        runs = [r for r in runs if len(r) == max(len(x) for x in runs)]
        if runs:
            k = len(runs[0])
            print(f"{key}: {len(runs)} steps x {k} rounds; average us per round: " + ", ".join(f"{sum(r[i] for r in runs) / len(runs):.1f}" for i in range(k)))


if __name__ == "__main__":
    main()
